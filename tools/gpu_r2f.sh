#!/bin/bash
# round-2 GPU batch F: submit / collect pipeline, library-owned RCCL communicator (one rank), shared-GPU dry run of bench_fastddc --gpus 2
cd $GRAFT_REPO_ROOT; out=gpurun_out/r2f; mkdir -p $out
sel="c4 or bank or fastddc or general"
timeout 600 python -m pytest tests/test_configs_gpu.py tests/test_gpu_parity.py tests/test_cli_gpu.py -m gpu -q --tb=short -k "$sel" > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log; tail -25 $out/pytest.log
timeout 300 python bench_fastddc.py --verify > $out/bench_fastddc.json 2> $out/bench_fastddc.err; tail -c 2600 $out/bench_fastddc.json
CSDR_BENCH_SHARED_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench_fastddc.py --gpus 2 --steps 20 --no-cpu-baseline > $out/bench_fastddc_shared2.json 2> $out/bench_fastddc_shared2.err; tail -c 700 $out/bench_fastddc_shared2.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench_fastddc.py --steps 10 --warmup 2 --no-cpu-baseline > $out/trace.log 2>&1
ks=$(find $out/trace -name "*kernel_stats.csv" | head -1); [ -n "$ks" ] && python tools/tidy_kernel_stats.py $ks $out/r2f_fastddc_kernel_stats.csv "r2f: rocprofv3 --kernel-trace --stats -- python bench_fastddc.py --steps 10 --warmup 2 (ns)" && head -10 $out/r2f_fastddc_kernel_stats.csv | cut -c1-150
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r2f/trace/runc/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[-14]['Start_Timestamp'])
for r in rows[-14:]:
    print("%9.1f %8.1f q%s %s" % ((int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r['Queue_Id'], r['Kernel_Name'][:70]))
PY
for f in $out/*.err; do [ -s $f ] && { grep -v amdgpu.ids $f | tail -6 | cut -c1-300; }; done
