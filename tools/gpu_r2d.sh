#!/bin/bash
# round-2 GPU batch D: persistent fold / 8-block inverse transforms A-B, coalesced chains
cd $GRAFT_REPO_ROOT; out=gpurun_out/r2d; mkdir -p $out
sel="c4 or bank or fastddc or general"
timeout 600 python -m pytest tests/test_configs_gpu.py tests/test_gpu_parity.py tests/test_cli_gpu.py -m gpu -q --tb=short -k "$sel" > $out/pytest_default.log 2>&1; echo "rc=$?" >> $out/pytest_default.log; tail -4 $out/pytest_default.log
CSDR_AMD_DDC_GEMM=simple CSDR_AMD_DDC_IFFT=16 timeout 600 python -m pytest tests/test_configs_gpu.py -m gpu -q --tb=short -k "$sel" > $out/pytest_simple16.log 2>&1; echo "rc=$?" >> $out/pytest_simple16.log; tail -4 $out/pytest_simple16.log
for g in persist simple; do for i in 8 16; do
  echo "== gemm=$g ifft=$i"; CSDR_AMD_DDC_GEMM=$g CSDR_AMD_DDC_IFFT=$i timeout 200 python bench_fastddc.py --steps 300 --no-cpu-baseline 2> $out/b_${g}_$i.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"
done; done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench_fastddc.py --steps 10 --warmup 2 --no-cpu-baseline > $out/trace.log 2>&1
ks=$(find $out/trace -name "*kernel_stats.csv" | head -1); [ -n "$ks" ] && python tools/tidy_kernel_stats.py $ks $out/r2d_fastddc_kernel_stats.csv "r2d: rocprofv3 --kernel-trace --stats -- python bench_fastddc.py --steps 10 --warmup 2 (ns)" && head -12 $out/r2d_fastddc_kernel_stats.csv | cut -c1-150
CSDR_AMD_DDC_GEMM=simple CSDR_AMD_DDC_IFFT=16 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace2 -- python bench_fastddc.py --steps 10 --warmup 2 --no-cpu-baseline > $out/trace2.log 2>&1
ks=$(find $out/trace2 -name "*kernel_stats.csv" | head -1); [ -n "$ks" ] && python tools/tidy_kernel_stats.py $ks $out/r2d_fastddc_simple16_kernel_stats.csv "r2d simple/16" && head -8 $out/r2d_fastddc_simple16_kernel_stats.csv | cut -c1-150
for f in $out/*.err; do [ -s $f ] && { echo "== $f"; tail -3 $f | cut -c1-300; }; done
