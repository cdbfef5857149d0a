#!/bin/bash
# round-2 GPU batch E: half-size inverse transforms, forward pass tile A-B
cd $GRAFT_REPO_ROOT; out=gpurun_out/r2e; mkdir -p $out
sel="c4 or bank or fastddc or general"
timeout 600 python -m pytest tests/test_configs_gpu.py tests/test_gpu_parity.py tests/test_cli_gpu.py -m gpu -q --tb=short -k "$sel" > $out/pytest_default.log 2>&1; echo "rc=$?" >> $out/pytest_default.log; tail -6 $out/pytest_default.log
CSDR_AMD_DDC_IFFT=16 CSDR_AMD_DDC_FWD=16 timeout 600 python -m pytest tests/test_configs_gpu.py -m gpu -q --tb=short -k "$sel" > $out/pytest_16.log 2>&1; echo "rc=$?" >> $out/pytest_16.log; tail -4 $out/pytest_16.log
CSDR_AMD_DDC_IFFT=512 timeout 600 python -m pytest tests/test_configs_gpu.py -m gpu -q --tb=short -k "$sel" > $out/pytest_512.log 2>&1; echo "rc=$?" >> $out/pytest_512.log; tail -4 $out/pytest_512.log
for i in 8 16 512 512,16; do for f in 8 16; do
  echo "== ifft=$i fwd=$f"; CSDR_AMD_DDC_IFFT=$i CSDR_AMD_DDC_FWD=$f timeout 200 python bench_fastddc.py --steps 300 --no-cpu-baseline 2> $out/b_${i}_$f.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"
done; done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench_fastddc.py --steps 10 --warmup 2 --no-cpu-baseline > $out/trace.log 2>&1
ks=$(find $out/trace -name "*kernel_stats.csv" | head -1); [ -n "$ks" ] && python tools/tidy_kernel_stats.py $ks $out/r2e_fastddc_kernel_stats.csv "r2e: rocprofv3 --kernel-trace --stats -- python bench_fastddc.py --steps 10 --warmup 2 (ns)" && head -10 $out/r2e_fastddc_kernel_stats.csv | cut -c1-150
CSDR_AMD_DDC_IFFT=16 CSDR_AMD_DDC_FWD=16 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace2 -- python bench_fastddc.py --steps 10 --warmup 2 --no-cpu-baseline > $out/trace2.log 2>&1
ks=$(find $out/trace2 -name "*kernel_stats.csv" | head -1); [ -n "$ks" ] && python tools/tidy_kernel_stats.py $ks $out/r2e_16_kernel_stats.csv "r2e 16/16" && head -8 $out/r2e_16_kernel_stats.csv | cut -c1-150
for f in $out/*.err; do [ -s $f ] && { grep -v amdgpu.ids $f | tail -3 | cut -c1-300; }; done
