#!/bin/bash
# round-2 GPU batch C: fused forward transform + side-stream chains (parity, bench, kernel trace), bench_fir lines
cd $GRAFT_REPO_ROOT; out=gpurun_out/r2c; mkdir -p $out
timeout 900 python -m pytest tests/test_configs_gpu.py tests/test_gpu_parity.py tests/test_cli_gpu.py tests/test_compat_gpu.py -m gpu -q --tb=short -k "c4 or bank or fastddc or general" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -30 $out/pytest.log
timeout 300 python bench_fastddc.py --verify --no-cpu-baseline > $out/bench_fastddc.json 2> $out/bench_fastddc.err; tail -c 1800 $out/bench_fastddc.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench_fastddc.py --steps 10 --warmup 2 --no-cpu-baseline > $out/trace.log 2>&1
ks=$(find $out/trace -name "*kernel_stats.csv" | head -1); [ -n "$ks" ] && python tools/tidy_kernel_stats.py $ks $out/r2c_fastddc_kernel_stats.csv "r2c: rocprofv3 --kernel-trace --stats -- python bench_fastddc.py --steps 10 --warmup 2 (ns)" && head -14 $out/r2c_fastddc_kernel_stats.csv | cut -c1-150
timeout 300 python bench_fir.py --verify > $out/bench_fir.json 2> $out/bench_fir.err; tail -c 1800 $out/bench_fir.json
timeout 300 python bench_fir.py --decimation 50 --tbw 0.005 --streams 64 --verify > $out/bench_fir50.json 2> $out/bench_fir50.err; tail -c 1800 $out/bench_fir50.json
for f in $out/*.err; do [ -s $f ] && { echo "== $f"; tail -5 $f; }; done
