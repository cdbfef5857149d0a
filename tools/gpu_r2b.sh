#!/bin/bash
# round-2 GPU batch B: the matrix-core fastddc path (parity + bench + kernel trace), CLI tests, bench_fir crash trace
cd $GRAFT_REPO_ROOT; out=gpurun_out/r2b; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q --tb=short > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -40 $out/pytest.log
timeout 300 python bench_fastddc.py --verify > $out/bench_fastddc.json 2> $out/bench_fastddc.err; tail -c 3000 $out/bench_fastddc.json
CSDR_AMD_DDC_MFMA_OFF=1 timeout 300 python bench_fastddc.py --no-cpu-baseline --steps 50 > $out/bench_fastddc_general.json 2> $out/bench_fastddc_general.err; tail -c 600 $out/bench_fastddc_general.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench_fastddc.py --steps 10 --warmup 2 --no-cpu-baseline > $out/trace.log 2>&1
ks=$(find $out/trace -name "*kernel_stats.csv" | head -1); [ -n "$ks" ] && python tools/tidy_kernel_stats.py $ks $out/r2b_fastddc_kernel_stats.csv "r2b: rocprofv3 --kernel-trace --stats -- python bench_fastddc.py --steps 10 --warmup 2 (ns)" && head -12 $out/r2b_fastddc_kernel_stats.csv | cut -c1-160
timeout 200 python -X faulthandler bench_fir.py --steps 5 --no-cpu-baseline > $out/bench_fir.json 2> $out/bench_fir.err; tail -c 1500 $out/bench_fir.json; tail -30 $out/bench_fir.err
timeout 200 python -X faulthandler bench_fir.py --steps 5 --no-cpu-baseline --streams 64 --block 2097152 > $out/bench_fir_b.json 2> $out/bench_fir_b.err; tail -c 800 $out/bench_fir_b.json; tail -5 $out/bench_fir_b.err
for f in $out/bench_fastddc*.err; do [ -s $f ] && { echo "== $f"; tail -5 $f; }; done
