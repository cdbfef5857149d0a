#!/bin/bash
# round-2 GPU batch A: all GPU tests, every config's bench line with --verify, CLI pipe bench, fftfilt profile
cd $GRAFT_REPO_ROOT; out=gpurun_out/r2a; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q --tb=short -x > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -15 $out/pytest.log
timeout 300 python bench.py --verify > $out/bench.json 2> $out/bench.err; tail -c 3000 $out/bench.json
timeout 300 python bench_nfm.py --verify > $out/bench_nfm.json 2> $out/bench_nfm.err; tail -c 2500 $out/bench_nfm.json
timeout 300 python bench_fir.py --verify > $out/bench_fir.json 2> $out/bench_fir.err; tail -c 2000 $out/bench_fir.json
timeout 300 python bench_fir.py --decimation 50 --tbw 0.005 --streams 64 --verify > $out/bench_fir50.json 2> $out/bench_fir50.err; tail -c 2000 $out/bench_fir50.json
timeout 300 python bench_fftfilt.py --verify > $out/bench_fftfilt.json 2> $out/bench_fftfilt.err; tail -c 2500 $out/bench_fftfilt.json
timeout 300 python bench_fastddc.py --blocks 64 --steps 50 > $out/bench_fastddc.json 2> $out/bench_fastddc.err; tail -c 1500 $out/bench_fastddc.json
timeout 300 bash tools/bench_cli.sh > $out/cli_bench.txt 2>&1; cat $out/cli_bench.txt
timeout 600 bash tools/profile_bench.sh r2a_fftfilt k_f64 bench_fftfilt.py --no-sweep > $out/profile_fftfilt.log 2>&1; tail -12 $out/profile_fftfilt.log
for f in $out/*.err; do [ -s $f ] && { echo "== $f"; tail -5 $f; }; done
