#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r2i; mkdir -p $out
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_compat_gpu.py tests/test_cli_gpu.py -m gpu -q --tb=short -k "bandpass or fft" 2>&1 | tail -4
for g in 0 256; do echo "== group=$g"; CSDR_AMD_FFT64K_GROUP=$g timeout 200 python bench_fftfilt.py --steps 100 --no-cpu-baseline --no-sweep 2> $out/b_$g.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'])"; done
timeout 600 bash tools/profile_bench.sh r2i_fftfilt k_f64 bench_fftfilt.py --no-sweep > $out/profile_fftfilt.log 2>&1; tail -8 $out/profile_fftfilt.log | cut -c1-200
