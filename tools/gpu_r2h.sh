#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r2h; mkdir -p $out
for g in 0 32 64 128 256; do echo "== group=$g"; CSDR_AMD_FFT64K_GROUP=$g timeout 200 python bench_fftfilt.py --steps 100 --no-cpu-baseline --no-sweep 2> $out/b_$g.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'])"; done
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -k "bandpass" 2>&1 | tail -3
