#!/bin/bash
# round-2 GPU batch Q: the one-pass FFT filter as committed: parity tests, the measured line with sweep + CPU baseline, kernel trace + PMC traffic
cd $GRAFT_REPO_ROOT; out=gpurun_out/r2q; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_cli_gpu.py tests/test_compat_gpu.py -q -x 2>&1 | tail -3
timeout 300 python bench_fftfilt.py --verify > $out/r2q_fftfilt_n1.json 2> $out/fftfilt.err; cut -c1-400 $out/r2q_fftfilt_n1.json
timeout 600 bash tools/profile_bench.sh r2q_fftfilt k_fftfilt_lds bench_fftfilt.py --no-sweep > $out/prof.log 2>&1; tail -6 $out/prof.log | cut -c1-200
for f in $out/*.err; do [ -s $f ] && { grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $f | tail -3 | cut -c1-300; }; done
