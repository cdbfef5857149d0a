#!/bin/bash
# round-2 GPU batch N: variants of the one-pass FFT filter kernel (prefetch / residency)
cd $GRAFT_REPO_ROOT; out=gpurun_out/r2n; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "bandpass" 2>&1 | tail -3
for mode in 0 1 2 3; do
  CSDR_AMD_FFTFILT_LDS_MODE=$mode timeout 200 python bench_fftfilt.py --steps 50 --no-sweep --no-cpu-baseline --method full 2>/dev/null >/dev/null
  CSDR_AMD_FFTFILT_LDS_MODE=$mode timeout 200 python bench_fftfilt.py --steps 50 --no-sweep --no-cpu-baseline --verify 2> $out/m$mode.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mode $mode', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'], d['verify']['ok'], d['verify']['max_rel_rms'])"
  CSDR_AMD_FFTFILT_LDS_MODE=$mode timeout 200 python bench_fftfilt.py --steps 50 --no-sweep --no-cpu-baseline --taps 2047 --verify 2>> $out/m$mode.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mode $mode taps 2047', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'], d['verify']['ok'])"
done
for f in $out/*.err; do [ -s $f ] && { grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $f | tail -3 | cut -c1-300; }; done
