#!/bin/bash
# round-2 GPU batch M: the one-pass FFT filter kernel (fftfilt_lds.hip): parity tests, bench line, window-size comparison
cd $GRAFT_REPO_ROOT; out=gpurun_out/r2m; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "bandpass or fft" 2>&1 | tail -5
timeout 300 python bench_fftfilt.py --verify --steps 100 > $out/r2m_fftfilt_n1.json 2> $out/fftfilt.err; cut -c1-1500 $out/r2m_fftfilt_n1.json
for n in 8192 16384; do
  CSDR_AMD_FFTFILT_LDS_N=$n timeout 200 python bench_fftfilt.py --steps 50 --no-sweep --no-cpu-baseline 2> $out/n$n.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'])"
done
for f in $out/*.err; do [ -s $f ] && { grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $f | tail -3 | cut -c1-300; }; done
