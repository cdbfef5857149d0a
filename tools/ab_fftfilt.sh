#!/bin/bash
# tools/ab_fftfilt.sh -- the one-pass FFT filter kernel at 1023 / 2047 / 4095 taps under CSDR_AMD_FFTFILT_LDS_MODE (A/B of the window kernels' variants), with verify
for spec in "1023 0" "1023 1" "1023 3" "2047 0" "2047 2" "4095 0" "4095 1" "4095 2"; do
  set -- $spec
  CSDR_AMD_FFTFILT_LDS_MODE=$2 timeout 200 python bench_fftfilt.py --steps 100 --no-sweep --taps $1 --no-cpu-baseline --verify 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith(chr(123))][-1]); r=d['roofline']; print('taps $1 mode $2', r['kernel'], r['kernel_avg_ms'], r['frac'], d['verify']['ok'])"
done
