#!/bin/bash
# tools/ab_fftwave.sh -- the 4096-point window of the one-pass FFT filter: 256-thread kernel (mode 5) against the wave-per-window kernel (mode 0, the default), with verify
for spec in ${SPECS:-"1023 5" "1023 0" "511 5" "511 0" "255 5" "255 0" "63 5" "63 0"}; do
  set -- $spec
  CSDR_AMD_FFTFILT_LDS_MODE=$2 timeout 200 python bench_fftfilt.py --steps 100 --no-sweep --taps $1 --no-cpu-baseline --verify 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith(chr(123))][-1]); r=d['roofline']; print('taps $1 mode $2', r['kernel'], r['kernel_avg_ms'], r['frac'], d['verify']['ok'], d['verify']['max_rel_rms'])"
done
