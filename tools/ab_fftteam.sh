#!/bin/bash
# tools/ab_fftteam.sh -- the 8192- / 16384-point windows of the one-pass FFT filter: the 512-thread kernels (mode 5) against the team kernels (mode 0, the default), with verify
for spec in ${SPECS:-2047:5 2047:0 4095:5 4095:0 3071:0 1535:0}; do
  t=${spec%%:*}; m=${spec##*:}
  CSDR_AMD_FFTFILT_LDS_MODE=$m timeout 200 python bench_fftfilt.py --steps 100 --no-sweep --taps $t --no-cpu-baseline --verify 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith(chr(123))][-1]); r=d['roofline']; print('taps $t mode $m', r['kernel'], r['kernel_avg_ms'], r['frac'], d['verify']['ok'], d['verify']['max_rel_rms'])"
done
