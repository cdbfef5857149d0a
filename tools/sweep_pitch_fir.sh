#!/bin/bash
# tools/sweep_pitch_fir.sh -- bench_fir.py (config 1) with padded input row pitches, both kernels (is the 256-row pitch the cause of C1's box spread / 0.62?)
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
for pad in 0 16 80 272 1040 4112; do
  for k5 in 1 0; do
    CSDR_AMD_FIR_MFMA5=$k5 CSDR_BENCH_PITCH_PAD=$pad timeout 100 python bench_fir.py --steps 100 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d['roofline']; print('pad %5d  %-12s kernel %.4f ms frac %.4f' % ($pad, r['kernel'], r['kernel_avg_ms'], r['frac']))"
  done
done
