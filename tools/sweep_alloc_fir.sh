#!/bin/bash
# tools/sweep_alloc_fir.sh -- config 1, new process per run, alternating torch's allocation and a physically contiguous one for the 4.9 GB input
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
for i in 1 2 3 4 5 6; do
  for a in torch contig; do
    CSDR_BENCH_ALLOC=$a timeout 100 python bench_fir.py --steps 200 --no-cpu-baseline 2>&1 | python -c "import sys,json; ls=[l for l in sys.stdin.read().splitlines() if l.startswith('{')]; d=json.loads(ls[-1]) if ls else None; print('run $i  %-7s' % '$a', ('kernel %.4f ms frac %.4f' % (d['roofline']['kernel_avg_ms'], d['roofline']['frac'])) if d else 'FAILED')"
  done
done
