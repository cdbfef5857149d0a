#!/bin/bash
# tools/sweep_time_fir.sh -- the SAME bench_fir.py (config 1) run eight times in a row on a fresh box, with the GPU's clocks / power / temperature between runs:
# is C1's "box spread" (0.70 on some runs, 0.62 on others: VERDICT r4 weak #6) a matter of time since the box went busy?
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
smi() { rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Average Graphics Package Power|Current Socket Graphics Package Power|Temperature \(Sensor (junction|memory)" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';' | cut -c1-400; echo; }
smi
for i in 1 2 3 4 5 6 7 8; do
  timeout 100 python bench_fir.py --steps 300 --no-cpu-baseline 2>/dev/null | python -c "import sys,json,time; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d['roofline']; print('run $i  %-12s kernel %.4f ms frac %.4f' % (r['kernel'], r['kernel_avg_ms'], r['frac']))"
  smi
done
