#!/bin/bash
# tools/probes/spinup_sweep.sh -- the driver's shape (20 timed steps after idle) against the length of the untimed spin-up in front: ms per step, three interleaved rounds
cd $GRAFT_REPO_ROOT 2>/dev/null
for round in 1 2 3; do
  for ms in 0 60 300 1000; do
    r=$(timeout 100 python bench.py --steps 20 --warmup 5 --spinup-ms $ms --no-cpu-baseline --no-other-configs --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['spinup_steps_before_warmup'])")
    echo "round $round spinup_ms $ms: $r"
    sleep 1
  done
done
