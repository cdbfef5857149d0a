#!/bin/bash
# tools/sweep_env.sh VAR v1 v2 ... -- runs bench.py (no CPU baseline) once per value of the environment variable VAR and prints
# value / ms_per_step / kernel_avg_ms / frac per run (experiment helper for the GPU box).
var=$1; shift
for v in "$@"; do
  env "$var=$v" timeout 120 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$var=$v', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"
done
