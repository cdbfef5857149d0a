#!/bin/bash
# tools/sweep_nfm.sh VAR v1 v2 ... -- bench_nfm.py --front-end-only once per value of VAR: value / ms_per_step / kernel ms / frac
var=$1; shift
for v in "$@"; do
  env "$var=$v" timeout 120 python bench_nfm.py --front-end-only $NFM_ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('$var=$v', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"
done
