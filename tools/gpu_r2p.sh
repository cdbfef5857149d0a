#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  CSDR_AMD_WFM_SIDE=0 timeout 200 python bench.py --steps 300 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('side off', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
  timeout 200 python bench.py --steps 300 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('side on ', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
done
