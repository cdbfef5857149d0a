#!/bin/bash
# NFM chain: demodulator + limiter + digit planes fused into the front end's reducer (A/B with CSDR_AMD_NFM_FUSE=0) + parity tests
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -x -m gpu -k "nfm or ddc or agc or am or ssb or chain or golden" 2>&1 | grep -E "passed|failed|Error|error" | tail -5
for rep in 1 2; do
  CSDR_AMD_NFM_FUSE=0 timeout 200 python bench_nfm.py --steps 100 --no-cpu-baseline --verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('unfused', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['verify'])" | cut -c1-250
  timeout 200 python bench_nfm.py --steps 100 --no-cpu-baseline --verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused  ', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['verify'])" | cut -c1-250
done
