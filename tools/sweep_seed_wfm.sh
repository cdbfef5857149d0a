#!/bin/bash
# tools/sweep_seed_wfm.sh -- per-stream WFM operating point against where the seed generator's waves sit (CSDR_AMD_SEED_BLOCK lanes per workgroup) and how long the
# per-stream kernel's columns are (CSDR_AMD_WFM_PS_SPLIT); CSDR_AMD_SEED_FREEZE = the generator off (timing only).  Interleaved, two rounds.
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
run() { env POINTS=0 "$@" TAG="$*" timeout 100 python tools/bench_wfm_points.py 2>/dev/null | grep -m1 "x  2400256\|x 2400256"; }
for i in 1 2; do
  for cfg in "${@:-A=0}"; do run $cfg; done
done
