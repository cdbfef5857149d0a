#!/bin/bash
# tools/ab_nfm.sh [alt.so] -- A/B of bench_nfm.py between this tree's library and another build (see tools/ab_wfm.sh) on ONE box, interleaved
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
alt=${1:-csdr_amd/libcsdr_amd_alt.so}
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['kernel_avg_ms'])"; }
for i in 1 2 3; do
  python bench_nfm.py --no-cpu-baseline --steps 200 | pr new
  [ -f $alt ] && CSDR_AMD_LIB=$PWD/$alt python bench_nfm.py --no-cpu-baseline --steps 200 | pr alt
done
