#!/bin/bash
# tools/ab_libs.sh <rounds> <bench script + args, quoted> <lib> [<lib> ...] -- interleaved A/B of one bench between builds of the library on ONE box
# (CSDR_AMD_LIB selects the build; "default" = csdr_amd/libcsdr_amd.so).  Prints ms per step and the dominant kernel's average per run.
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
rounds=$1; cmd=$2; shift 2
for i in $(seq 1 $rounds); do
  for lib in "$@"; do
    if [ "$lib" = default ]; then unset CSDR_AMD_LIB; else export CSDR_AMD_LIB=$PWD/csdr_amd/$lib; fi
    timeout 120 python $cmd --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d['roofline']; print('%-28s ms/step %.4f  kernel %.4f ms  frac %.4f  %s' % ('$lib', d['ms_per_step'], r['kernel_avg_ms'], r['frac'], r['kernel'][:40]))"
  done
done
