#!/bin/bash
# tools/ab_points.sh <rounds> <points, e.g. 0,2> <lib> [<lib> ...] -- bench.py's WFM operating points, interleaved between builds of the library (CSDR_AMD_LIB; "default" =
# csdr_amd/libcsdr_amd.so; "default:VAR=v" adds an environment variable), plus the headline (bench.py --no-other-configs) once per build and round
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
rounds=$1; pts=$2; shift 2
for i in $(seq 1 $rounds); do
  for spec in "$@"; do
    lib=${spec%%:*}; extra=; [ "$spec" != "$lib" ] && extra=${spec#*:}
    if [ "$lib" = default ]; then unset CSDR_AMD_LIB; else export CSDR_AMD_LIB=$PWD/csdr_amd/$lib; fi
    env $extra POINTS=$pts TAG="$spec" timeout 100 python tools/bench_wfm_points.py 2>/dev/null | grep streams
    env $extra timeout 100 python bench.py --no-cpu-baseline --no-other-configs --no-verify --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-28s headline ms/step %.4f kernel %.4f' % ('$spec', d['ms_per_step'], d['roofline']['kernel_avg_ms']))"
  done
done
