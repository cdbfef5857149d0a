#!/bin/bash
# tools/ab3_wfm.sh [runs] [steps] -- bench.py interleaved between this tree's library, csdr_amd/libcsdr_amd_alt.so and csdr_amd/libcsdr_amd_old.so (when present) on ONE
# box; prints every run's ms per step and the medians (runs of one build differ by up to +-6 % on one box -- each process gets other physical pages for its 4.9 GB of
# input --, boxes by more: only medians of interleaved runs compare)
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
runs=${1:-7}; steps=${2:-100}
for i in $(seq $runs); do
  for v in new alt old; do
    lib=csdr_amd/libcsdr_amd_$v.so; [ $v = new ] && lib=csdr_amd/libcsdr_amd.so
    [ -f $lib ] && CSDR_AMD_LIB=$PWD/$lib python bench.py --no-cpu-baseline --steps $steps --warmup 5 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
  done
done | tee /tmp/ab3.txt
python - <<'PY'
import statistics as st
r = {}
for l in open('/tmp/ab3.txt'):
    v, a, b = l.split(); r.setdefault(v, []).append((float(a), float(b)))
for v, x in r.items():
    print("median %s: step %.4f kernel %.4f  (min %.4f max %.4f, %d runs)" % (v, st.median(a for a, _ in x), st.median(b for _, b in x), min(a for a, _ in x), max(a for a, _ in x), len(x)))
PY
