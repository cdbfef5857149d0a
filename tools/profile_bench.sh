#!/bin/bash
# tools/profile_bench.sh <tag> <kernel substring> <bench script> [bench args...] -- on the GPU box, in THIS order (round 5: the measured line is generated LAST, so that
# its roofline.traffic can only come from the PMC passes of the same run -- VERDICT r4 weak #9):
#   1. rocprofv3 --kernel-trace --stats of the bench (120 steps)            -> <tag>_kernel_stats.csv; the run's own JSON line names the kernel and its algorithmic bytes
#   2. two PMC passes (FETCH_SIZE / WRITE_SIZE, each in its own run, counters + --kernel-trace only) -> tools/pmc_summary.py -> profiles/<tag>_pmc_traffic.json
#   3. the bench itself with --verify and the CPU baseline                   -> <tag>_n1.json (traffic_source = the file of step 2)
# Raw output under gpurun_out/<tag>/, summaries in gpurun_out/profiles_<tag>/ (copy them into profiles/ and commit).  Every run under its own timeout.
tag=$1; kern=$2; script=$3; shift 3
[ "$script" = "bench.py" ] && set -- "$@" --no-other-configs      # the driver line's extra legs are child processes of their own: not part of this profile
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$PWD
out=$root/gpurun_out/$tag; mkdir -p $out $root/gpurun_out/profiles_$tag
cd /tmp && export TMPDIR=/tmp; cd $root
T=${PROFILE_TIMEOUT:-150}
timeout $T rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $script "$@" --steps 120 --warmup 5 --no-cpu-baseline > $out/trace.log 2> $out/trace.err
timeout $T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pf -- python $script "$@" --steps 2 --warmup 1 --no-cpu-baseline > $out/pf.log 2>&1
timeout $T rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pw -- python $script "$@" --steps 2 --warmup 1 --no-cpu-baseline > $out/pw.log 2>&1
mkdir -p $out/run/pmc_fetch $out/run/pmc_write
cp $(find $out/pf -name "*counter_collection.csv" | head -1) $out/run/pmc_fetch/ 2>/dev/null
cp $(find $out/pw -name "*counter_collection.csv" | head -1) $out/run/pmc_write/ 2>/dev/null
grep '^{' $out/trace.log | tail -1 > $out/run/bench.json
(cd $root && python tools/pmc_summary.py $out/run $tag "$kern" "$script $*" && cp profiles/${tag}_pmc_traffic.json gpurun_out/profiles_$tag/)
ks=$(find $out/trace -name "*kernel_stats.csv" | head -1)
[ -n "$ks" ] && python tools/tidy_kernel_stats.py $ks $root/gpurun_out/profiles_$tag/${tag}_kernel_stats.csv "$tag: rocprofv3 --kernel-trace --stats -- python $script $* --steps 120 --warmup 5 (durations in ns)"
vflag="--verify"; [ "$script" = "bench.py" ] && vflag=""          # (bench.py verifies by default)
timeout ${LINE_TIMEOUT:-240} python $script "$@" $vflag > $out/n1.log 2> $out/n1.err
grep '^{' $out/n1.log | tail -1 > $root/gpurun_out/profiles_$tag/${tag}_n1.json
head -6 $root/gpurun_out/profiles_$tag/${tag}_kernel_stats.csv | cut -c1-200
python - $root/gpurun_out/profiles_$tag/${tag}_n1.json <<'PYEOF'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d.get("roofline", {})
    print("[n1] %s %s ms/step %s | kernel %s avg %s ms frac %s | traffic %s (%s) | verify %s" % (d.get("value"), d.get("unit"), d.get("ms_per_step"), r.get("kernel"), r.get("kernel_avg_ms"), r.get("frac"),
          r.get("traffic"), str(r.get("traffic_source"))[:40], d.get("verify", {}).get("ok")))
except Exception as e:
    print("[n1] no line:", e)
PYEOF
