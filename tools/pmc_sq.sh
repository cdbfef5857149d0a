#!/bin/bash
# tools/pmc_sq.sh <tag> <kernel substring> <counter list, comma separated per pass; passes separated by ':'> -- <command...>
# SQ-side counter passes of one command (rocprofv3 --pmc with --kernel-trace only), summed per kernel over the dispatches that match.
tag=$1; kern=$2; passes=$3; shift 4
root=$GRAFT_REPO_ROOT; [ -z "$root" ] && root=$PWD
out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $root
IFS=':' read -ra P <<< "$passes"
i=0
for p in "${P[@]}"; do
  timeout 300 rocprofv3 --pmc ${p//,/ } --kernel-trace --output-format csv -d $out/p$i -- "$@" > $out/p$i.log 2>&1
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$kern" <<'PYEOF'
import csv, sys
from collections import defaultdict
acc = defaultdict(float); n = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(acc): print("%-28s %16.0f per dispatch (%d dispatches)" % (k, acc[k] / max(len(n[k]), 1), len(n[k])))
PYEOF
  i=$((i+1))
done
