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
  timeout ${PMC_TIMEOUT:-150} rocprofv3 --pmc ${p//,/ } --kernel-trace --output-format csv -d $out/p$i -- "$@" > $out/p$i.log 2>&1
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$kern" "$out/summary.json" "$p" "$*" <<'PYEOF'
import csv, json, os, sys
from collections import defaultdict
acc = defaultdict(float); n = defaultdict(set); names = set()
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"]); names.add(r["Kernel_Name"].split("(")[0])
for k in sorted(acc): print("%-28s %16.0f per dispatch (%d dispatches)" % (k, acc[k] / max(len(n[k]), 1), len(n[k])))
# one JSON per tag, a pass per key (copy it to profiles/<tag>_pmc_issue.json)
path = sys.argv[3]
d = json.load(open(path)) if os.path.exists(path) else {"kernel_substring": sys.argv[2], "command": "rocprofv3 --pmc <pass> --kernel-trace --output-format csv -- " + sys.argv[5],
                                                         "note": "values are sums over all SEs / XCDs, averaged per dispatch of the matching kernel(s); one rocprofv3 run per pass", "passes": {}}
d["kernels_matched"] = sorted(set(d.get("kernels_matched", [])) | names)
d["passes"][sys.argv[4]] = {k: {"per_dispatch": acc[k] / max(len(n[k]), 1), "dispatches": len(n[k])} for k in sorted(acc)}
json.dump(d, open(path, "w"), indent=1)
PYEOF
  i=$((i+1))
done
