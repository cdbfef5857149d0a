#!/bin/bash
# tools/pmc_quick.sh <outdir> <kernel-substring> "<counter set 1>" "<counter set 2>" ... -- <command...>   (few targeted PMC passes)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=$1; kern=$2; shift 2
sets=()
while [ "$1" != "--" ]; do sets+=("$1"); shift; done
shift
mkdir -p $out; i=0
for set in "${sets[@]}"; do i=$((i+1)); timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- "$@" > $out/p$i.log 2>&1; done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "$kern" not in r["Kernel_Name"]: continue
        agg[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()): print("   %-40s mean %.4g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
