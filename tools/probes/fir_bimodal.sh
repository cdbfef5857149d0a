#!/bin/bash
# tools/probes/fir_bimodal.sh [runs] -- config 1's k_fir_poly runs at one of two levels per PROCESS (profiles/r5_fir_c1_bimodal_runs.txt).  N processes of bench_fir.py under
# rocprofv3 with L2 / fabric stall counters: duration of the kernel against the counters, per process.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/firbi; mkdir -p $out
P=${PMC:-TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TAG_STALL TCC_EA0_WRREQ_STALL}
for i in $(seq 1 ${1:-8}); do
  timeout 120 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $out/r$i -- python bench_fir.py --steps 30 --warmup 3 --no-cpu-baseline > $out/r$i.log 2>&1
  python - $out/r$i $i <<'PYEOF'
import csv, glob, sys, collections
d, i = sys.argv[1], sys.argv[2]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True); kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
if not cc or not kt: print("run", i, "no output"); sys.exit()
dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0])) if "k_fir_poly" in r["Kernel_Name"]}
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(cc[0])):
    if r["Dispatch_Id"] in dur: acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
ds = sorted(dur.values())
if "GRBM_GUI_ACTIVE" in acc: acc["shader GHz (GRBM_GUI_ACTIVE / 8 XCDs / ns)"] = acc.pop("GRBM_GUI_ACTIVE") / 8 / (sum(dur.values()) / len(dur)); n["shader GHz (GRBM_GUI_ACTIVE / 8 XCDs / ns)"] = n.pop("GRBM_GUI_ACTIVE")
print("run %s  k_fir_poly median %.1f us (%d launches)  " % (i, ds[len(ds) // 2] / 1e3, len(ds)) + "  ".join("%s %.3g" % (k.replace("TCC_", ""), acc[k] / max(n[k], 1)) for k in sorted(acc)))
PYEOF
done
