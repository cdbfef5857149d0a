#!/bin/bash
# tools/probes/clock_under_load.sh -- the shader clock a kernel actually ran at: GRBM_GUI_ACTIVE (cycles) over the dispatch's duration (kernel trace), for the FFT-filter
# kernels and for a compute-only kernel (tools/probes/dft64_rate)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/clk; mkdir -p $out
run() { # name, command...
  local name=$1; shift
  timeout 150 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/$name -- "$@" > $out/$name.log 2>&1
  python - $out/$name <<'PYEOF'
import csv, glob, sys, collections
d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True); kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
if not cc or not kt: print(d, "no output"); sys.exit()
dur = {}
for r in csv.DictReader(open(kt[0])): dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(cc[0])):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur: continue
    name, ns = dur[r["Dispatch_Id"]]
    if ns < 50000: continue
    a = acc[name.split("(")[0][-60:]]; a[0] += float(r["Counter_Value"]); a[1] += ns; a[2] += 1
for k, (cyc, ns, n) in acc.items(): print("%-62s %3d dispatches, %8.1f us each, GRBM_GUI_ACTIVE / ns = %.3f" % (k, n, ns / n / 1e3, cyc / ns))
PYEOF
}
run wave python bench_fftfilt.py --steps 20 --warmup 2 --no-sweep --taps 1023 --no-cpu-baseline
run dft tools/probes/dft64_rate
run wfm python bench.py --steps 10 --warmup 2 --no-other-configs --no-cpu-baseline
