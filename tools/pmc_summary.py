#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; --output-format csv) + the bench JSON of the same workload into
profiles/<tag>_pmc_traffic.json.  usage: pmc_summary.py <run_dir with pmc_fetch/ pmc_write/ bench.json> <tag> <kernel substring>"""
import collections
import csv
import json
import os
import sys

run, tag, kern = sys.argv[1], sys.argv[2], sys.argv[3]
cmdline = sys.argv[4] if len(sys.argv) > 4 else "bench.py"
out = {}
for name, ctr in [("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")]:
    f = [x for x in os.listdir(os.path.join(run, name)) if x.endswith("counter_collection.csv")][0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(run, name, f))):
        if r["Counter_Name"] == ctr:
            nm = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("csdr_amd::", "")
            nm = nm.split("(")[0].strip() if not nm.startswith("(") else nm
            agg[nm].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out.setdefault(k, {})[ctr + "_KiB_mean"] = sum(v) / len(v)
        out[k][ctr + "_launches"] = len(v)
bench = json.load(open(os.path.join(run, "bench.json")))
# Which kernel(s) the summary is about.  A trailing "+" on the substring asks for the SUM over all matching kernels (several kernels of one step, e.g. the three
# passes of fft64k.hip).  Otherwise ONE kernel: the template instance the bench line itself names (roofline.kernel, e.g. k_fftfilt_lds<4096>) -- round 3 summed the
# <4096>, <8192> and <16384> instances of the sweep into one figure, 3.4 x the algorithmic bytes of the instance the line was about.
squash = lambda t: t.replace(" ", "")
want_sum = kern.endswith("+")
kern = kern.rstrip("+")
cands = [k for k in out if kern in k and "<true>" not in k]
if not cands:
    raise SystemExit("pmc_summary: no kernel matches %r among %s" % (kern, sorted(out)))
line_kernel = squash(str(bench.get("roofline", {}).get("kernel", "")).split(" (")[0])
if want_sum:
    keys = cands
else:
    exact = [k for k in cands if line_kernel and (squash(k) == line_kernel or squash(k).startswith(line_kernel + "<") or line_kernel.startswith(squash(k)))]
    keys = exact[:1] if exact else sorted(cands, key=lambda k: -out[k].get("FETCH_SIZE_launches", 0))[:1]
fetch = sum(out[k].get("FETCH_SIZE_KiB_mean", 0) * 1024 for k in keys)
write = sum(out[k].get("WRITE_SIZE_KiB_mean", 0) * 1024 for k in keys)
key = " + ".join(keys)
per_instance = {k: {"traffic_bytes_per_launch": 2 * out[k].get("FETCH_SIZE_KiB_mean", 0) * 1024 + out[k].get("WRITE_SIZE_KiB_mean", 0) * 1024,
                    "launches": out[k].get("FETCH_SIZE_launches", 0)} for k in cands}
algo = bench["roofline"]["algorithmic_bytes_per_launch"]
summary = {
    "tag": tag,
    "command": "rocprofv3 --pmc FETCH_SIZE (pass 1) / WRITE_SIZE (pass 2) --kernel-trace --output-format csv -- python %s --steps 2 --warmup 1 --no-cpu-baseline" % cmdline,
    "workload": bench["config"], "kernel": key,
    "FETCH_SIZE_bytes_raw": fetch, "WRITE_SIZE_bytes": write,
    "correction": "FETCH_SIZE x2 (MI355X_MICROARCH.md section HBM: gfx950 rocprofv3 reports 1/2 of coalesced streaming reads; cross-checked on k_wfm_back, "
                  "which reads a 196.6 MB float array and reports ~99.9 MB); WRITE_SIZE as is (torch randint: 1.2288 GB per call reported exactly)",
    "traffic_bytes_per_launch": 2 * fetch + write, "algorithmic_bytes_per_launch": algo,
    "traffic_over_algorithmic": (2 * fetch + write) / algo,
    "matching_kernels": per_instance,      # every kernel the substring matched, each on its own (other template instances of a sweep: compare with THEIR algorithmic bytes)
    "all_kernels": out,
}
json.dump(summary, open(os.path.join("profiles", tag + "_pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: summary[k] for k in ("kernel", "traffic_bytes_per_launch", "algorithmic_bytes_per_launch", "traffic_over_algorithmic")}))
