"""tools/probes/ring_in_bench.py -- the resident ring's operating point in the order bench.py runs it (behind the three launch-per-call operating points), and again."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch, numpy as np
import csdr_amd, bench
ctx = csdr_amd.Context(0)
taps = ctx.firdes_lowpass_f(ctx.firdes_filter_len(0.05), 0.5 / 10, "HAMMING")
keys = ("kernel_avg_ms", "frac", "block_latency_us_median", "us_per_item_waiting_body_completion", "ms_per_step", "error")
e = bench.resident_point(ctx, taps, verify=False); print("first thing", {k: e.get(k) for k in keys})
pts = bench.operating_points(ctx, taps, verify=True)
print("operating points", [(p.get("streams"), p.get("block_samples_per_stream"), p.get("frac")) for p in pts])
for i in range(3):
    e = bench.resident_point(ctx, taps, verify=(i == 1)); print("behind them, run", i, {k: e.get(k) for k in keys})
e = bench.resident_point(ctx, taps, verify=False, n_slots=16); print("16 slots", {k: e.get(k) for k in keys})
