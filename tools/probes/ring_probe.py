"""tools/probes/ring_probe.py -- the resident ring's operating point under experiment switches (CSDR_AMD_RING_FENCE, CSDR_AMD_RING_GRID)."""
import sys, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch, numpy as np
import csdr_amd, bench
ctx = csdr_amd.Context(0)
taps = ctx.firdes_lowpass_f(ctx.firdes_filter_len(0.05), 0.5 / 10, "HAMMING")
for prio in ("1", "0", "1", "0"):
    os.environ["CSDR_AMD_RING_PRIO"] = prio
    e = bench.resident_point(ctx, taps, verify=False)
    print("prio", prio, {k: e.get(k) for k in ("kernel_avg_ms", "frac", "block_latency_us_median", "us_per_item_waiting_body_completion", "error")})
for slots in (8, 12, 16):
    e = bench.resident_point(ctx, taps, verify=False, n_slots=slots)
    print("slots", slots, {k: e.get(k) for k in ("kernel_avg_ms", "frac", "block_latency_us_median", "us_per_item_waiting_body_completion", "error")})
for grid in ("64", "128"):
    os.environ["CSDR_AMD_RING_GRID"] = grid
    e = bench.resident_point(ctx, taps, verify=False)
    print("grid", grid, {k: e.get(k) for k in ("kernel_avg_ms", "frac", "block_latency_us_median", "us_per_item_waiting_body_completion", "error")})
