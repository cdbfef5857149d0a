#!/usr/bin/env python3
"""tools/bench_wfm_points.py -- bench.py's operating points of the WFM chain on their own (per-stream rates at the headline size, 1024 x 16384, 65536 x 24576), no verify:
for A/B runs.  One line per point."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import csdr_amd, bench
ctx = csdr_amd.Context(0)
taps = ctx.firdes_lowpass_f(ctx.firdes_filter_len(0.05), 0.5 / 10, "HAMMING")
for p in bench.operating_points(ctx, taps, verify=False, only=[int(v) for v in os.environ["POINTS"].split(",")] if os.environ.get("POINTS") else None):
    print("%-8s streams %6d x %8d (%s): ms/step %.4f kernel %.4f frac %.4f" % (os.environ.get("TAG", ""), p["streams"], p["block_samples_per_stream"], p["shift_rates"][:10], p["ms_per_step"], p["kernel_avg_ms"], p["frac"]))
ctx.close()
