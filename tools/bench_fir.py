#!/usr/bin/env python3
"""tools/bench_fir.py (fir_decimate_cc lines of bench_ops.py only; experiment helper)
tools/bench_ops.py -- per-operator throughput of the device batch API on one MI355X (HIP-event timed on the library's stream),
reported as algorithmic GB/s against the 8 TB/s HBM roofline.  Covers BASELINE configs[0] (single-block-size fir_decimate_cc,
batched), configs[2] (bandpass_fir_fft_cc @65536, taps sweep) and the stand-alone kernels of the chains.
Writes one JSON object per line; `python tools/bench_ops.py > profiles/rN_ops.jsonl` on the GPU box."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (one HIP runtime for the process)
import csdr_amd  # noqa: E402

ctx = csdr_amd.Context(0)
L = ctx.L
PEAK = 8000.0


def timeit(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    return ctx.timer_stop_ms() / reps


def report(name, ms, algo_bytes, samples, extra=None):
    gbs = algo_bytes / (ms * 1e-3) / 1e9
    rec = {"op": name, "ms": round(ms, 4), "algorithmic_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / PEAK, 4),
           "Msamples_per_s": round(samples / (ms * 1e-3) / 1e6, 1)}
    if extra:
        rec.update(extra)
    print(json.dumps(rec), flush=True)


def dev_rand_bytes(n):
    t = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    return t


def dev_rand_f32(n):
    t = (torch.rand((n,), device="cuda") * 2 - 1).contiguous()
    torch.cuda.synchronize()
    return t


N = 1 << 28
xf = dev_rand_f32(N); yf = torch.empty(N, dtype=torch.float32, device="cuda")
# ---- fir_decimate_cc, config C1 shape batched: decim 10, 79 taps (8.8 B per input sample); and NFM shape: decim 50, 801 taps
for (D, tbw, S2, n2) in [(10, 0.05, 1024, 16384), (10, 0.05, 64, 1 << 21), (50, 0.005, 64, 1 << 21)]:
    nt = ctx.firdes_filter_len(tbw)
    taps = ctx.upload(ctx.firdes_lowpass_f(nt, 0.5 / D))
    xi = xf[:2 * S2 * n2]
    op = n2 // D + 2
    yo = yf[:2 * S2 * op]
    ms = timeit(lambda: L.csdr_amd_fir_decimate_cc(ctx.h, xi.data_ptr(), yo.data_ptr(), S2, n2, n2, op, D, taps.ptr, nt))
    report("fir_decimate_cc D=%d taps=%d (%d streams x %d)" % (D, nt, S2, n2), ms, (8 + 8.0 / D) * S2 * n2, S2 * n2)

ctx.close()
