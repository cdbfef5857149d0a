#!/usr/bin/env python3
"""tools/diag_wfm.py -- where a wave of k_wfm_mfma_seq spends its cycles (library built with -DWFM_PROF=1:
   make -C csdr_amd/csrc -j8 OBJDIR=build_p1 TARGET=../libcsdr_amd_p1.so EXTRA=-DWFM_PROF=1 ../libcsdr_amd_p1.so
   CSDR_AMD_LIB=$PWD/csdr_amd/libcsdr_amd_p1.so python tools/diag_wfm.py).  bench.py's workload; prints shader-clock cycles per step and wave."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import csdr_amd

S, T, steps = 1024, 2344 * 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 100
ctx = csdr_amd.Context(0)
L = ctx.L
taps = ctx.firdes_lowpass_f(ctx.firdes_filter_len(0.05), 0.05, "HAMMING")
x = torch.randint(0, 256, (S, 2 * T), dtype=torch.uint8, device="cuda")
n_max = (T // 50 + 64 + 63) // 64 * 64
out = torch.empty((S, n_max), dtype=torch.int16, device="cuda")
torch.cuda.synchronize()
if os.environ.get("DIAG_RATES") == "distinct":                                   # a shift rate per stream (the per-stream kernel)
    import numpy as np
    rates = (-0.45 + 0.9 * (np.arange(S) + 0.5) / S).astype(np.float32)
    w = L.csdr_amd_wfm_create_rates(ctx.h, S, rates.ctypes.data_as(C.c_void_p), 10, taps.ctypes.data_as(C.c_void_p), taps.size, 5, 50e-6, 48000, T)
else:
    w = L.csdr_amd_wfm_create(ctx.h, S, -0.085, 10, taps.ctypes.data_as(C.c_void_p), taps.size, 5, 50e-6, 48000, T)
for _ in range(60):
    L.csdr_amd_wfm_process(w, x.data_ptr(), 2 * T, T, out.data_ptr(), None, n_max)
ctx.sync()
prof = (C.c_ulonglong * 64)()
L.csdr_amd_debug_wfm_prof.argtypes = [C.c_void_p, C.c_int]
L.csdr_amd_debug_wfm_prof(prof, 1)
t0 = time.perf_counter()
for _ in range(steps):
    L.csdr_amd_wfm_process(w, x.data_ptr(), 2 * T, T, out.data_ptr(), None, n_max)
ctx.sync()
ms = (time.perf_counter() - t0) / steps * 1e3
L.csdr_amd_debug_wfm_prof(prof, 0)
names = ["compute", "wait_vm", "barrier", "dma_issue", "emit", "deemph", "steps"]
print("ms per call %.4f (instrumented)" % ms)
print("wave " + " ".join("%10s" % n for n in names[:6]) + "   total   (cycles per step, averaged over all workgroups)")
for wv in range(8):
    n = prof[wv * 8 + 6]
    row = [prof[wv * 8 + k] / max(n, 1) for k in range(6)]
    print("%4d " % wv + " ".join("%10.0f" % v for v in row) + " %8.0f" % sum(row) + "   steps per call %.0f, cycles before the first step (sum over workgroups, per call) %.0f" % (n / steps, prof[wv * 8 + 7] / steps))
