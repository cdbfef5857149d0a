#!/usr/bin/env python3
"""tools/diag_wfm_life.py <streams> <samples> [distinct] -- a workgroup's life in k_wfm_mfma_seq (library built with -DWFM_PROF=1, see tools/diag_wfm.py): shader-clock
cycles from kernel entry to the first step, inside the step loop and behind it, per wave, averaged over the workgroups of a call."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
import csdr_amd

S, T = int(sys.argv[1]), int(sys.argv[2]); distinct = len(sys.argv) > 3 and sys.argv[3] == "distinct"
steps = 40
ctx = csdr_amd.Context(0)
L = ctx.L
taps = ctx.firdes_lowpass_f(ctx.firdes_filter_len(0.05), 0.05, "HAMMING")
x = torch.randint(0, 256, (S, 2 * T), dtype=torch.uint8, device="cuda")
n_max = (T // 50 + 64 + 63) // 64 * 64
out = torch.empty((S, n_max), dtype=torch.int16, device="cuda")
if distinct:
    rates = (-0.45 + 0.9 * (np.arange(S) + 0.5) / S).astype(np.float32)
    w = L.csdr_amd_wfm_create_rates(ctx.h, S, rates.ctypes.data_as(C.c_void_p), 10, taps.ctypes.data_as(C.c_void_p), taps.size, 5, 50e-6, 48000, T)
else:
    w = L.csdr_amd_wfm_create(ctx.h, S, -0.085, 10, taps.ctypes.data_as(C.c_void_p), taps.size, 5, 50e-6, 48000, T)
for _ in range(20):
    L.csdr_amd_wfm_process(w, x.data_ptr(), 2 * T, T, out.data_ptr(), None, n_max)
ctx.sync()
life = (C.c_ulonglong * 32)()
L.csdr_amd_debug_wfm_life.argtypes = [C.c_void_p, C.c_int]
L.csdr_amd_debug_wfm_life(life, 1)
t0 = time.perf_counter()
for _ in range(steps):
    L.csdr_amd_wfm_process(w, x.data_ptr(), 2 * T, T, out.data_ptr(), None, n_max)
ctx.sync()
ms = (time.perf_counter() - t0) / steps * 1e3
L.csdr_amd_debug_wfm_life(life, 0)
print("%d streams x %d samples%s: %.4f ms per call (instrumented), %s" % (S, T, " (a rate per stream)" if distinct else "", ms, L.csdr_amd_wfm_kernel_name(w).decode()))
print("wave   before the first step   step loop   behind it   workgroups per call   (cycles per workgroup)")
for wv in range(8):
    n = max(life[wv * 4 + 3], 1)
    print("%4d %18.0f %14.0f %11.0f %14.1f" % (wv, life[wv * 4] / n, life[wv * 4 + 1] / n, life[wv * 4 + 2] / n, life[wv * 4 + 3] / steps))
