#!/usr/bin/env python3
"""tools/diag_nfm.py -- where a wave of k_ddc_mfma (NFM chain, fused epilogue) spends its cycles (library built with -DDDC_PROF=1:
   make -C csdr_amd/csrc -j8 OBJDIR=build_p1 TARGET=../libcsdr_amd_p1.so EXTRA=-DDDC_PROF=1 ../libcsdr_amd_p1.so
   CSDR_AMD_LIB=$PWD/csdr_amd/libcsdr_amd_p1.so python tools/diag_nfm.py).  bench_nfm.py's workload; prints shader-clock cycles per group and wave."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import csdr_amd

S, T, D, steps = 512, 2344 * 1024, 50, int(sys.argv[1]) if len(sys.argv) > 1 else 100
ctx = csdr_amd.Context(0)
L = ctx.L
taps = ctx.firdes_lowpass_f(ctx.firdes_filter_len(0.005), 0.5 / D, "HAMMING")
x = torch.randint(0, 256, (S, 2 * T), dtype=torch.uint8, device="cuda")
n_max = (T // D + 2048 + 63) // 64 * 64
out = torch.empty((S, n_max), dtype=torch.int16, device="cuda")
torch.cuda.synchronize()
if os.environ.get("DIAG_RATES", "uniform") == "uniform":
    obj = L.csdr_amd_nfm_create(ctx.h, S, -0.05, D, taps.ctypes.data_as(C.c_void_p), taps.size, 48000, 1024, 1.0, 1.0, T)
else:                                                     # DIAG_RATES=distinct: a rate per channel (the per-stream kernel)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import verify_configs as vc
    rates = vc.c5_rates(S)
    obj = L.csdr_amd_nfm_create_rates(ctx.h, S, rates.ctypes.data_as(C.c_void_p), D, taps.ctypes.data_as(C.c_void_p), taps.size, 48000, 1024, 1.0, 1.0, T)
for _ in range(60):
    L.csdr_amd_nfm_process(obj, x.data_ptr(), 2 * T, T, out.data_ptr(), None, n_max)
ctx.sync()
prof = (C.c_ulonglong * 64)()
L.csdr_amd_debug_ddc_prof.argtypes = [C.c_void_p, C.c_int]
L.csdr_amd_debug_ddc_prof(prof, 1)
t0 = time.perf_counter()
for _ in range(steps):
    L.csdr_amd_nfm_process(obj, x.data_ptr(), 2 * T, T, out.data_ptr(), None, n_max)
ctx.sync()
ms = (time.perf_counter() - t0) / steps * 1e3
L.csdr_amd_debug_ddc_prof(prof, 0)
names = ["compute", "wait_vm", "barrier", "dma_issue", "epilogue"]
print("ms per call %.4f (whole chain, instrumented front end)" % ms)
print("wave " + " ".join("%10s" % n for n in names) + "   total   (cycles per group of 2 tiles, averaged over all workgroups)")
for wv in range(8):
    n = prof[wv * 8 + 5]
    row = [prof[wv * 8 + k] / max(n, 1) for k in range(5)]
    print("%4d " % wv + " ".join("%10.0f" % v for v in row) + " %8.0f" % sum(row))
