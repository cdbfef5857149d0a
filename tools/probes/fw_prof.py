#!/usr/bin/env python3
"""tools/probes/fw_prof.py -- where a wave of k_fftfilt_wave spends a window (library built with -DFW_PROF:
   make -C csdr_amd/csrc -j8 OBJDIR=build_p1 TARGET=../libcsdr_amd_p1.so EXTRA=-DFW_PROF=1 ../libcsdr_amd_p1.so
   CSDR_AMD_LIB=$PWD/csdr_amd/libcsdr_amd_p1.so CSDR_AMD_FFTFILT_LDS_MODE=6 python tools/probes/fw_prof.py [taps]).  bench_fftfilt.py's workload (64 streams x 16 blocks)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import csdr_amd
ntaps = int(sys.argv[1]) if len(sys.argv) > 1 else 1023
S, NB, FFT = 64, 16, 65536
ctx = csdr_amd.Context(0); L = ctx.L
rng = np.random.default_rng(1)
taps = ((rng.standard_normal(ntaps) + 1j * rng.standard_normal(ntaps)) / ntaps).astype(np.complex64)
f = L.csdr_amd_fftfilt_create(ctx.h, FFT, taps.ctypes.data_as(C.c_void_p), ntaps, S, NB)
inp = L.csdr_amd_fftfilt_input_size(f); n = NB * inp
x = torch.randn((S, 2 * n), dtype=torch.float32, device="cuda"); y = torch.empty_like(x)
print("kernel", L.csdr_amd_fftfilt_kernel_name(f).decode(), "window", L.csdr_amd_fftfilt_window(f))
for _ in range(10): L.csdr_amd_fftfilt_process(f, x.data_ptr(), y.data_ptr(), NB, n, n)
ctx.sync()
prof = (C.c_ulonglong * 8)()
have = hasattr(L, "csdr_amd_debug_fw_prof")
if have:
    L.csdr_amd_debug_fw_prof.argtypes = [C.c_void_p, C.c_int]; L.csdr_amd_debug_fw_prof(prof, 1)
steps = 50
t0 = time.perf_counter()
for _ in range(steps): L.csdr_amd_fftfilt_process(f, x.data_ptr(), y.data_ptr(), NB, n, n)
ctx.sync()
print("ms per step %.4f" % ((time.perf_counter() - t0) / steps * 1e3))
if have:
    L.csdr_amd_debug_fw_prof(prof, 0)
    names = ["input wait", "pass 0", "transpose 1", "pass 1 (+H)", "pass 2", "transpose 2", "pass 3"]
    nwin = prof[7]
    tot = sum(prof[k] for k in range(7))
    for k in range(7): print("%-14s %7.2f us per window" % (names[k], prof[k] / nwin / 100.0))
    print("%-14s %7.2f us per window (%d windows of wave 0 of every workgroup, %d steps; stores: inside the next window's input wait)" % ("sum", tot / nwin / 100.0, nwin, steps))
