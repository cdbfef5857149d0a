#!/usr/bin/env python3
"""tools/probes/fftfilt_sizes.py -- the one-pass FFT filter at small call sizes (a CLI process: ONE stream, 64 blocks): which window kernel serves how many windows faster.
   CSDR_AMD_FFTFILT_LDS_MODE=5 (256- / 512-thread kernels) against the default; run both in one gpurun call."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import csdr_amd
ctx = csdr_amd.Context(0); L = ctx.L
FFT = 65536
rng = np.random.default_rng(1)
for ntaps in (1023, 4095):
    taps = ((rng.standard_normal(ntaps) + 1j * rng.standard_normal(ntaps)) / ntaps).astype(np.complex64)
    for S, NB in ((1, 16), (1, 64), (2, 64), (4, 64), (8, 64), (16, 64)):
        f = L.csdr_amd_fftfilt_create(ctx.h, FFT, taps.ctypes.data_as(C.c_void_p), ntaps, S, NB)
        inp = L.csdr_amd_fftfilt_input_size(f); n = NB * inp
        x = torch.randn((S, 2 * n), dtype=torch.float32, device="cuda"); y = torch.empty_like(x)
        for _ in range(20): L.csdr_amd_fftfilt_process(f, x.data_ptr(), y.data_ptr(), NB, n, n)
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(200): L.csdr_amd_fftfilt_process(f, x.data_ptr(), y.data_ptr(), NB, n, n)
        ctx.sync(); ms = (time.perf_counter() - t0) / 200 * 1e3
        win = L.csdr_amd_fftfilt_window(f); k1p = (ntaps - 1 + 15) & ~15
        print("taps %4d  %2d streams x %2d blocks = %5d windows  %-22s %.4f ms" % (ntaps, S, NB, S * -(-n // (win - k1p)), L.csdr_amd_fftfilt_kernel_name(f).decode(), ms))
        L.csdr_amd_fftfilt_destroy(f)
