#!/usr/bin/env python3
"""tools/probes/fir_realloc.py -- config 1's k_fir_poly runs at one of two levels per PROCESS (0.97 / 1.07 ms).  Does the level follow the ALLOCATION?  One process frees and
re-allocates its buffers (hipMalloc / hipFree directly, junk allocations of changing size in between so that the next ones land elsewhere) and times the same call each time."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import csdr_amd
ctx = csdr_amd.Context(0); L = ctx.L
hip = C.CDLL("libamdhip64.so")
S, T, D = 256, 2344 * 1024, 10
nt = ctx.firdes_filter_len(0.05); taps = ctx.upload(ctx.firdes_lowpass_f(nt, 0.5 / D, "HAMMING"))
opitch = T // D + 8
nb_in, nb_out = S * T * 8, S * opitch * 8
def malloc(n):
    p = C.c_void_p(); rc = hip.hipMalloc(C.byref(p), C.c_size_t(n)); assert rc == 0, rc; return p
junk = []
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    px, py = malloc(nb_in), malloc(nb_out)
    hip.hipMemset(px, 0x3c, C.c_size_t(nb_in)); hip.hipDeviceSynchronize()
    def step():
        n = L.csdr_amd_fir_decimate_cc(ctx.h, px, py, S, T, T, opitch, D, taps.ptr, nt); assert n >= 0
    for _ in range(40): step()
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(60): step()
    ctx.sync(); ms = (time.perf_counter() - t0) / 60 * 1e3
    # the same buffers under a plain device-to-device copy of the input (read + write of 4.9 GB each)
    pz = malloc(nb_in)
    for _ in range(3): hip.hipMemcpyAsync(pz, px, C.c_size_t(nb_in), C.c_int(3), None)
    hip.hipDeviceSynchronize(); t0 = time.perf_counter()
    for _ in range(10): hip.hipMemcpyAsync(pz, px, C.c_size_t(nb_in), C.c_int(3), None)
    hip.hipDeviceSynchronize(); cms = (time.perf_counter() - t0) / 10 * 1e3
    print("trial %d  in %#x out %#x  k_fir_poly %.4f ms per call (%.3f of 8 TB/s)   copy of the input %.3f ms (%.2f TB/s read + write)" % (trial, px.value, py.value, ms, (nb_in + nb_out) / ms / 8e9, cms, 2 * nb_in / cms / 1e9), flush=True)
    hip.hipFree(px); hip.hipFree(py); hip.hipFree(pz)
    junk.append(malloc((37 + 61 * trial) << 20))          # shifts what the next pair gets
    if len(junk) > 3: hip.hipFree(junk.pop(0))
