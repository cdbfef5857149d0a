"""repro: WFM per-stream chain in 65536-sample calls, every call's audio at the start of the (aligned) output rows, s16 only (the bank CLI's case)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ctypes as C
import csdr_amd
import oracle
from tests_helpers import wfm_signal_u8
port = oracle.port()
gpu = csdr_amd.Context(0); L = gpu.L
n = 3 * 65536 + 5 * 1024
eff = [-0.05, -0.2, 0.1234]
sig = np.stack([wfm_signal_u8(950 + k, n, offset=-eff[k]) for k in range(3)])
taps = port.firdes_lowpass_f(79, 0.05)
T = 65536
for per_stream in (False, True):
    for wf in (True, False):
        rates = np.array(eff, np.float32)
        if per_stream:
            w = L.csdr_amd_wfm_create_rates(gpu.h, 3, rates.ctypes.data_as(C.c_void_p), 10, taps.ctypes.data_as(C.c_void_p), taps.size, 5, 50e-6, 48000, T)
        else:
            w = L.csdr_amd_wfm_create(gpu.h, 3, eff[0], 10, taps.ctypes.data_as(C.c_void_p), taps.size, 5, 50e-6, 48000, T)
        di = gpu.alloc(3 * 2 * T + 256); apitch = ((T // 50 + 4096 + 63) // 64) * 64
        ds = gpu.alloc(2 * 3 * apitch + 256); df = gpu.alloc(4 * 3 * apitch + 256)
        outs = [[], [], []]; pos = 0
        while pos < n:
            k = min(T, n - pos)
            blk = np.full((3, 2 * T), 0x80, np.uint8); blk[:, :2 * k] = sig[:, 2 * pos:2 * pos + 2 * k]
            gpu.upload_into(di, blk) if hasattr(gpu, "upload_into") else L.csdr_amd_h2d(gpu.h, di.ptr, blk.ctypes.data_as(C.c_void_p), blk.size)
            na = L.csdr_amd_wfm_process(w, di.ptr, 2 * T, k, ds.ptr, df.ptr if wf else None, apitch)
            y = gpu.download(ds, np.int16, 3 * apitch).reshape(3, apitch)
            for s in range(3): outs[s].append(y[s, :na].copy())
            pos += k
        L.csdr_amd_wfm_destroy(w)
        for s in range(3 if per_stream else 1):
            got = np.concatenate(outs[s]); want, _ = port.wfm_chain(sig[s], eff[s], 10, taps)
            m = min(got.size, want.size)
            d = np.abs(got[:m].astype(np.int32) - want[:m].astype(np.int32))
            bad = np.nonzero(d > 1)[0]
            print("per_stream", per_stream, "float", wf, "stream", s, got.size, want.size, "bad", bad.size, bad[:10], bad[-3:] if bad.size else "", [o.size for o in outs[s]])
