#!/usr/bin/env python3
"""tools/bench_ops.py -- per-operator throughput of the device batch API on one MI355X (HIP-event timed on the library's stream),
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


N = 1 << 28    # 268 M values
# ---- converters
x8 = dev_rand_bytes(N); xf = dev_rand_f32(N); yf = torch.empty(N, dtype=torch.float32, device="cuda"); y16 = torch.empty(N, dtype=torch.int16, device="cuda")
report("convert_u8_f", timeit(lambda: L.csdr_amd_convert_u8_f(ctx.h, x8.data_ptr(), yf.data_ptr(), N)), 5 * N, N)
report("convert_f_s16", timeit(lambda: L.csdr_amd_convert_f_s16(ctx.h, xf.data_ptr(), y16.data_ptr(), N)), 6 * N, N)
report("limit_ff", timeit(lambda: L.csdr_amd_limit_ff(ctx.h, xf.data_ptr(), yf.data_ptr(), N, 1.0)), 8 * N, N)
del x8, y16

# ---- shift_addition_cc: 64 streams x 2^21 samples (rotator table shared)
S, n = 64, 1 << 21
xin = xf[:2 * S * n]; yout = yf[:2 * S * n]
ph = C.c_float(0.0)
report("shift_addition_cc (64 streams x 2M, gen+mix)", timeit(lambda: L.csdr_amd_shift_cc(ctx.h, 0, -0.085, C.byref(ph), xin.data_ptr(), yout.data_ptr(), S, n, n, n, 1024, 0)),
       16 * S * n, S * n)
for vname, variant, aux in (("shift_math_cc", 1, 0), ("shift_table_cc", 2, 65536), ("shift_unroll_cc", 3, 1024), ("shift_addfast_cc", 4, 0)):
    report("%s (64 streams x 2M, gen+mix; math / table: the host's sequential float phase scan -- run ahead on a helper thread -- bounds a tight loop of calls)" % vname,
           timeit(lambda: L.csdr_amd_shift_cc(ctx.h, variant, -0.085, C.byref(ph), xin.data_ptr(), yout.data_ptr(), S, n, n, n, 1024, aux), reps=5, warm=1), 16 * S * n, S * n,
           {"bound": "serial: the reference's float phase recurrence is one sequential chain per call (math / table: one phase add + wrap per SAMPLE, on the host); the HBM roofline does not apply"}
           if variant in (1, 2) else None)
rot = torch.empty(2 * n + 16, dtype=torch.float32, device="cuda")
L.csdr_amd_rotator_generate(ctx.h, 0, -0.085, C.byref(ph), rot.data_ptr(), n, 1024, 0)
report("mix_cc only (64 streams x 2M)", timeit(lambda: L.csdr_amd_mix_cc(ctx.h, xin.data_ptr(), yout.data_ptr(), rot.data_ptr(), S, n, n, n)), 16 * S * n, S * n)

# ---- fir_decimate_cc, config C1 shape batched: decim 10, 79 taps (8.8 B per input sample); and NFM shape: decim 50, 801 taps
for (D, tbw, S2, n2) in [(10, 0.05, 1024, 16384), (10, 0.05, 64, 1 << 21), (50, 0.005, 64, 1 << 21)]:
    nt = ctx.firdes_filter_len(tbw)
    taps = ctx.upload(ctx.firdes_lowpass_f(nt, 0.5 / D))
    xi = xf[:2 * S2 * n2]
    op = n2 // D + 2
    yo = yf[:2 * S2 * op]
    ms = timeit(lambda: L.csdr_amd_fir_decimate_cc(ctx.h, xi.data_ptr(), yo.data_ptr(), S2, n2, n2, op, D, taps.ptr, nt))
    report("fir_decimate_cc D=%d taps=%d (%d streams x %d)" % (D, nt, S2, n2), ms, (8 + 8.0 / D) * S2 * n2, S2 * n2)

# ---- fmdemod_quadri_cf
S3, n3 = 256, 240000
last = ctx.upload(np.zeros(S3, np.complex64))
report("fmdemod_quadri_cf (256 x 240k)", timeit(lambda: L.csdr_amd_fmdemod_quadri_cf(ctx.h, xf.data_ptr(), yf.data_ptr(), S3, n3, n3, n3, last.ptr)), 12 * S3 * n3, S3 * n3)

# ---- bandpass_fir_fft_cc @65536 (config C3), 64 blocks, taps sweep
for ntaps in [63, 255, 1023, 4095]:
    taps = ctx.firdes_bandpass_c(ntaps, -0.1, 0.2)
    nb = 64
    f = L.csdr_amd_fftfilt_create(ctx.h, 65536, taps.ctypes.data_as(C.c_void_p), ntaps, 1, nb)
    inp = L.csdr_amd_fftfilt_input_size(f)
    xi = xf[:2 * nb * inp]; yo = yf[:2 * nb * inp]
    ms = timeit(lambda: L.csdr_amd_fftfilt_process(f, xi.data_ptr(), yo.data_ptr(), nb, nb * inp, nb * inp))
    report("bandpass_fir_fft_cc fft=65536 taps=%d (64 blocks, 1 stream)" % ntaps, ms, 16 * nb * inp, nb * inp)
    L.csdr_amd_fftfilt_destroy(f)
nb, S4 = 16, 64
taps = ctx.firdes_bandpass_c(1023, -0.1, 0.2)
f = L.csdr_amd_fftfilt_create(ctx.h, 65536, taps.ctypes.data_as(C.c_void_p), 1023, S4, nb)
inp = L.csdr_amd_fftfilt_input_size(f)
xi = xf[:2 * S4 * nb * inp]; yo = yf[:2 * S4 * nb * inp]
ms = timeit(lambda: L.csdr_amd_fftfilt_process(f, xi.data_ptr(), yo.data_ptr(), nb, nb * inp, nb * inp))
report("bandpass_fir_fft_cc fft=65536 taps=1023 (64 streams x 16 blocks)", ms, 16 * S4 * nb * inp, S4 * nb * inp)
L.csdr_amd_fftfilt_destroy(f)

# ---- the audio-rate operators of ONE stream (what a `csdr <command>` process of a shell pipeline runs): few streams, long in time -- the shapes whose first versions
# (a lane per stream, coefficients built on the host) bounded the literal README.md:66 pipeline (profiles/r4_notes.md)
n1 = 1 << 22
x1 = (torch.rand(n1 + 64, dtype=torch.float32, device="cuda") - 0.5); y1 = torch.empty(n1 + 64, dtype=torch.float32, device="cuda")
st1 = ctx.upload(np.zeros(4, np.float32))
report("deemphasis_wfm_ff 48000 50e-6 (1 stream x 4 Mi: chunks on their own lanes, verified run-in, bit exact)",
       timeit(lambda: L.csdr_amd_deemphasis_wfm_ff(ctx.h, x1.data_ptr(), y1.data_ptr(), 1, n1, n1, n1, 50e-6, 48000, st1.ptr)), 8 * n1, n1)
L.csdr_amd_fracdec_set_where.argtypes = [C.c_void_p, C.c_float]
for rate in (5.0, 5.5):
    fd = L.csdr_amd_fracdec_create(rate, 12, None, 0)
    proc = C.c_int(0)
    def fd_step():
        L.csdr_amd_fracdec_set_where(fd, 5.0)            # (every call a fresh plan, as the CLI's calls with their changing sizes have)
        L.csdr_amd_fractional_decimator_ff(ctx.h, fd, x1.data_ptr(), y1.data_ptr(), 1, n1 - 8 * int(_fd_k[0] % 7), n1, n1, C.byref(proc)); _fd_k[0] += 1
    _fd_k = [0]
    report("fractional_decimator_ff %g (1 stream x 4 Mi, a new plan per call: exact rates -- the host walks the WINDOWS in closed form, the device evaluates positions and Lagrange coefficients)" % rate,
           timeit(fd_step, reps=6, warm=1), (4 + 4 / rate) * n1, n1)
    L.csdr_amd_fracdec_destroy(fd)
g1 = ctx.upload(np.ones(1, np.float32))
n_agc = 1 << 20
report("agc_ff (1 stream x 1 Mi, calls of 1024 samples)",
       timeit(lambda: L.csdr_amd_agc_ff(ctx.h, x1.data_ptr(), y1.data_ptr(), 1, n_agc, 1024, n_agc, n_agc, 0.2, 0.01, 0.0001, 65536.0, 200, 0, 0.999, g1.ptr), reps=3, warm=1), 8 * n_agc, n_agc,
       {"bound": "serial per stream: a data-dependent state machine per sample (libcsdr_gpl.c:163-260) on one lane; the HBM roofline does not apply"})
del x1, y1

# ---- IMA ADPCM codec (f3): a serial state machine per stream (one lane per stream).  The call's time is the per-stream chain (~55 dependent instructions per sample at
# one wave instruction per ~5 cycles = ~190 ns per sample) whatever the stream count, up to 2 x 1024 x 64 = 131072 streams; staging the rows through LDS for coalesced
# accesses, the step table in LDS and spreading the streams over more waves were all measured and changed nothing (profiles/r2_notes.md)
for S5, n5 in ((4096, 48000), (65536, 12000)):
    x16 = torch.randint(-20000, 20000, (S5 * n5,), dtype=torch.int16, device="cuda")
    enc = torch.empty(S5 * n5 // 2, dtype=torch.uint8, device="cuda"); dec = torch.empty(S5 * n5, dtype=torch.int16, device="cuda")
    stt = torch.zeros(2 * S5, dtype=torch.int32, device="cuda")
    report("encode_ima_adpcm_i16_u8 (%d streams x %d)" % (S5, n5),
           timeit(lambda: L.csdr_amd_encode_ima_adpcm_i16_u8(ctx.h, x16.data_ptr(), enc.data_ptr(), S5, n5, n5, n5 // 2, stt.data_ptr()), reps=3, warm=1), 2.5 * S5 * n5, S5 * n5,
           {"bound": "serial per stream: every code depends on the predictor the previous code left (ima_adpcm.c:136-152): one lane per stream, time = one stream's chain whatever the stream count; the HBM roofline does not apply"})
    report("decode_ima_adpcm_u8_i16 (%d streams x %d)" % (S5, n5),
           timeit(lambda: L.csdr_amd_decode_ima_adpcm_u8_i16(ctx.h, enc.data_ptr(), dec.data_ptr(), S5, n5 // 2, n5 // 2, n5, stt.data_ptr()), reps=3, warm=1), 2.5 * S5 * n5, S5 * n5)
    del x16, enc, dec
ctx.close()
