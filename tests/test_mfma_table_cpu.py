"""CPU validation of the matrix-core WFM front end's weight table and lane/byte layout (csdr_amd/csrc/wfm_mfma.hip):
one tile evaluated through the table (exact int8 digit arithmetic, as v_mfma_i32_16x16x64_i8 does) must equal the
direct double-precision evaluation of  y[k] = sum_t h[t] * R[n] * u8_to_float(x[n])  for the 8 FIR outputs of the tile,
for every tile phase including the ones that straddle a shift_addition_cc chunk boundary."""
import ctypes as C
import numpy as np
import pytest

f32 = np.float32


def test_mfma_tile_table_matches_direct_evaluation(port):
    import csdr_amd
    L = csdr_amd.lib()
    fn = L.csdr_amd_debug_wfm_mfma_tile
    fn.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    D, Lt, F, rate = 10, 79, 5, -0.085
    taps = port.firdes_lowpass_f(Lt, 0.05)
    rng = np.random.default_rng(5)
    PI = f32(3.14159265358979323846)
    inc = f32(f32(rate * 2) * PI)
    d = complex(float(f32(np.cos(np.float64(inc)))), float(f32(np.sin(np.float64(inc)))))
    Dk = d ** np.arange(1024)
    nph = C.c_int(); S = C.c_int(); stride = C.c_int(); woff = C.c_int()
    out = np.zeros(32, f32)
    worst = 0.0; n_straddle = 0
    c0 = np.array([np.cos(0.3), np.sin(0.3)], f32); c1 = np.array([np.cos(-1.1), np.sin(-1.1)], f32)
    window = rng.integers(0, 256, 512, dtype=np.uint8)
    assert fn(D, Lt, F, rate, taps.ctypes.data, 0, window.ctypes.data, c0.ctypes.data, c1.ctypes.data, out.ctypes.data, nph, S, stride, woff) == 0
    assert nph.value == 128 and stride.value == 400 and woff.value == 176
    for ph in range(nph.value):
        window = rng.integers(0, 256, 512, dtype=np.uint8)
        assert fn(D, Lt, F, rate, taps.ctypes.data, ph, window.ctypes.data, c0.ctypes.data, c1.ctypes.data, out.ctypes.data, nph, S, stride, woff) == 0
        n_straddle += S.value >= 0
        s0 = 200 * ph + woff.value // 2                      # window base sample in the periodic frame
        chunk0 = s0 // 1024
        xs = (window.astype(np.float64) / 127.5 - 1.0)
        xc = xs[0::2] + 1j * xs[1::2]
        C0 = complex(float(c0[0]), float(c0[1])); C1 = complex(float(c1[0]), float(c1[1]))
        for q in range(4):
            for which in range(2):
                off = D * (F * q + 9 + which) - woff.value // 2
                g = s0 + off + np.arange(Lt)
                R = np.where(g // 1024 == chunk0, C0, C1) * Dk[g % 1024]
                y = np.sum(taps.astype(np.float64) * R * xc[off:off + Lt])
                got = complex(out[4 * q + 2 * which], out[4 * q + 2 * which + 1])
                worst = max(worst, abs(got - y))
    assert n_straddle > 20                                   # ~25 % of the phases contain a chunk boundary
    assert worst < 2e-6, worst                               # 23-bit weights, outputs are O(0.1..1)


def test_seq_tile_matches_direct_evaluation(port):
    """The sequential kernel's phase-independent weight set: every window position inside a chunk (all chunk-boundary positions at 16-byte
    granularity) against the direct double-precision evaluation."""
    import csdr_amd
    L = csdr_amd.lib()
    fn = L.csdr_amd_debug_wfm_seq_tile
    fn.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p]
    D, Lt, F, rate = 10, 79, 5, -0.085
    taps = port.firdes_lowpass_f(Lt, 0.05)
    rng = np.random.default_rng(6)
    PI = f32(3.14159265358979323846)
    inc = f32(f32(rate * 2) * PI)
    d = complex(float(f32(np.cos(np.float64(inc)))), float(f32(np.sin(np.float64(inc)))))
    Dk = d ** np.arange(1024)
    out = np.zeros(16, f32)
    worst = 0.0
    for pos in range(0, 1024, 8):
        n0 = 1024 * 5 + pos
        window = rng.integers(0, 256, 512, dtype=np.uint8)
        ct = np.array([[np.cos(a), np.sin(a)] for a in rng.uniform(-np.pi, np.pi, 2)], f32)
        assert fn(D, Lt, F, rate, taps.ctypes.data, n0, window.ctypes.data, ct.ctypes.data, out.ctypes.data) == 0
        xs = window.astype(np.float64) / 127.5 - 1.0
        xc = xs[0::2] + 1j * xs[1::2]
        Cc = ct[:, 0].astype(np.float64) + 1j * ct[:, 1].astype(np.float64)
        for q in range(4):
            for which in range(2):
                off = D * (F * q + 9 + which) - 88               # row's first sample relative to the window base (win_off = 176 bytes)
                n = n0 + off + np.arange(Lt)
                R = Cc[(n >> 10) - (n0 >> 10)] * Dk[n & 1023]
                y = np.sum(taps.astype(np.float64) * R * xc[off:off + Lt])
                got = complex(out[4 * q + 2 * which], out[4 * q + 2 * which + 1])
                worst = max(worst, abs(got - y))
    assert worst < 1e-6, worst
