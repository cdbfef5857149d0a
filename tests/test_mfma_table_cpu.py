"""CPU validation of the matrix-core WFM front end's weight table and lane/byte layout (csdr_amd/csrc/wfm_mfma.hip):
one tile evaluated through the table (exact int8 digit arithmetic, as v_mfma_i32_16x16x64_i8 does) must equal the
direct double-precision evaluation of  y[k] = sum_t h[t] * R[n] * u8_to_float(x[n])  for the 8 FIR outputs of the tile,
for every window position inside a chunk including the ones that straddle a shift_addition_cc chunk boundary."""
import ctypes as C
import numpy as np
import pytest

f32 = np.float32


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
