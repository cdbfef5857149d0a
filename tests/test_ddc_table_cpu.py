"""CPU validation of the matrix-core receiver front end's weight table and index arithmetic (csdr_amd/csrc/ddc_mfma.hip):
one tile evaluated the way k_ddc_mfma does (one phase-independent weight set, K-range split over four waves, snapshot / half-K-step
handling of 1024-chunk boundaries, post factors C_m D^e, prefix-sum offset constants; exact int8 digit arithmetic as
v_mfma_i32_16x16x64_i8 does) must equal the direct double-precision evaluation of
    y[k] = sum_t h[t] * R[n] * u8_to_float(x[n]),   R[n] = C_chunk(n) * D^(n mod 1024)
for every 16-sample position of the tile inside a chunk (all boundary positions in all four waves)."""
import ctypes as C
import numpy as np
import pytest

f32 = np.float32


@pytest.mark.parametrize("D,Lt,rate", [(50, 801, 0.11), (50, 801, -0.4321), (10, 79, -0.085), (20, 321, 0.3)])
def test_ddc_tile_matches_direct_evaluation(port, D, Lt, rate):
    import csdr_amd
    L = csdr_amd.lib()
    fn = L.csdr_amd_debug_ddc_mfma_tile
    fn.argtypes = [C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p]
    taps = port.firdes_lowpass_f(Lt, 0.5 / D)
    rng = np.random.default_rng(11)
    PI = f32(3.14159265358979323846)
    inc = f32(f32(f32(rate) * f32(2)) * PI)
    d = complex(float(f32(np.cos(np.float64(inc)))), float(f32(np.sin(np.float64(inc)))))
    Dk = d ** np.arange(1024)
    out = np.zeros(16, f32)
    worst = 0.0
    scale = np.abs(taps).sum()
    for pos in range(0, 1024, 16):                               # every position of the tile start inside a chunk
        n0 = 1024 * 7 + pos
        window = rng.integers(0, 256, 2304, dtype=np.uint8)
        ct = np.array([[np.cos(a), np.sin(a)] for a in rng.uniform(-np.pi, np.pi, 3)], f32)   # unrelated seeds: a wrong chunk assignment shows
        assert fn(D, Lt, rate, taps.ctypes.data, n0, window.ctypes.data, ct.ctypes.data, out.ctypes.data) == 0
        xs = window.astype(np.float64) / 127.5 - 1.0
        xc = xs[0::2] + 1j * xs[1::2]
        Cc = ct[:, 0].astype(np.float64) + 1j * ct[:, 1].astype(np.float64)
        for o in range(8):
            n = n0 + D * o + np.arange(Lt)
            R = Cc[(n >> 10) - (n0 >> 10)] * Dk[n & 1023]
            y = np.sum(taps.astype(np.float64) * R * xc[D * o:D * o + Lt])
            got = complex(out[2 * o], out[2 * o + 1])
            worst = max(worst, abs(got - y))
    assert worst < 1e-6 * max(scale, 1.0), worst                 # 23-bit weights, float D^e table and float partial sums


def test_ddc_unsupported_shapes():
    import csdr_amd
    L = csdr_amd.lib()
    fn = L.csdr_amd_debug_ddc_mfma_tile
    fn.argtypes = [C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p]
    taps = np.ones(2000, f32); w = np.zeros(2304, np.uint8); ct = np.zeros(6, f32); out = np.zeros(16, f32)
    assert fn(51, 801, 0.1, taps.ctypes.data, 0, w.ctypes.data, ct.ctypes.data, out.ctypes.data) == -1      # odd decimation
    assert fn(50, 1200, 0.1, taps.ctypes.data, 0, w.ctypes.data, ct.ctypes.data, out.ctypes.data) == -1     # window too long
    assert fn(50, 801, 0.1, taps.ctypes.data, 8, w.ctypes.data, ct.ctypes.data, out.ctypes.data) == -1      # tile start not 16-sample aligned
