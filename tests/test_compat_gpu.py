"""The drop-in boundary on the GPU: libcsdr_amd.so is loaded through the SAME ctypes harness that drives the compiled reference
(oracle.Ref: reference symbol names, signatures, by-value structs, host pointers) and must reproduce the oracle on every function
of the hot path.  This is the test a user of the reference would write after swapping the shared object."""
import numpy as np
import pytest
from oracle import relrms

pytestmark = pytest.mark.gpu
c64 = np.complex64
f32 = np.float32
TOL = 1e-5


@pytest.fixture(scope="module")
def ours():
    import torch  # noqa: F401
    import csdr_amd
    import oracle
    return oracle.Ref(csdr_amd.LIB_PATH)


def crand(rng, n):
    return (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(c64)


def test_design_and_geometry(ours, port):
    for tbw in [0.05, 0.005, 0.001]:
        assert ours.firdes_filter_len(tbw) == port.firdes_filter_len(tbw)
    assert np.array_equal(ours.firdes_lowpass_f(79, 0.05), port.firdes_lowpass_f(79, 0.05))
    assert np.array_equal(ours.firdes_bandpass_c(255, -0.1, 0.2), port.firdes_bandpass_c(255, -0.1, 0.2))
    for x in [1, 127, 128, 65536]:
        assert ours.next_pow2(x) == port.next_pow2(x) and ours.log2n(x) == port.log2n(x)
    a, ea = ours.fastddc_init(0.001, 256, 0.123)
    b, eb = port.fastddc_init(0.001, 256, 0.123)
    assert ea == eb and a.as_dict() == b.as_dict()


def test_converters_bit_exact(ours, port):
    u8 = np.arange(256, dtype=np.uint8)
    assert np.array_equal(ours.convert_u8_f(u8).view(np.uint32), port.convert_u8_f(u8).view(np.uint32))
    s16 = np.arange(-32768, 32768, dtype=np.int16)
    assert np.array_equal(ours.convert_s16_f(s16).view(np.uint32), port.convert_s16_f(s16).view(np.uint32))
    x = np.concatenate([np.linspace(-1.5, 1.5, 4099, dtype=f32), np.array([3.0, -3.0, 1e20, np.nan, np.inf], f32)])
    for name in ("convert_f_u8", "convert_f_s8", "convert_f_s16"):
        assert np.array_equal(getattr(ours, name)(x), getattr(port, name)(x))
    for big in (0, 1):
        assert np.array_equal(ours.convert_f_s24(x, big), port.convert_f_s24(x, big))
        raw = np.random.default_rng(3).integers(0, 256, 3 * 1001, dtype=np.uint8)
        assert np.array_equal(ours.convert_s24_f(raw, big).view(np.uint32), port.convert_s24_f(raw, big).view(np.uint32))


@pytest.mark.parametrize("name", ["shift_addition_cc", "shift_math_cc", "shift_table_cc", "shift_unroll_cc", "shift_addfast_cc"])
def test_shifters_with_cli_chunking(ours, port, name):
    x = crand(np.random.default_rng(12), 1024 * 24)
    (a, pa), (b, pb) = getattr(ours, name)(x, -0.085), getattr(port, name)(x, -0.085)
    assert relrms(a, b) < TOL and abs(pa - pb) < 1e-5


def test_shift_fc_and_decimating(ours, port):
    rng = np.random.default_rng(13)
    xr = rng.uniform(-1, 1, 4096).astype(f32)
    assert relrms(ours.shift_addition_fc(xr, 0.11)[0], port.shift_addition_fc(xr, 0.11)[0]) < 1e-6
    x = crand(rng, 448)
    sa = sb = (0, 0.0, 0)
    for _ in range(6):
        ya, sa = ours.decimating_shift_addition_cc(x, 0.0123, 3, sa)
        yb, sb = port.decimating_shift_addition_cc(x, 0.0123, 3, sb)
        assert sa[0] == sb[0] and sa[2] == sb[2] and relrms(ya, yb) < 1e-6


def test_fir_decimate_block_loop(ours, port):
    x = crand(np.random.default_rng(1234), 16384 * 3 + 777)
    taps = port.firdes_lowpass_f(79, 0.05)
    assert ours.fir_decimate_cc_block(x[:16384], 10, taps).size == 1631
    a = ours.fir_decimate_cc(x, 10, taps)          # the CLI's refeed loop (csdr.c:1160-1176) driven on host buffers
    b = port.fir_decimate_cc(x, 10, taps)
    assert a.size == b.size and relrms(a, b) < TOL


def test_demod_audio_blocks(ours, port):
    rng = np.random.default_rng(21)
    t = np.arange(1024 * 8)
    x = (0.7 * np.exp(2j * np.pi * np.cumsum(0.03 * np.sin(t / 300.0)))).astype(c64)
    (a, la), (b, lb) = ours.fmdemod_quadri_cf(x), port.fmdemod_quadri_cf(x)
    assert relrms(a, b) < 1e-6 and la == lb
    r = rng.uniform(-1.5, 1.5, 1024 * 6).astype(f32)
    (a, la), (b, lb) = ours.deemphasis_wfm_ff(r, 50e-6, 48000), port.deemphasis_wfm_ff(r, 50e-6, 48000)
    assert np.array_equal(a, b) and f32(la) == f32(lb)
    assert np.array_equal(ours.limit_ff(r, 1.0), port.limit_ff(r, 1.0))
    assert np.array_equal(ours.gain_ff(r, 0.37), port.gain_ff(r, 0.37))
    import csdr_amd, ctypes as C
    taps = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "nfm_deemph_taps.npz"))["sr48000"]
    a, b = ours.deemphasis_nfm_ff(r[:2048], 48000), port.deemphasis_nfm_ff(r[:2048], taps)
    assert a.size == b.size and relrms(a, b) < TOL
    assert ours.deemphasis_nfm_ff(r[:1024], 12345).size == 0
    assert relrms(ours.fastagc_ff(r), port.fastagc_ff(r)) < 1e-6
    for rate in (5.0, 2.5):
        a, b = ours.fractional_decimator_ff(r, rate, bufsize=1024), port.fractional_decimator_ff(r, rate)
        n = min(a.size, b.size)
        assert n > 1000 and relrms(a[:n], b[:n]) < TOL


def test_fft_layer_and_overlap_add(ours, port):
    x = crand(np.random.default_rng(31), 4096)
    assert relrms(ours.fft_c2c(x, True), port.fft_c2c(x, True)) < 2e-6
    taps = port.firdes_bandpass_c(255, -0.1, 0.2)
    xx = crand(np.random.default_rng(3), (4096 - 255 + 1) * 5)
    assert relrms(ours.bandpass_fir_fft_cc(xx, taps, 4096), port.bandpass_fir_fft_cc(xx, taps, 4096)) < TOL


def test_fastddc_inv_cc_dropin(ours, port):
    D, tbw, shift = 16, 0.05, -0.1
    ddc, _ = port.fastddc_init(tbw, D, shift)
    d2, _ = ours.fastddc_init(tbw, D, shift)
    x = crand(np.random.default_rng(4), ddc.input_size * 8)
    spec = port.fastddc_fwd_cc(x, ddc)
    tf = port.fastddc_taps_fft(ddc, shift, D)
    a = ours.fastddc_inv_cc(spec, d2, tf)
    b = port.fastddc_inv_cc(spec, ddc, tf)
    assert a.size == b.size and relrms(a, b) < TOL


def test_f2_dropin_symbols(ours, port):
    """amdemod_cf ... agc_ff called exactly as a libcsdr client would (host pointers, by-value state structs)."""
    rng = np.random.default_rng(51)
    x = crand(rng, 9000)
    assert relrms(ours.amdemod_cf(x), port.amdemod_cf(x)) <= TOL
    assert np.array_equal(ours.amdemod_estimator_cf(x), port.amdemod_estimator_cf(x))
    a, pa = ours.fmdemod_atan_cf(x, 0.3); b, pb = port.fmdemod_atan_cf(x, 0.3)
    assert relrms(a, b) <= TOL and abs(pa - pb) <= 1e-6
    assert relrms(ours.logpower_cf(x, 3.0), port.logpower_cf(x, 3.0)) <= TOL
    r = (rng.uniform(-1, 1, 9000) + 0.3).astype(f32)
    (y, s), (w, ws) = ours.dcblock_ff(r, 0, (0.1, 0.2)), port.dcblock_ff(r, 0, (0.1, 0.2))
    assert relrms(y, w) <= TOL and np.allclose(s, ws, atol=1e-5)
    (y, l), (w, wl) = ours.fastdcblock_ff(r, 1024, 0.1), port.fastdcblock_ff(r, 1024, 0.1)
    assert relrms(y, w) <= TOL and abs(l - wl) <= 1e-6
    sig = (r * np.repeat(rng.uniform(0.01, 1, 90), 100)).astype(f32)
    (y, g), (w, wg) = ours.agc_ff(sig, 1024), port.agc_ff(sig, 1024)
    assert relrms(y, w) <= TOL and abs(g - wg) <= 1e-4 * max(1.0, abs(wg))
    assert np.array_equal(ours.precalculate_window(512, "BLACKMAN"), port.precalculate_window(512, "BLACKMAN"))
    wnd = port.precalculate_window(512)
    assert np.array_equal(ours.apply_precalculated_window_c(x[:512], wnd), (x[:512].view(f32).reshape(-1, 2) * wnd[:, None]).reshape(-1).view(c64))


def test_f3_adpcm_dropin(ours, port):
    rng = np.random.default_rng(52)
    x = (8000 * np.sin(np.arange(9001) * 0.01) + rng.integers(-3000, 3000, 9001)).astype(np.int16)
    (a, sa), (b, sb) = ours.encode_ima_adpcm_i16_u8(x, (5, -100)), port.encode_ima_adpcm_i16_u8(x, (5, -100))
    assert np.array_equal(a, b) and sa == sb
    (c, sc), (d, sd) = ours.decode_ima_adpcm_u8_i16(a, (3, 77)), port.decode_ima_adpcm_u8_i16(a, (3, 77))
    assert np.array_equal(c, d) and sc == sd
