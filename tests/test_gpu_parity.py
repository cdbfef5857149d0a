"""GPU parity tests: the HIP path (through the C ABI of libcsdr_amd.so) against the CPU oracle
(oracle/csdr_oracle.c, pinned to the compiled reference by tests/test_oracle_vs_ref.py) on identical seeded
inputs.  Gates (BASELINE.json north_star): bit exact for convert_*; relative RMS <= 1e-5 for float paths."""
import os
import numpy as np
import pytest
from oracle import relrms

pytestmark = pytest.mark.gpu
c64 = np.complex64
f32 = np.float32
TOL = 1e-5


@pytest.fixture(scope="module")
def gpu():
    import torch  # noqa: F401  (loads the HIP runtime torch bundles first, as bench.py does)
    import csdr_amd
    ctx = csdr_amd.Context(0)
    assert ctx.arch().startswith("gfx950")
    yield ctx
    ctx.close()


def crand(rng, n):
    return (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(c64)


def fm_signal(rng, n, dev=0.03125, offset=0.0):
    t = np.arange(n)
    msg = np.sin(2 * np.pi * 1e3 / 2.4e6 * t) + 0.3 * rng.uniform(-1, 1, n)
    ph = 2 * np.pi * np.cumsum(dev * msg) + 2 * np.pi * offset * t
    return (0.7 * np.exp(1j * ph) + 0.01 * (rng.normal(size=n) + 1j * rng.normal(size=n))).astype(c64)


def to_u8(sig):
    iq = np.empty(2 * sig.size, f32); iq[0::2] = sig.real; iq[1::2] = sig.imag
    return np.clip(np.round(127.5 * (iq + 1)), 0, 255).astype(np.uint8)


# ---------------------------------------------------------------- converters: bit exact
def float_probe_set():
    rng = np.random.default_rng(7)
    grid = np.linspace(-1, 1, 65537, dtype=f32)
    special = np.array([0.0, -0.0, 1.0, -1.0, np.nextafter(f32(1), f32(0)), 1.5, -1.5, 3.0, -3.0, 1e-40, -1e-40,
                        2.0, -2.0, 100.0, -100.0, 65535.9, -65536.2, 7e4, -7e4, 3e9, -3e9, 1e20, np.inf, -np.inf, np.nan], dtype=f32)
    wide = rng.uniform(-4, 4, 20001).astype(f32)
    return np.concatenate([grid, special, wide])


def test_convert_to_float_exhaustive(gpu, port):
    u8 = np.arange(256, dtype=np.uint8).repeat(3)[:-1]            # odd length: exercises the scalar tail
    assert np.array_equal(gpu.convert_u8_f(u8).view(np.uint32), port.convert_u8_f(u8).view(np.uint32))
    s8 = np.arange(-128, 128, dtype=np.int8)
    assert np.array_equal(gpu.convert_s8_f(s8).view(np.uint32), port.convert_s8_f(s8).view(np.uint32))
    s16 = np.arange(-32768, 32768, dtype=np.int16)
    assert np.array_equal(gpu.convert_s16_f(s16).view(np.uint32), port.convert_s16_f(s16).view(np.uint32))


@pytest.mark.parametrize("name", ["convert_f_u8", "convert_f_s8", "convert_f_s16"])
def test_convert_from_float_bit_exact(gpu, port, name):
    x = float_probe_set()
    assert np.array_equal(getattr(gpu, name)(x), getattr(port, name)(x))


@pytest.mark.parametrize("big", [0, 1])
def test_convert_s24_bit_exact(gpu, port, big):
    x = float_probe_set()
    assert np.array_equal(gpu.convert_f_s24(x, big), port.convert_f_s24(x, big))
    raw = np.random.default_rng(3).integers(0, 256, 3 * 50001, dtype=np.uint8)
    assert np.array_equal(gpu.convert_s24_f(raw, big).view(np.uint32), port.convert_s24_f(raw, big).view(np.uint32))


def test_convert_empty(gpu):
    assert gpu.convert_u8_f(np.zeros(0, np.uint8)).size == 0


# ---------------------------------------------------------------- shifters
@pytest.mark.parametrize("rate", [-0.085, 0.3141, 4e-4, 0.5, -0.5])
def test_shift_addition(gpu, port, rate):
    rng = np.random.default_rng(11)
    x = crand(rng, 1024 * 1000)                                  # >= 1000 consecutive 1024-chunks
    (a, pa), (b, pb) = gpu.shift_cc(x, rate, "addition"), port.shift_addition_cc(x, rate)
    assert relrms(a, b) < 1e-6                                    # exact replay: expected bit identical up to libm last bits
    assert abs(pa - pb) < 1e-6
    # state carry across calls
    (a1, p1) = gpu.shift_cc(x[:1024 * 300], rate, "addition")
    (a2, p2) = gpu.shift_cc(x[1024 * 300:], rate, "addition", phase=p1)
    assert relrms(np.concatenate([a1, a2]), b) < 1e-6


@pytest.mark.parametrize("variant", ["math", "table", "unroll", "addfast"])
@pytest.mark.parametrize("rate", [-0.085, 0.3141, 4e-4])
def test_shift_variants(gpu, port, variant, rate):
    rng = np.random.default_rng(12)
    x = crand(rng, 1024 * 64)
    a, pa = gpu.shift_cc(x, rate, variant)
    b, pb = getattr(port, "shift_%s_cc" % variant)(x, rate)
    assert relrms(a, b) < TOL
    assert abs(pa - pb) < 1e-5


def test_shift_batch_and_fc_and_decimating(gpu, port):
    rng = np.random.default_rng(13)
    x = np.stack([crand(rng, 4096) for _ in range(5)])
    a, _ = gpu.shift_cc(x, 0.11)
    for s in range(5):
        assert relrms(a[s], port.shift_addition_cc(x[s], 0.11)[0]) < 1e-6
    xr = rng.uniform(-1, 1, 4096 * 4).astype(f32)
    assert relrms(gpu.shift_addition_fc(xr, 0.11)[0], port.shift_addition_fc(xr, 0.11)[0]) < 1e-6
    x1 = crand(rng, 448)
    sa = sb = (0, 0.0, 0)
    for _ in range(20):
        ya, sa = gpu.decimating_shift_addition_cc(x1, 0.0123, 3, sa)
        yb, sb = port.decimating_shift_addition_cc(x1, 0.0123, 3, sb)
        assert sa[0] == sb[0] and sa[2] == sb[2] and abs(sa[1] - sb[1]) < 1e-6
        assert relrms(ya, yb) < 1e-6


# ---------------------------------------------------------------- FIR decimator
@pytest.mark.parametrize("D,ntaps", [(10, 79), (50, 801), (2, 133), (256, 3999), (10, 1023), (3, 7)])
def test_fir_decimate(gpu, port, D, ntaps):
    rng = np.random.default_rng(1234)
    x = crand(rng, 16384 * 5 + 777)                                # length not a multiple of D
    taps = port.firdes_lowpass_f(ntaps, 0.5 / D)
    a, b = gpu.fir_decimate_cc(x, D, taps), port.fir_decimate_cc(x, D, taps)
    assert a.size == b.size == (x.size - ntaps) // D + 1
    assert relrms(a, b) < TOL


@pytest.mark.parametrize("D,ntaps", [(50, 801), (32, 513), (20, 401)])
def test_fir_decimate_long_filter_dma_kernel(gpu, port, D, ntaps):
    """k_fir_mfma3 (round 5: taps operand in registers, window by LDS-DMA, swizzle applied to the source granules): long filters on complexf with an EVEN stream
    length take it -- several tiles of 128 outputs with a ragged last one, five streams, a stream shorter than one tile --, an odd length stays on k_fir_mfma (the
    DMA moves 16-byte granules); all against the oracle, and the two kernels against each other."""
    rng = np.random.default_rng(4321)
    taps = port.firdes_lowpass_f(ntaps, 0.5 / D)
    n_even = 128 * D * 3 + ntaps + D * 37 + ((ntaps + D) % 2)        # three full tiles and 37 / 38 outputs more
    n_even += n_even % 2
    xs = np.stack([crand(rng, n_even) for _ in range(5)])
    ys = gpu.fir_decimate_cc(xs, D, taps)
    assert gpu.L.csdr_amd_fir_last_kernel() == b"k_fir_mfma3"
    for s in range(5):
        want = port.fir_decimate_cc(xs[s], D, taps)
        assert ys[s].size == want.size and relrms(ys[s], want) < TOL, s
    x_odd = xs[0][:n_even - 1]
    y_odd = gpu.fir_decimate_cc(x_odd, D, taps)
    assert gpu.L.csdr_amd_fir_last_kernel() == b"k_fir_mfma"
    w_odd = port.fir_decimate_cc(x_odd, D, taps)
    assert y_odd.size == w_odd.size and relrms(y_odd, w_odd) < TOL
    m = min(y_odd.size, ys[0].size)
    assert relrms(y_odd[:m], ys[0][:m]) < 2e-6                        # the same sums up to the K-split points
    n_short = ntaps + D * 20 + ((ntaps + D * 20) % 2)                # 21 outputs: one partial tile, most of the window behind the stream's end
    x1 = crand(rng, n_short)
    y1 = gpu.fir_decimate_cc(x1, D, taps)
    assert gpu.L.csdr_amd_fir_last_kernel() == b"k_fir_mfma3"
    w1 = port.fir_decimate_cc(x1, D, taps)
    assert y1.size == w1.size and relrms(y1, w1) < TOL


def test_fir_decimate_c1_and_edges(gpu, port):
    rng = np.random.default_rng(1234)
    x = crand(rng, 16384)                                          # BASELINE config 1
    taps = port.firdes_lowpass_f(79, 0.05)
    a = gpu.fir_decimate_cc(x, 10, taps)
    assert a.size == 1631 and relrms(a, port.fir_decimate_cc(x, 10, taps)) < TOL
    assert gpu.fir_decimate_cc(x[:78], 10, taps).size == 0          # shorter than the filter: no output
    assert gpu.fir_decimate_cc(x[:79], 10, taps).size == 1
    xs = np.stack([crand(rng, 5000) for _ in range(7)])             # batch of ragged-length-free streams
    ys = gpu.fir_decimate_cc(xs, 10, taps)
    for s in range(7):
        assert relrms(ys[s], port.fir_decimate_cc(xs[s], 10, taps)) < TOL


# ---------------------------------------------------------------- demod + audio
def test_fmdemod(gpu, port):
    rng = np.random.default_rng(21)
    x = fm_signal(rng, 1024 * 40)
    x[100] = 0; x[5000:5003] = 0
    (a, la), (b, lb) = gpu.fmdemod_quadri_cf(x), port.fmdemod_quadri_cf(x)
    assert relrms(a, b) < 1e-6 and a[100] == 0
    assert (float(la.real), float(la.imag)) == lb
    (a2, _) = gpu.fmdemod_quadri_cf(x[1024:], last=np.array([x[1023]]))
    assert relrms(a2, b[1024:]) < 1e-6


def test_fractional_decimator(gpu, port):
    rng = np.random.default_rng(22)
    x = rng.uniform(-1, 1, 1024 * 30).astype(f32)
    a = gpu.fractional_decimator_ff(x, 5.0)
    assert np.array_equal(a, port.fractional_decimator_ff(x, 5.0))
    assert np.array_equal(a, x[10:10 + 5 * a.size:5])
    for rate in [2.5, 4.17]:
        a, b = gpu.fractional_decimator_ff(x, rate), port.fractional_decimator_ff(x, rate)
        assert a.size == b.size and relrms(a, b) < TOL
    taps = port.firdes_lowpass_f(133, 0.5 / (2.5 - 0.03))
    a, b = gpu.fractional_decimator_ff(x, 2.5, taps=taps), port.fractional_decimator_ff(x, 2.5, taps=taps)
    assert a.size == b.size and relrms(a, b) < TOL
    for rate in (3.3, 4.17):                          # the CLI's window loop replayed by the device operator (csdr_amd_fracdec_set_cli_bufsize)
        a, b = gpu.fractional_decimator_ff(x, rate, bufsize=1024), port.fractional_decimator_ff(x, rate, bufsize=1024)
        assert a.size == b.size and relrms(a, b) <= TOL


def test_deemphasis_limit_gain(gpu, port):
    rng = np.random.default_rng(23)
    x = rng.uniform(-1.5, 1.5, (70, 1024 * 3 + 17)).astype(f32)
    a, la = gpu.deemphasis_wfm_ff(x, 50e-6, 48000)
    for s in range(70):
        b, lb = port.deemphasis_wfm_ff(x[s], 50e-6, 48000)
        assert np.array_equal(a[s], b) and la[s] == f32(lb)          # same operations in the same order: bit exact
    # few long streams (a CLI process has one): chunks of 256 samples on their own lanes with a run-in from state zero, checked against the predecessor's end state
    # (audio.hip: k_deemph_wfm_spec / _check / _fix) -- still the reference's bits: a carried state, a NaN state, a ragged end, slow time constants (longer run-ins)
    for nst, n, tau, sr, last in [(1, 100000, 50e-6, 48000, 0.37), (3, 70001, 75e-6, 48000, float("nan")), (2, 50000, 50e-6, 240000, -1.2), (1, 9000, 50e-6, 44100, 0.0),
                                  (1, 40000, 1e-3, 48000, 0.5), (1, 40000, 2e-3, 48000, 0.5)]:      # (1e-3: a run-in of 8 chunks; 2e-3: b^(256 * 8) > 2^-40 -> the serial kernel)
        xs = rng.uniform(-1.5, 1.5, (nst, n)).astype(f32)
        a, la = gpu.deemphasis_wfm_ff(xs, tau, sr, last=np.full(nst, last, f32))
        for s in range(nst):
            b, lb = port.deemphasis_wfm_ff(xs[s], tau, sr, last)
            assert np.array_equal(a[s], b) and la[s] == f32(lb), (nst, n, tau, sr)
    assert np.array_equal(gpu.limit_ff(x, 1.0).ravel(), port.limit_ff(x.ravel(), 1.0))
    assert np.array_equal(gpu.gain_ff(x, 0.37).ravel(), port.gain_ff(x.ravel(), 0.37))
    for sr in [48000, 44100, 8000, 11025]:
        taps = gpu.nfm_taps(sr)
        a, b = gpu.fir_ff(x[0, :3000], taps), port.deemphasis_nfm_ff(x[0, :3000], taps)
        assert a.size == b.size == 3000 - taps.size and relrms(a, b) < TOL
    assert gpu.nfm_taps(12345).size == 0


def test_fastagc(gpu, port):
    rng = np.random.default_rng(24)
    n = 1024 * 24
    env = (0.05 + np.abs(np.sin(np.arange(n) / 3000.0))).astype(f32)
    x = (rng.uniform(-1.5, 1.5, (3, n)).astype(f32)) * env
    for calls in (1, 5, 24):
        a = gpu.fastagc_ff(x, calls=calls)
        for s in range(3):
            b = port.fastagc_ff(x[s])
            assert np.all(a[s, :2048] == 0)
            assert relrms(a[s], b) < 1e-6


# ---------------------------------------------------------------- FFT paths
def test_fft(gpu):
    rng = np.random.default_rng(31)
    for n in [512, 65536]:
        x = crand(rng, n)
        assert relrms(gpu.fft_c2c(x, True), np.fft.fft(x.astype(np.complex128))) < 2e-6


@pytest.mark.parametrize("ntaps", [63, 127, 255, 511, 1023, 2047, 4095])
def test_bandpass_fir_fft_c3(gpu, port, ntaps):
    """BASELINE config 3: fft 65536, taps sweep; vs oracle on 4 blocks, vs direct convolution prefix."""
    rng = np.random.default_rng(3)
    fft = 65536; inp = fft - ntaps + 1
    x = crand(rng, inp * 4)
    taps = port.firdes_bandpass_c(ntaps, -0.1, 0.2)
    a = gpu.bandpass_fir_fft_cc(x, taps, fft)
    b = port.bandpass_fir_fft_cc(x, taps, fft)
    assert relrms(a, b) < TOL
    a2 = gpu.bandpass_fir_fft_cc(x, taps, fft, blocks_per_call=1)     # carry across calls
    assert relrms(a2, b) < TOL
    direct = np.convolve(x[:20000].astype(np.complex128), taps.astype(np.complex128))[:20000]
    assert relrms(a[:20000], direct) < TOL


@pytest.mark.parametrize("ntaps", [500, 2000, 4000])
def test_bandpass_fir_fft_even_taps_mix_the_kernels(gpu, port, ntaps):
    """An even tap count makes the block (fft_size - taps + 1) odd.  The wave / team kernels of the one-pass filter move 16 bytes per lane and take the calls with an
    even sample count; odd ones go to the 256- / 512-thread kernels.  Five blocks in calls of 2, 2, 1 (and 3, 2; and one by one) cross both kinds on ONE object: the
    state they hand each other is the same last taps - 1 input samples, so every split gives the oracle's stream (libcsdr.c:814-849)."""
    rng = np.random.default_rng(ntaps)
    fft = 16384; inp = fft - ntaps + 1
    assert inp % 2 == 1
    x = np.stack([crand(rng, inp * 5) for _ in range(3)])
    taps = port.firdes_bandpass_c(ntaps, -0.2, 0.1)
    want = [port.bandpass_fir_fft_cc(x[s], taps, fft) for s in range(3)]
    for per_call in (2, 3, 1, 5):
        a = gpu.bandpass_fir_fft_cc(x, taps, fft, blocks_per_call=per_call)
        for s in range(3):
            assert relrms(a[s], want[s]) < TOL, (per_call, s)


def test_bandpass_fir_fft_c3_streams(gpu, port):
    """fft 65536 (the three-pass transform of fft64k.hip) on several streams, blocks split over calls; the hipFFT path gives the same result."""
    rng = np.random.default_rng(33)
    fft, ntaps = 65536, 1023
    inp = fft - ntaps + 1
    x = np.stack([crand(rng, inp * 3) for _ in range(3)])
    taps = port.firdes_bandpass_c(ntaps, -0.3, 0.05)
    a = gpu.bandpass_fir_fft_cc(x, taps, fft, blocks_per_call=2)
    for s in range(3):
        want = np.convolve(x[s, :30000].astype(np.complex128), taps.astype(np.complex128))[:30000]
        assert relrms(a[s, :30000], want) < TOL
    assert relrms(a[1], port.bandpass_fir_fft_cc(x[1], taps, fft)) < TOL


def test_bandpass_fir_fft_paths(gpu, port, monkeypatch):
    """Which path serves which filter: <= 4096 taps the one-pass kernel (windows of 4096 / 8192 / 16384 points in LDS), longer filters and
    CSDR_AMD_FFTFILT_LDS_OFF the full-size transform (three passes at 65536, hipFFT otherwise); all of them give the oracle's samples (libcsdr.c:814-849)."""
    import ctypes as C
    rng = np.random.default_rng(35)
    L = gpu.L

    def path_of(ntaps, fft):
        taps = port.firdes_bandpass_c(ntaps, -0.1, 0.2)
        f = L.csdr_amd_fftfilt_create(gpu.h, fft, taps.ctypes.data_as(C.c_void_p), ntaps, 1, 1)
        assert f
        name, win = L.csdr_amd_fftfilt_kernel_name(f).decode(), L.csdr_amd_fftfilt_window(f)
        L.csdr_amd_fftfilt_destroy(f)
        return name, win
    assert path_of(63, 65536) == ("k_fftfilt_wave", 4096) and path_of(1023, 65536)[1] == 4096
    assert path_of(1025, 65536)[1] == 4096 and path_of(1041, 65536)[1] == 8192 and path_of(2047, 65536)[1] == 8192 and path_of(4095, 65536)[1] == 16384
    assert path_of(8191, 65536) == ("", 0)
    # a filter too long for LDS windows: the 65536-point path
    fft, ntaps = 65536, 8191
    x = crand(rng, (fft - ntaps + 1) * 2)
    taps = port.firdes_bandpass_c(ntaps, -0.1, 0.2)
    assert relrms(gpu.bandpass_fir_fft_cc(x, taps, fft, blocks_per_call=1), port.bandpass_fir_fft_cc(x, taps, fft)) < TOL
    # the full-size paths on the filters the one-pass kernel normally takes
    monkeypatch.setenv("CSDR_AMD_FFTFILT_LDS_OFF", "1")
    assert path_of(1023, 65536) == ("", 0)
    for ntaps, fft in [(1023, 65536), (255, 1024)]:
        x = np.stack([crand(rng, (fft - ntaps + 1) * 3) for _ in range(2)])
        taps = port.firdes_bandpass_c(ntaps, -0.1, 0.2)
        a = gpu.bandpass_fir_fft_cc(x, taps, fft, blocks_per_call=2)
        for s in range(2):
            assert relrms(a[s], port.bandpass_fir_fft_cc(x[s], taps, fft)) < TOL


def test_bandpass_one_pass_ragged(gpu, port):
    """The one-pass kernel at every window size: streams x blocks that do not divide into windows, calls shorter than the filter's history, pitches larger
    than the row; vs the oracle's block-by-block overlap-add."""
    rng = np.random.default_rng(36)
    for ntaps, fft, per in [(63, 256, 1), (1023, 2048, 3), (1500, 4096, None), (4095, 8192, 2), (3001, 65536, 1)]:
        inp = fft - ntaps + 1
        x = np.stack([crand(rng, inp * 5) for _ in range(5)])
        taps = port.firdes_bandpass_c(ntaps, -0.2, 0.3)
        a = gpu.bandpass_fir_fft_cc(x, taps, fft, blocks_per_call=per)
        for s in range(5):
            assert relrms(a[s], port.bandpass_fir_fft_cc(x[s], taps, fft)) < TOL


def test_bandpass_small_and_chained_overlap(gpu, port):
    rng = np.random.default_rng(5)
    for ntaps, fft in [(79, 256), (601, 1024)]:                        # second case: input_size (424) < overlap (600)
        inp = fft - ntaps + 1
        x = np.stack([crand(rng, inp * 9) for _ in range(3)])
        taps = port.firdes_bandpass_c(ntaps, 0.1, 0.3)
        for per in (None, 2):
            a = gpu.bandpass_fir_fft_cc(x, taps, fft, blocks_per_call=per)
            for s in range(3):
                assert relrms(a[s], port.bandpass_fir_fft_cc(x[s], taps, fft)) < TOL


@pytest.mark.parametrize("D,tbw,shifts", [(16, 0.05, [-0.1, 0.2, 0.33]), (256, 0.005, [0.3 + 0.5 / 256, -0.45]), (6, 0.05, [0.2])])
def test_fastddc(gpu, port, D, tbw, shifts):
    rng = np.random.default_rng(4)
    ddc, _ = gpu.fastddc_init(tbw, D, 0.0)
    pd, _ = port.fastddc_init(tbw, D, 0.0)
    nb = 40 if ddc.fft_size <= 4096 else 12
    x = crand(rng, ddc.input_size * nb)
    spec = gpu.fastddc_fwd_cc(x, ddc, blocks_per_call=7)
    pspec = port.fastddc_fwd_cc(x, pd)
    assert relrms(spec, pspec) < TOL
    outs = gpu.fastddc_inv_cc(pspec, tbw, D, shifts, blocks_per_call=5)
    for c, s in enumerate(shifts):
        pdc, _ = port.fastddc_init(tbw, D, s)
        ref_out = port.fastddc_inv_cc(pspec, pdc, port.fastddc_taps_fft(pdc, s, D))
        assert outs[c].size == ref_out.size and ref_out.size > 0
        assert relrms(outs[c], ref_out) < TOL


@pytest.mark.parametrize("blocks_per_call", [20, 3])
def test_fastddc_channel_tiled_fold(gpu, port, blocks_per_call):
    """>= 4 channels take the channel-tiled fold kernel (8 channels x 8 blocks or 4 x 4 per thread, ragged tiles on both axes)."""
    rng = np.random.default_rng(5)
    D, tbw = 16, 0.05
    shifts = [-0.4, -0.3, -0.1, 0.0, 0.07, 0.12, 0.2, 0.33, 0.41, 0.45, -0.22]      # 11 channels: full and ragged tiles of 8 (20 blocks) and of 4 (3 blocks)
    pd, _ = port.fastddc_init(tbw, D, 0.0)
    x = crand(rng, pd.input_size * 41)
    pspec = port.fastddc_fwd_cc(x, pd)
    outs = gpu.fastddc_inv_cc(pspec, tbw, D, shifts, blocks_per_call=blocks_per_call)
    for c, sft in enumerate(shifts):
        pdc, _ = port.fastddc_init(tbw, D, sft)
        ref_out = port.fastddc_inv_cc(pspec, pdc, port.fastddc_taps_fft(pdc, sft, D))
        assert outs[c].size == ref_out.size and relrms(outs[c], ref_out) < TOL


# ---------------------------------------------------------------- the fused WFM chain (BASELINE config 2)
def wfm_inputs(n_streams, n):
    return np.stack([to_u8(fm_signal(np.random.default_rng(1000 + s), n, offset=0.085)) for s in range(n_streams)])


def check_chain(s16, af, port, u8, taps, n_streams):
    for s in range(n_streams):
        ps, pf = port.wfm_chain(u8[s], -0.085, 10, taps)
        n = min(pf.size, af.shape[1])
        assert n >= u8.shape[1] // 2 // 50 - 8
        assert relrms(af[s, :n], pf[:n]) < TOL, "stream %d" % s
        d = np.abs(s16[s, :n].astype(np.int32) - ps[:n].astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 0.05


def test_wfm_chain_single_call(gpu, port):
    taps = port.firdes_lowpass_f(79, 0.05)
    u8 = wfm_inputs(4, 16384 * 12)
    s16, af = gpu.wfm_chain(u8, -0.085, 10, taps)
    check_chain(s16, af, port, u8, taps, 4)


def test_wfm_chain_streaming_blocks(gpu, port):
    taps = port.firdes_lowpass_f(79, 0.05)
    u8 = wfm_inputs(3, 16384 * 10 + 1024 * 3 + 500)                  # ragged tail: last block not a multiple of 1024
    for block in (16384, 1024 * 7):
        s16, af = gpu.wfm_chain(u8, -0.085, 10, taps, block=block)
        check_chain(s16, af, port, u8, taps, 3)
    one, _ = gpu.wfm_chain(u8, -0.085, 10, taps)
    blk, _ = gpu.wfm_chain(u8, -0.085, 10, taps, block=16384)
    n = min(one.shape[1], blk.shape[1])
    assert np.abs(one[:, :n].astype(np.int32) - blk[:, :n]).max() <= 1  # block-size invariance


def test_wfm_chain_full_size_properties(gpu, port):
    """BASELINE config-2 shape at reduced stream count x full block length: properties that need no oracle run.
    (a) streams with identical input give identical output; (b) first seconds match the oracle on one stream."""
    taps = port.firdes_lowpass_f(79, 0.05)
    n = 2400000
    base = to_u8(fm_signal(np.random.default_rng(1000), n, offset=0.085))
    u8 = np.stack([base, base, to_u8(fm_signal(np.random.default_rng(1001), n, offset=0.085))])
    s16, af = gpu.wfm_chain(u8, -0.085, 10, taps)
    assert s16.shape[1] >= 48000 - 4
    assert np.array_equal(s16[0], s16[1]) and not np.array_equal(s16[0], s16[2])
    ps, pf = port.wfm_chain(base, -0.085, 10, taps)
    m = min(pf.size, af.shape[1])
    assert relrms(af[0, :m], pf[:m]) < TOL


def test_dropin_client_binary(gpu):
    """The client of tests/data/dropin_client.c (compiled against the reference's headers on the build host, linked against
    libcsdr_amd.so) runs on the GPU and reproduces the reference's results."""
    import os, re, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "data", "dropin_client.bin")
    if not os.path.exists(exe):
        pytest.skip("dropin_client.bin not built (needs the reference headers on the build host)")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(root, "csdr_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr
    m = re.search(r"ntaps=(\d+) outputs=(\d+) phase=(\S+) demod_energy=(\S+) last=\((\S+),(\S+)\) fft=(\d+) err=(\d+)", out.stdout)
    assert m, out.stdout
    assert int(m.group(1)) == 79 and int(m.group(2)) == 1631 and int(m.group(7)) == 65536 and int(m.group(8)) == 0
    golden = open(os.path.join(root, "tests", "golden", "dropin_client_ref.txt")).read()
    g = re.search(r"phase=(\S+) demod_energy=(\S+)", golden)
    assert abs(float(m.group(3)) - float(g.group(1))) < 2e-5
    assert abs(float(m.group(4)) / float(g.group(2)) - 1) < 1e-4


@pytest.mark.parametrize("mode", ["chain", "chain_blocks", "chain_small_blocks", "unfused"])
def test_nfm_chain_device_resident(gpu, port, mode):
    """BASELINE config 5 shape (README.md:87) at reduced size: 3 channels x 0.25 s.  chain = the csdr_amd_nfm object (matrix-core front end,
    audio-rate back end) in one call / in blocks that split AGC blocks and filter history / in blocks too small for the matrix-core kernel;
    unfused = every stage a device batch call."""
    n = 600000 if mode != "chain_small_blocks" else 1024 * 150
    u8 = np.stack([to_u8(fm_signal(np.random.default_rng(5000 + s), n, dev=5e3 / 2.4e6, offset=0.05)) for s in range(3)])
    if mode == "unfused":
        pcm, af = gpu.nfm_chain_unfused(u8, -0.05)
    else:
        pcm, af = gpu.nfm_chain(u8, -0.05, block={"chain": None, "chain_blocks": 1024 * 150, "chain_small_blocks": 1024 * 5}[mode])
        if mode != "chain_small_blocks":
            assert gpu.last_ddc_kernel == "k_ddc_mfma" or mode == "chain_blocks"
    taps = gpu.nfm_taps(48000)
    for s in range(3):
        ps, pf = port.nfm_chain(u8[s], -0.05, taps)
        assert pf.size == af.shape[1] and pf.size >= 2 * 1024
        assert np.all(af[s, :2048] == 0)                               # fastagc's two-block latency
        assert relrms(af[s], pf) < TOL
        d = np.abs(pcm[s].astype(np.int32) - ps.astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 0.05


def test_nfm_fused_epilogue_is_bit_identical(gpu, monkeypatch):
    """The chain object's front end demodulates / limits in its reducer epilogue and leaves the samples at workgroup and kernel boundaries to
    k_nfm_demod_boundary; fastagc's peaks come from the de-emphasis kernel.  With CSDR_AMD_NFM_FUSE=0 the same object runs the separate passes over the
    decimated stream: both must give the SAME bits -- 19 channels (a partly filled 16-stream row), ragged block sizes (leading / trailing edge outputs of every
    length, blocks below the matrix-core kernel's minimum), an AGC block that is NOT the de-emphasis kernel's span."""
    n = 1024 * 700
    u8 = np.stack([to_u8(fm_signal(np.random.default_rng(6000 + s), n, dev=4e3 / 2.4e6, offset=0.03)) for s in range(19)])
    for agc_block, blocks in [(1024, [1024 * 300, 1024 * 150 + 512 * 2, 1024 * 249]), (512, [1024 * 120] * 5 + [1024 * 100])]:
        outs = []
        for fuse in ("1", "0"):
            monkeypatch.setenv("CSDR_AMD_NFM_FUSE", fuse)
            res = []
            for blk in (None, blocks):
                if blk is None:
                    res.append(gpu.nfm_chain(u8, -0.03, agc_block=agc_block))
                else:
                    res.append(gpu.nfm_chain(u8, -0.03, agc_block=agc_block, block=blk[1]))      # calls of blk[1] samples + a shorter last one
            outs.append(res)
        for a, b in zip(outs[0], outs[1]):
            assert a[0].shape == b[0].shape and np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
        d = np.abs(outs[0][0][0][:, :outs[0][1][0].shape[1]].astype(np.int32) - outs[0][1][0][:, :outs[0][0][0].shape[1]].astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 0.05                    # one call vs blocks: other tile boundaries, other summation order


# ---------------------------------------------------------------- f2 blocks
def test_f2_elementwise(gpu, port):
    rng = np.random.default_rng(41)
    x = (rng.normal(size=100003) + 1j * rng.normal(size=100003)).astype(c64)
    x[:4] = [0, 1, -1j, 1e-20]
    assert relrms(gpu.amdemod_cf(x), port.amdemod_cf(x)) <= TOL
    assert np.array_equal(gpu.amdemod_estimator_cf(x), port.amdemod_estimator_cf(x))          # no rounding freedom: two products and a sum
    assert np.array_equal(gpu.amdemod_estimator_cf(x, 0.9, 0.4), port.amdemod_estimator_cf(x, 0.9, 0.4))
    assert np.array_equal(gpu.realpart_cf(x), port.realpart_cf(x))
    assert relrms(gpu.logpower_cf(x[4:], -7.5), port.logpower_cf(x[4:], -7.5)) <= TOL
    for w in ("HAMMING", "BLACKMAN", "BOXCAR"):
        assert np.array_equal(gpu.precalculate_window(1024, w), port.precalculate_window(1024, w))


def test_f2_fmdemod_atan_streams_and_carry(gpu, port):
    rng = np.random.default_rng(42)
    x = crand(rng, 3 * 5000).reshape(3, 5000)
    y, lp = gpu.fmdemod_atan_cf(x, last_phase=[0.0, 0.5, -1.0], calls=3)
    for s, p0 in enumerate([0.0, 0.5, -1.0]):
        w, wl = port.fmdemod_atan_cf(x[s], p0)
        assert relrms(y[s], w) <= TOL and abs(lp[s] - wl) <= 1e-6


def test_f2_dcblock_and_fastdcblock(gpu, port):
    rng = np.random.default_rng(43)
    x = (rng.uniform(-1, 1, 2 * 70001) + 0.25).astype(f32).reshape(2, 70001)
    for a in (0.0, 0.95):
        y, st = gpu.dcblock_ff(x, a, state=[0.1, 0.2, -0.3, 0.05], calls=3)
        for s, st0 in enumerate([(0.1, 0.2), (-0.3, 0.05)]):
            w, ws = port.dcblock_ff(x[s], a, st0)
            assert relrms(y[s], w) <= TOL and np.allclose(st[s], ws, atol=1e-5)
    y, ld = gpu.fastdcblock_ff(x, 1024, last_dc=[0.1, -0.2], calls=2)
    for s, l0 in enumerate([0.1, -0.2]):
        w, wl = port.fastdcblock_ff(x[s], 1024, l0)
        assert y[s].size == w.size and relrms(y[s], w) <= TOL and abs(ld[s] - wl) <= 1e-6


def test_f2_agc(gpu, port):
    rng = np.random.default_rng(44)
    sig = (rng.uniform(-1, 1, 3 * 20000) * np.repeat(rng.uniform(0.01, 1, 600), 100)).astype(f32).reshape(3, 20000)
    sig[1, 500:520] = 0
    for kw in ({}, dict(hang_time=20, reference=0.5, attack_rate=0.05, decay_rate=0.001, max_gain=100.0, attack_wait=5, filter_alpha=0.99)):
        y, g = gpu.agc_ff(sig, 1024, **kw)
        for s in range(3):
            w, wg = port.agc_ff(sig[s], 1024, **kw)
            assert relrms(y[s], w) <= TOL and abs(g[s] - wg) <= 1e-4 * max(1.0, abs(wg))


@pytest.mark.parametrize("fft,every", [(1024, 300), (1024, 1024), (256, 1000), (4096, 4000)])
def test_f2_fft_cc(gpu, port, fft, every):
    rng = np.random.default_rng(45)
    x = crand(rng, 20000)
    want = port.fft_cc(x, fft, every, "HAMMING")
    for calls in (1, 3):
        got = gpu.fft_cc(x, fft, every, "HAMMING", calls=calls)
        assert got.size == want.size and relrms(got, want) <= TOL


# ---------------------------------------------------------------- f3: IMA ADPCM, bit exact
def test_f3_adpcm(gpu, port):
    rng = np.random.default_rng(46)
    x = np.stack([(8000 * np.sin(np.arange(30000) * 0.01 * (s + 1)) + rng.integers(-3000, 3000, 30000)) for s in range(5)]).astype(np.int16)
    x[4] = rng.integers(-32768, 32768, 30000)
    st0 = np.array([[0, 0], [5, -100], [88, 32767], [0, -32768], [40, 1]], np.int32)
    for calls in (1, 3):
        y, st = gpu.encode_ima_adpcm_i16_u8(x, st0, calls=calls)
        for s in range(5):
            w, ws = port.encode_ima_adpcm_i16_u8(x[s], tuple(st0[s]))
            assert np.array_equal(y[s], w) and tuple(st[s]) == ws
        z, zt = gpu.decode_ima_adpcm_u8_i16(y, st0, calls=calls)
        for s in range(5):
            w, ws = port.decode_ima_adpcm_u8_i16(y[s], tuple(st0[s]))
            assert np.array_equal(z[s], w) and tuple(zt[s]) == ws
    # more streams than a workgroup (130: a partly filled third one), lengths of every residue, random start states
    for n in (128 * 5, 128 * 3 + 2, 1000 + 4 * 7, 4 * 33):
        xs = rng.integers(-20000, 20000, (130, n)).astype(np.int16)
        stn = np.stack([rng.integers(0, 89, 130), rng.integers(-32768, 32768, 130)], axis=1).astype(np.int32)
        y, st = gpu.encode_ima_adpcm_i16_u8(xs, stn)
        z, zt = gpu.decode_ima_adpcm_u8_i16(y, stn)
        for s in (0, 63, 64, 129):
            w, ws = port.encode_ima_adpcm_i16_u8(xs[s], tuple(stn[s]))
            assert np.array_equal(y[s], w) and tuple(st[s]) == ws
            w, ws = port.decode_ima_adpcm_u8_i16(y[s], tuple(stn[s]))
            assert np.array_equal(z[s], w) and tuple(zt[s]) == ws
    # the encoder that stages through LDS (rows on 16-byte boundaries): lengths of every residue of its 128-sample chunks and 8-sample pieces (odd: the last sample
    # is dropped, ima_adpcm.c:154-163), two calls with the state carried between them, 130 streams, padded pitches
    L = gpu.L
    for n in (2, 6, 14, 128, 130, 254, 1000 + 2, 3 * 128 + 77):
        S = 130
        xs = rng.integers(-25000, 25000, (S, n)).astype(np.int16)
        stn = np.stack([rng.integers(0, 89, S), rng.integers(-32768, 32768, S)], axis=1).astype(np.int32)
        ip = (n + 8 + 7) // 8 * 8; op = (n // 2 + 16 + 15) // 16 * 16
        xp = np.zeros((S, ip), np.int16); xp[:, :n] = xs
        di = gpu.upload(xp); do = gpu.alloc(S * op); ds = gpu.upload(stn.reshape(-1))
        first = (n // 2) & ~31 if n >= 128 else n                # a second call starts on a 16-byte boundary of both rows
        gpu.check(L.csdr_amd_encode_ima_adpcm_i16_u8(gpu.h, di.ptr, do.ptr, S, first, ip, op, ds.ptr), "adpcm encode")
        if first < n:
            gpu.check(L.csdr_amd_encode_ima_adpcm_i16_u8(gpu.h, di.at(2 * first), do.at(first // 2), S, n - first, ip, op, ds.ptr), "adpcm encode")
        y = gpu.download(do, np.uint8, S * op).reshape(S, op)[:, :n // 2]
        so = gpu.download(ds, np.int32, 2 * S).reshape(S, 2)
        for s_ in (0, 1, 63, 64, 129):
            w, ws = port.encode_ima_adpcm_i16_u8(xs[s_], tuple(stn[s_]))
            assert np.array_equal(y[s_], w) and tuple(so[s_]) == ws, (n, s_)
    # encode -> decode round trip tracks the input (size-independent property at a larger size)
    big = (12000 * np.sin(np.arange(1 << 20) * 0.002)).astype(np.int16)
    enc, _ = gpu.encode_ima_adpcm_i16_u8(big)
    dec, _ = gpu.decode_ima_adpcm_u8_i16(enc)
    assert np.abs(dec.astype(np.int32) - big.astype(np.int32))[64:].max() < 400
    fftrows = (rng.uniform(-120, 10, 37 * 2048) + 20 * np.sin(np.arange(37 * 2048) * 0.05)).astype(f32)
    fftrows[5] = 400.0; fftrows[6] = -400.0; fftrows[7] = 3e7
    assert np.array_equal(gpu.compress_fft_adpcm_f_u8(fftrows, 2048), port.compress_fft_adpcm_f_u8(fftrows, 2048))


def test_adpcm_decode_scans_bit_exact(gpu, port):
    """decode_ima_adpcm_u8_i16 as two parallel scans of clamped-add maps (adpcm.hip: k_adpcm_decode_scan) against the serial definition (ima_adpcm.c:109-131,
    165-174), bit for bit: EVERY start index 0..88 with predictors at both rails and in between (one stream each), random codes and runs that pin the index /
    the predictor at their clamps, lengths around the 4096-byte chunk and the 16-byte lane, several calls with the state carried, an unaligned row pitch."""
    rng = np.random.default_rng(89)
    starts = [(i, p) for i in range(89) for p in (-32768, -1234, 0, 32767)]
    S = len(starts)
    for n in (1, 7, 8, 9, 15, 16, 17, 4095, 4096, 4097, 9000, 3 * 4096 + 13):
        x = rng.integers(0, 256, (S, n), dtype=np.uint8)
        x[::3, : n // 2] = 0x77                                     # +7, +7, ...: index and predictor run into their upper clamps
        x[1::3, : n // 3] = 0xff                                    # -7, -7, ...: the lower rail
        st = np.array(starts, np.int32)
        got, gs = gpu.decode_ima_adpcm_u8_i16(x, st, calls=1 if n < 100 else 3)
        for s in range(0, S, 7):
            want, ws = port.decode_ima_adpcm_u8_i16(x[s], tuple(int(v) for v in starts[s]))
            assert np.array_equal(got[s], want), (n, s)
            assert [int(v) for v in gs[s]] == [int(v) for v in ws], (n, s)
    # an odd pitch (rows not 8-byte aligned): the byte-wise load path
    L = gpu.L
    n, S2 = 4099, 5
    x = rng.integers(0, 256, (S2, n), dtype=np.uint8)
    pitch_in, pitch_out = n + 3, 2 * n + 1
    xi = np.zeros((S2, pitch_in), np.uint8); xi[:, :n] = x
    di = gpu.upload(xi); do = gpu.alloc(2 * S2 * pitch_out + 64); ds = gpu.upload(np.zeros(2 * S2, np.int32))
    import ctypes as C
    gpu.check(L.csdr_amd_decode_ima_adpcm_u8_i16(gpu.h, di.ptr, do.ptr, S2, n, pitch_in, pitch_out, ds.ptr), "decode")
    y = gpu.download(do, np.int16, S2 * pitch_out).reshape(S2, pitch_out)[:, :2 * n]
    for s in range(S2):
        assert np.array_equal(y[s], port.decode_ima_adpcm_u8_i16(x[s])[0])
    # many short streams: the one-lane-per-stream kernel is chosen (adpcm.hip's estimate); a start index outside the step table is clamped to it (the reference
    # would read past its table, ima_adpcm.c:110) -- on either kernel
    for S3, n3 in ((16384, 100), (8, 5000)):
        x = rng.integers(0, 256, (S3, n3), dtype=np.uint8)
        st = np.zeros((S3, 2), np.int32); st[:, 0] = rng.integers(0, 89, S3); st[:, 1] = rng.integers(-32768, 32768, S3)
        st[0] = (200, 5); st[1] = (-7, 5)
        got, gs = gpu.decode_ima_adpcm_u8_i16(x, st)
        for s_ in list(range(0, S3, 97)) + [1]:
            i0 = min(max(int(st[s_, 0]), 0), 88)
            want, ws = port.decode_ima_adpcm_u8_i16(x[s_], (i0, int(st[s_, 1])))
            assert np.array_equal(got[s_], want) and [int(v) for v in gs[s_]] == [int(v) for v in ws], (S3, s_)
