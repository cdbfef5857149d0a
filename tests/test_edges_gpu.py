"""Edge cases of the device batch API on the GPU: empty and one-element inputs, lengths that are not multiples of any tile,
stream counts that do not fill a 16/64-stream group, inputs shorter than a filter, minimum block sizes, odd ADPCM lengths,
error reporting.  Everything against the CPU oracle on identical input (tests/conftest.py::port)."""
import numpy as np
import pytest
from oracle import relrms

pytestmark = pytest.mark.gpu
c64, f32 = np.complex64, np.float32
TOL = 1e-5


@pytest.fixture(scope="module")
def gpu():
    import torch  # noqa: F401
    import csdr_amd
    ctx = csdr_amd.Context(0)
    yield ctx
    ctx.close()


def crand(rng, n):
    return (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(c64)


def test_empty_and_single_element(gpu, port):
    e8, ef, ec = np.zeros(0, np.uint8), np.zeros(0, f32), np.zeros(0, c64)
    assert gpu.convert_u8_f(e8).size == 0 and gpu.convert_f_s16(ef).size == 0 and gpu.limit_ff(ef).size == 0
    assert gpu.amdemod_cf(ec).size == 0 and gpu.logpower_cf(ec).size == 0 and gpu.realpart_cf(ec).size == 0
    one = np.array([0.3 - 0.7j], c64)
    assert np.array_equal(gpu.shift_addition_cc(one, 0.1)[0].view(np.uint32), port.shift_addition_cc(one, 0.1)[0].view(np.uint32))
    assert relrms(gpu.fmdemod_quadri_cf(one, np.array([0.2 + 0.1j], c64))[0], port.fmdemod_quadri_cf(one, (0.2, 0.1))[0]) <= TOL
    assert relrms(gpu.fmdemod_atan_cf(one)[0], port.fmdemod_atan_cf(one)[0]) <= TOL
    r1 = np.array([0.25], f32)
    assert relrms(gpu.dcblock_ff(r1)[0], port.dcblock_ff(r1)[0]) <= TOL
    assert relrms(gpu.agc_ff(r1)[0], port.agc_ff(r1)[0]) <= TOL
    assert relrms(gpu.deemphasis_wfm_ff(r1, 50e-6, 48000)[0], port.deemphasis_wfm_ff(r1, 50e-6, 48000)[0]) <= TOL
    assert gpu.encode_ima_adpcm_i16_u8(np.array([1234], np.int16))[0].size == 0          # an odd last sample is dropped (ima_adpcm.c:157)
    assert gpu.decode_ima_adpcm_u8_i16(e8)[0].size == 0


@pytest.mark.parametrize("n", [1, 3, 63, 64, 65, 255, 257, 1023, 1025, 4099])
def test_odd_lengths_elementwise(gpu, port, n):
    rng = np.random.default_rng(n)
    u8 = rng.integers(0, 256, n, dtype=np.uint8)
    assert np.array_equal(gpu.convert_u8_f(u8).view(np.uint32), port.convert_u8_f(u8).view(np.uint32))
    x = rng.uniform(-1.3, 1.3, n).astype(f32)
    assert np.array_equal(gpu.convert_f_s16(x), port.convert_f_s16(x)) and np.array_equal(gpu.convert_f_u8(x), port.convert_f_u8(x))
    assert np.array_equal(gpu.limit_ff(x, 0.9).view(np.uint32), port.limit_ff(x, 0.9).view(np.uint32))
    c = crand(rng, n)
    for name in ("shift_addition_cc", "shift_math_cc", "shift_addfast_cc", "shift_unroll_cc"):
        m = n // 4 * 4 if name == "shift_addfast_cc" else n       # the reference's addfast loop only covers whole groups of four (libcsdr.c:406-434)
        if m:
            assert relrms(getattr(gpu, name)(c, 0.123)[0][:m], getattr(port, name)(c, 0.123)[0][:m]) <= TOL, name
    assert relrms(gpu.fmdemod_quadri_cf(c)[0], port.fmdemod_quadri_cf(c)[0]) <= TOL
    assert relrms(gpu.amdemod_cf(c), port.amdemod_cf(c)) <= TOL
    assert relrms(gpu.dcblock_ff(x)[0], port.dcblock_ff(x)[0]) <= TOL
    assert relrms(gpu.agc_ff(x, 64)[0], port.agc_ff(x, 64)[0]) <= TOL


def test_fir_shorter_than_taps_and_exact_fit(gpu, port):
    rng = np.random.default_rng(2)
    taps = port.firdes_lowpass_f(79, 0.05)
    assert gpu.fir_decimate_cc(crand(rng, 78), 10, taps).size == 0 and port.fir_decimate_cc(crand(rng, 78), 10, taps).size == 0
    for n in (79, 80, 88, 89, 79 + 10 * 255, 79 + 10 * 256, 79 + 10 * 257):           # one output, tile boundaries of the polyphase kernel
        x = crand(rng, n)
        a, b = gpu.fir_decimate_cc(x, 10, taps), port.fir_decimate_cc(x, 10, taps)
        assert a.size == b.size and relrms(a, b) <= TOL, n
    # other decimations / tap counts (polyphase configurations and the generic fallback)
    for D, nt in ((1, 31), (2, 9), (3, 101), (7, 255), (16, 127), (50, 801), (100, 1601)):
        t = port.firdes_lowpass_f(nt, 0.4 / D)
        x = crand(rng, nt + D * 700 + 3)
        a, b = gpu.fir_decimate_cc(x, D, t), port.fir_decimate_cc(x, D, t)
        assert a.size == b.size and relrms(a, b) <= TOL, (D, nt)


def test_many_short_streams_and_pitch(gpu, port):
    rng = np.random.default_rng(3)
    x = np.stack([crand(rng, 300) for _ in range(130)])                # 130 streams: not a multiple of 64
    taps = port.firdes_lowpass_f(31, 0.1)
    y = gpu.fir_decimate_cc(x, 4, taps)
    for s in (0, 63, 64, 129):
        assert relrms(y[s], port.fir_decimate_cc(x[s], 4, taps)) <= TOL
    d, _ = gpu.fmdemod_quadri_cf(x)
    for s in (0, 129):
        assert relrms(d[s], port.fmdemod_quadri_cf(x[s])[0]) <= TOL
    r = rng.uniform(-1, 1, (130, 300)).astype(f32)
    e, _ = gpu.deemphasis_wfm_ff(r, 50e-6, 48000)
    for s in (0, 77, 129):
        assert relrms(e[s], port.deemphasis_wfm_ff(r[s], 50e-6, 48000)[0]) <= TOL


@pytest.mark.parametrize("n_streams", [1, 15, 17, 65])
def test_wfm_stream_counts(gpu, port, n_streams):
    """stream counts around the 16-stream MFMA group and the 64-stream block"""
    from tests_helpers import wfm_signal_u8
    taps = port.firdes_lowpass_f(79, 0.05)
    base = [wfm_signal_u8(100 + s, 16384 * 3) for s in range(min(n_streams, 3))]
    u8 = np.stack([base[s % len(base)] for s in range(n_streams)])
    s16, af = gpu.wfm_chain(u8, -0.085, 10, taps)
    want = [port.wfm_chain(b, -0.085, 10, taps) for b in base]
    for s in sorted({0, n_streams // 2, n_streams - 1}):
        ps, pf = want[s % len(base)]
        m = min(pf.size, af.shape[1])
        assert m >= 16384 * 3 // 50 - 8 and relrms(af[s, :m], pf[:m]) <= TOL
        assert np.abs(s16[s, :m].astype(np.int32) - ps[:m]).max() <= 1


@pytest.mark.parametrize("pitch_pad,block", [(0, None), (0, 16384 * 5), (16, None), (16, 16384 * 5), (48, 1024 * 7), (112, 1024 * 33), (0, 1024 * 3)])
def test_wfm_chain_kernel_pitches_and_blocks(gpu, port, pitch_pad, block):
    """The ONE chain kernel (k_wfm_mfma_seq) on the same input with row pitches that are / are not multiples of 128 bytes (any 16-byte-aligned pitch is
    taken: round 2 needed the quad kernel for those), in one call and in blocks whose audio start is not a multiple of 4 (partial tiles at both ends of
    every call, windows that start in the previous block's history, calls shorter than one segment); 19 streams = one full and one ragged 16-stream block."""
    from tests_helpers import wfm_signal_u8
    taps = port.firdes_lowpass_f(79, 0.05)
    n = 16384 * 15
    base = [wfm_signal_u8(300 + s, n) for s in range(3)]
    u8 = np.stack([base[s % 3] for s in range(19)])
    s16, af = gpu.wfm_chain(u8, -0.085, 10, taps, block=block, pitch_pad=pitch_pad)
    assert gpu.last_wfm_kernel == "k_wfm_mfma_seq"
    want = [port.wfm_chain(b, -0.085, 10, taps) for b in base]
    for s in (0, 1, 2, 15, 16, 18):
        ps, pf = want[s % 3]
        m = min(pf.size, af.shape[1])
        assert m >= n // 50 - 8 and relrms(af[s, :m], pf[:m]) <= TOL
        assert np.abs(s16[s, :m].astype(np.int32) - ps[:m]).max() <= 1


def test_wfm_ragged_last_block(gpu, port):
    """a stream whose last block is not a multiple of 1024 samples (it ends the stream): the fetch is masked at the row's last 16-byte piece"""
    from tests_helpers import wfm_signal_u8
    taps = port.firdes_lowpass_f(79, 0.05)
    for n in (16384 * 4 + 1000, 16384 * 4 + 8, 16384 * 2 + 777):
        u8 = np.stack([wfm_signal_u8(40 + s, n) for s in range(3)])
        s16, af = gpu.wfm_chain(u8, -0.085, 10, taps, block=16384)
        for s in range(3):
            ps, pf = port.wfm_chain(u8[s], -0.085, 10, taps)
            m = min(pf.size, af.shape[1])
            assert m >= n // 50 - 8 and relrms(af[s, :m], pf[:m]) <= TOL
            assert np.abs(s16[s, :m].astype(np.int32) - ps[:m]).max() <= 1


@pytest.mark.parametrize("D,F,L", [(8, 6, 63), (10, 4, 79), (12, 5, 47), (6, 8, 95)])
def test_wfm_other_decimations(gpu, port, D, F, L):
    """tile strides other than the benchmark's 400 bytes (8 D F): the ring and the weight set are built for any supported (D, F, taps)"""
    from tests_helpers import wfm_signal_u8
    taps = port.firdes_lowpass_f(L, 0.5 / D)
    n = 16384 * 6
    u8 = np.stack([wfm_signal_u8(60 + s, n) for s in range(2)])
    s16, af = gpu.wfm_chain(u8, -0.085, D, taps, frac_rate=F, block=16384 * 2)
    assert gpu.last_wfm_kernel == "k_wfm_mfma_seq"
    for s in range(2):
        ps, pf = port.wfm_chain(u8[s], -0.085, D, taps, frac_rate=F)
        m = min(pf.size, af.shape[1])
        assert m >= n // (D * F) - 8 and relrms(af[s, :m], pf[:m]) <= TOL


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_wfm_random_block_schedules(gpu, port, seed):
    """csdr_amd_wfm_process over RANDOM call sizes (multiples of 1024 between 1 and 90 chunks, a ragged last call), stream counts and row pitches: every call's
    first / last tile is partial somewhere, windows start in the history, the seed table is rebuilt on the way; the concatenated audio equals the oracle's stream."""
    import ctypes as C
    from tests_helpers import wfm_signal_u8
    rng = np.random.default_rng(seed)
    L = gpu.L
    S = int(rng.integers(1, 40))
    sizes = [1024 * int(rng.integers(1, 91)) for _ in range(int(rng.integers(3, 9)))] + [int(rng.integers(1, 1024)) * 2]
    n = sum(sizes)
    pad = 16 * int(rng.integers(0, 9))
    taps = port.firdes_lowpass_f(79, 0.05)
    base = [wfm_signal_u8(900 + seed * 10 + k, n) for k in range(min(S, 3))]
    pitch = (2 * n + 15) // 16 * 16 + pad
    xx = np.zeros((S, pitch), np.uint8)
    for s in range(S):
        xx[s, :2 * n] = base[s % len(base)]
    w = L.csdr_amd_wfm_create(gpu.h, S, -0.085, 10, taps.ctypes.data_as(C.c_void_p), taps.size, 5, 50e-6, 48000, max(sizes))
    assert w, gpu.err()
    di = gpu.upload(xx)
    apitch = n // 50 + 64
    ds = gpu.alloc(2 * S * apitch); df = gpu.alloc(4 * S * apitch)
    pos = na = 0
    for k in sizes:
        got = L.csdr_amd_wfm_process(w, di.at(2 * pos), pitch, k, ds.at(2 * na), df.at(4 * na), apitch)
        assert got >= 0, gpu.err()
        pos += k; na += got
    s16 = gpu.download(ds, np.int16, S * apitch).reshape(S, apitch)[:, :na]
    af = gpu.download(df, f32, S * apitch).reshape(S, apitch)[:, :na]
    L.csdr_amd_wfm_destroy(w)
    want = [port.wfm_chain(b, -0.085, 10, taps) for b in base]
    for s in sorted({0, S // 2, S - 1}):
        ps, pf = want[s % len(base)]
        m = min(pf.size, na)
        assert m >= n // 50 - 8 and relrms(af[s, :m], pf[:m]) <= TOL, (seed, s, sizes)
        assert np.abs(s16[s, :m].astype(np.int32) - ps[:m]).max() <= 1


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_wfm_random_block_schedules_s16_only(gpu, port, seed):
    """The same walk over random call sizes the way a streaming caller runs it -- s16 only (no float audio), every call into a 16-byte-aligned row buffer of its
    own: the path on which the loader waves collect the finished lines in registers and store 4 KiB per stream at once (k_wfm_mfma_seq, "Stores").  Call sizes make
    every call's first audio sample fall anywhere in a 128-byte line of its output row; long calls fill and flush the registers several times per segment, short
    ones never fill them; stream counts that are not multiples of 16 leave rows of the last workgroup without an owner.  +-1 LSB against the oracle's stream."""
    import ctypes as C
    from tests_helpers import wfm_signal_u8
    rng = np.random.default_rng(seed)
    L = gpu.L
    S = int(rng.integers(1, 40))
    head = [1024 * int(rng.integers(1, 91)) for _ in range(int(rng.integers(3, 7)))] + [1024 * int(rng.integers(300, 420))]
    rng.shuffle(head)
    sizes = head + [int(rng.integers(1, 1024)) * 2]
    n = sum(sizes)
    taps = port.firdes_lowpass_f(79, 0.05)
    base = [wfm_signal_u8(1900 + seed * 10 + k, n) for k in range(min(S, 3))]
    pitch = (2 * n + 15) // 16 * 16
    xx = np.zeros((S, pitch), np.uint8)
    for s in range(S):
        xx[s, :2 * n] = base[s % len(base)]
    w = L.csdr_amd_wfm_create(gpu.h, S, -0.085, 10, taps.ctypes.data_as(C.c_void_p), taps.size, 5, 50e-6, 48000, max(sizes))
    assert w, gpu.err()
    di = gpu.upload(xx)
    apitch = (max(sizes) // 50 + 64 + 7) // 8 * 8
    ds = gpu.alloc(2 * S * apitch)
    out = np.zeros((S, n // 50 + 64), np.int16)
    pos = na = 0
    for k in sizes:
        got = L.csdr_amd_wfm_process(w, di.at(2 * pos), pitch, k, ds.ptr, None, apitch)
        assert got >= 0, gpu.err()
        if got:
            out[:, na:na + got] = gpu.download(ds, np.int16, S * apitch).reshape(S, apitch)[:, :got]
        pos += k; na += got
    L.csdr_amd_wfm_destroy(w)
    want = [port.wfm_chain(b, -0.085, 10, taps) for b in base]
    for s in sorted({0, S // 2, S - 1}):
        ps, _ = want[s % len(base)]
        m = min(ps.size, na)
        assert m >= n // 50 - 8, (seed, s, sizes)
        assert np.abs(out[s, :m].astype(np.int32) - ps[:m]).max() <= 1, (seed, s, sizes)


def test_wfm_float_audio_into_aligned_rows(gpu, port):
    """s16 AND float audio into 16-byte-aligned rows (pitch a multiple of 8 samples), three calls: the loader waves store every finished line at once (16-byte s16
    pieces + two float4 per lane) instead of holding it -- the third way the audio can leave k_wfm_mfma_seq besides the register-held lines (s16 only) and the
    sample-by-sample path (unaligned rows)."""
    import ctypes as C
    from tests_helpers import wfm_signal_u8
    L = gpu.L
    S, sizes = 19, [1024 * 37, 1024 * 64, 1024 * 11 + 640]
    n = sum(sizes)
    taps = port.firdes_lowpass_f(79, 0.05)
    base = [wfm_signal_u8(2900 + k, n) for k in range(3)]
    pitch = (2 * n + 15) // 16 * 16
    xx = np.zeros((S, pitch), np.uint8)
    for s_ in range(S):
        xx[s_, :2 * n] = base[s_ % 3]
    w = L.csdr_amd_wfm_create(gpu.h, S, -0.085, 10, taps.ctypes.data_as(C.c_void_p), taps.size, 5, 50e-6, 48000, max(sizes))
    assert w, gpu.err()
    di = gpu.upload(xx)
    apitch = (max(sizes) // 50 + 64 + 7) // 8 * 8
    ds = gpu.alloc(2 * S * apitch); df = gpu.alloc(4 * S * apitch)
    o16 = np.zeros((S, n // 50 + 64), np.int16); of = np.zeros((S, n // 50 + 64), f32)
    pos = na = 0
    for k in sizes:
        got = L.csdr_amd_wfm_process(w, di.at(2 * pos), pitch, k, ds.ptr, df.ptr, apitch)
        assert got >= 0, gpu.err()
        if got:
            o16[:, na:na + got] = gpu.download(ds, np.int16, S * apitch).reshape(S, apitch)[:, :got]
            of[:, na:na + got] = gpu.download(df, f32, S * apitch).reshape(S, apitch)[:, :got]
        pos += k; na += got
    L.csdr_amd_wfm_destroy(w)
    for s_ in (0, 7, 18):
        ps, pf = port.wfm_chain(base[s_ % 3], -0.085, 10, taps)
        m = min(pf.size, na)
        assert m >= n // 50 - 8 and relrms(of[s_, :m], pf[:m]) <= TOL
        assert np.abs(o16[s_, :m].astype(np.int32) - ps[:m]).max() <= 1


@pytest.mark.parametrize("rate", [0.25, 0.05, -0.3141])
def test_wfm_other_shift_rates(gpu, port, rate):
    """Shift rates other than the benchmark's, including 0.25 and 0.05 for which the reference's float phasor recurrence drifts by up to
    4e-5 per chunk away from C_m D^k: the FM demodulator is insensitive to that common-mode error (both FIR outputs of a pair share it)."""
    from tests_helpers import wfm_signal_u8
    taps = port.firdes_lowpass_f(79, 0.05)
    n = 16384 * 8
    u8 = np.stack([wfm_signal_u8(350 + s, n, offset=-rate) for s in range(2)])
    s16, af = gpu.wfm_chain(u8, rate, 10, taps)
    assert gpu.last_wfm_kernel == "k_wfm_mfma_seq"
    for s in range(2):
        ps, pf = port.wfm_chain(u8[s], rate, 10, taps)
        m = min(pf.size, af.shape[1])
        assert m >= n // 50 - 8 and relrms(af[s, :m], pf[:m]) <= TOL


def test_wfm_minimum_blocks(gpu, port):
    """1024-sample blocks (the smallest legal block): every tile straddles a block boundary, history path only"""
    from tests_helpers import wfm_signal_u8
    taps = port.firdes_lowpass_f(79, 0.05)
    u8 = wfm_signal_u8(7, 1024 * 23)[None, :]
    s16, af = gpu.wfm_chain(u8, -0.085, 10, taps, block=1024)
    ps, pf = port.wfm_chain(u8[0], -0.085, 10, taps)
    m = min(pf.size, af.shape[1])
    assert m >= 1024 * 23 // 50 - 8 and relrms(af[0, :m], pf[:m]) <= TOL


def test_error_reporting(gpu):
    import csdr_amd
    with pytest.raises(csdr_amd.CsdrAmdError):
        gpu.fir_decimate_cc(np.zeros(100, c64), 0, np.ones(3, f32))                     # decimation 0
    with pytest.raises(csdr_amd.CsdrAmdError):
        gpu.fft_cc(np.zeros(100, c64), 1000, 10)                                        # not a power of two
    f = gpu.L.csdr_amd_fftfilt_create(gpu.h, 1000, None, 3, 1, 1)                         # fft_size not a power of two
    assert not f and gpu.err()
