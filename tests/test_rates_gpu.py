"""A shift rate PER STREAM and live retune in the fused chain objects: csdr_amd_ddc_create_rates / _set_rate, csdr_amd_nfm_create_rates /
_set_rate against the oracle's stage-by-stage stream model, one (stream, shift_rate) pair at a time -- the reference's unit of work (`csdr shift_addition_cc --fifo`,
csdr.c:881-923; ddcd's per-client chains, ddcd_old.h:51-61).  Gates: relative RMS <= 1e-5 on complex samples, +-1 LSB on s16 audio."""
import numpy as np
import pytest
from oracle import relrms
from tests_helpers import nfm_signal_u8, wfm_signal_u8
import verify_configs as vc

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


def oracle_front(port, u8, rates_at, D, taps):
    """rates_at: [(first sample, rate), ...]: the rate changes at those samples (multiples of 1024), the phase carries over (csdr.c:896-923)."""
    xf = port.convert_u8_f(u8).view(c64)
    parts, ph = [], 0.0
    for i, (pos, r) in enumerate(rates_at):
        end = rates_at[i + 1][0] if i + 1 < len(rates_at) else xf.size
        y, ph = port.shift_addition_cc(xf[pos:end], r, phase=ph)
        parts.append(y)
    return port.fir_decimate_cc(np.concatenate(parts), D, taps)


RATES = [0.11, -0.4321, 0.05, 0.25, -0.05, 0.3, -0.2718, 0.0123, 0.499, -0.3333, 0.085]


@pytest.mark.parametrize("n,S", [(1024 * 517 + 0, 5), (1024 * 120, 11), (1024 * 40, 3)])
def test_ddc_rates_single_call(gpu, port, n, S):
    """One call; 517 chunks = 20.2 periods of 64 tiles -> two column groups, the last column partial (and columns behind the block's end); 120 chunks = 5 columns;
    40 chunks = one column.  Rates include the drifting ones (0.05, 0.25: per-stream correction rows)."""
    D, L = 50, 801
    taps = port.firdes_lowpass_f(L, 0.5 / D)
    rates = np.array(RATES[:S], f32)
    u8 = np.stack([nfm_signal_u8(900 + s, n, offset=-float(rates[s])) for s in range(S)])
    y = gpu.ddc_u8(u8, rates, D, taps)
    assert gpu.last_ddc_kernel == "k_ddc_mfma"
    for s in range(S):
        want = oracle_front(port, u8[s], [(0, float(rates[s]))], D, taps)
        assert y.shape[1] == want.size
        assert relrms(y[s], want) <= TOL, "stream %d rate %g: %g" % (s, rates[s], relrms(y[s], want))


def test_ddc_rates_equal_rates_match_shared_rate_object(gpu, port):
    """All streams at ONE rate through the per-stream object against the shared-rate object: the same integer sums and seeds, only the order of the float
    additions over the K-ranges' shares can differ."""
    D, L, n, S = 50, 801, 1024 * 300, 4
    taps = port.firdes_lowpass_f(L, 0.5 / D)
    u8 = np.stack([nfm_signal_u8(950 + s, n, offset=-0.11) for s in range(S)])
    a = gpu.ddc_u8(u8, 0.11, D, taps)
    b = gpu.ddc_u8(u8, np.full(S, 0.11, f32), D, taps)
    assert a.shape == b.shape
    for s in range(S):
        assert relrms(b[s], a[s]) <= 2e-7, relrms(b[s], a[s])


@pytest.mark.parametrize("sizes", [[1024 * 64, 1024 * 64, 1024 * 130, 1024 * 7], [1024 * 200, 1024 * 33]])
def test_ddc_rates_streaming_blocks(gpu, port, sizes):
    """Consecutive calls of different sizes (seed tables switched / regenerated on the side stream, history, output index), a ragged last call on the plain kernel."""
    D, L, S = 50, 801, 6
    n = sum(sizes) + 1024 * 2 + 346
    taps = port.firdes_lowpass_f(L, 0.5 / D)
    rates = np.array(RATES[3:3 + S], f32)
    u8 = np.stack([nfm_signal_u8(1000 + s, n, offset=-float(rates[s])) for s in range(S)])
    y = gpu.ddc_u8(u8, rates, D, taps, block=sizes + [1024 * 2])
    assert "k_ddc_mfma" in gpu.ddc_kernels
    for s in range(S):
        want = oracle_front(port, u8[s], [(0, float(rates[s]))], D, taps)
        assert y.shape[1] == want.size and relrms(y[s], want) <= TOL, "stream %d: %g" % (s, relrms(y[s], want))


def test_ddc_retune_between_calls(gpu, port):
    """csdr_amd_ddc_set_rate between calls = `shift_addition_cc --fifo` (csdr.c:881-923): new rate from the next block's first sample, starting_phase carried.
    Stream 0 is retuned twice (once to a drifting rate), stream 2 once, stream 1 never; the first outputs after a retune straddle two rates."""
    D, L, S = 50, 801, 3
    sizes = [1024 * 70, 1024 * 90, 1024 * 64, 1024 * 80]
    n = sum(sizes)
    taps = port.firdes_lowpass_f(L, 0.5 / D)
    rates0 = np.array([0.11, -0.2, 0.3], f32)
    retunes = {1: [(0, 0.05)], 2: [(2, -0.123)], 3: [(0, -0.31)]}
    pos = np.cumsum([0] + sizes)
    plan = {0: [(0, 0.11), (int(pos[1]), 0.05), (int(pos[3]), -0.31)], 1: [(0, -0.2)], 2: [(0, 0.3), (int(pos[2]), -0.123)]}
    # every stream's signal follows its tuning: a narrow-band FM signal at -rate in every stretch
    u8 = np.stack([np.concatenate([nfm_signal_u8(1100 + 10 * s + i, (plan[s][i + 1][0] if i + 1 < len(plan[s]) else n) - a, offset=-r)
                                   for i, (a, r) in enumerate(plan[s])]) for s in range(S)])
    y = gpu.ddc_u8(u8, rates0, D, taps, block=sizes, retunes=retunes)
    for s in range(S):
        want = oracle_front(port, u8[s], plan[s], D, taps)
        assert y.shape[1] == want.size
        e = np.abs(y[s] - want) / np.sqrt(np.mean(np.abs(want) ** 2))
        assert relrms(y[s], want) <= TOL and e.max() < 1e-4, "stream %d: rms %g max %g at %d" % (s, relrms(y[s], want), e.max(), int(e.argmax()))


def test_nfm_rates_and_retune(gpu, port):
    """The NFM chain object with a rate per channel: s16 audio +-1 LSB against the oracle's eight stages per channel; then the same with a retune in mid-stream
    (the oracle retunes its shift stage at that sample)."""
    D, S = 50, 5
    sizes = [1024 * 256, 1024 * 200]
    n = sum(sizes)
    rates = np.array([-0.05, 0.11, 0.25, -0.3456, 0.2], f32)
    u8 = np.stack([nfm_signal_u8(1200 + s, n, offset=-float(rates[s])) for s in range(S)])
    nfm_taps = gpu.nfm_taps(48000)
    pcm, _ = gpu.nfm_chain(u8, rates)
    assert gpu.last_ddc_kernel == "k_ddc_mfma"
    for s in range(S):
        ps, _ = port.nfm_chain(u8[s], float(rates[s]), nfm_taps, D, 0.005, 1024)
        assert pcm.shape[1] == ps.size and vc.s16_diff(pcm[s], ps).max() <= 1, "channel %d" % s
    # in blocks, bit identical to the single call
    pcm2, _ = gpu.nfm_chain(u8, rates, block=sizes)
    m = min(pcm.shape[1], pcm2.shape[1])
    assert m >= pcm.shape[1] - 2048 and np.array_equal(pcm[:, :m], pcm2[:, :m])
    # retune channel 1 after the first block: signal at -0.11 for the first block, at +0.07 afterwards
    a = nfm_signal_u8(1300, sizes[0], offset=-0.11); b = nfm_signal_u8(1301, sizes[1], offset=0.07)
    u8r = u8.copy(); u8r[1] = np.concatenate([a, b])
    pcm3, _ = gpu.nfm_chain(u8r, rates, block=sizes, retunes={1: [(1, -0.07)]})
    xf = port.convert_u8_f(u8r[1]).view(c64)
    s1, ph = port.shift_addition_cc(xf[:sizes[0]], 0.11)
    s2, _ = port.shift_addition_cc(xf[sizes[0]:], -0.07, phase=ph)
    dec = port.fir_decimate_cc(np.concatenate([s1, s2]), D, port.firdes_lowpass_f(port.firdes_filter_len(0.005), 0.5 / D))
    dem, _ = port.fmdemod_quadri_cf(dec)
    agc = port.fastagc_ff(port.deemphasis_nfm_ff_cli(port.limit_ff(dem, 1.0), nfm_taps), 1024, 1.0)
    want = port.convert_f_s16(agc)
    m = min(want.size, pcm3.shape[1])
    assert m >= want.size - 2048 and vc.s16_diff(pcm3[1, :m], want[:m]).max() <= 1
    assert np.array_equal(pcm3[0, :m], pcm2[0, :m])


WRATES = [-0.085, 0.11, 0.25, -0.3, 0.05, 0.2, -0.1234, 0.4, -0.45, 0.0]


@pytest.mark.parametrize("n,S,want_float", [(1024 * 517, 5, True), (1024 * 517, 4, False), (1024 * 60, 10, False), (1024 * 26 + 700, 3, True)])
def test_wfm_rates_single_call(gpu, port, n, S, want_float):
    """csdr_amd_wfm_create_rates: a shift rate per stream through the WFM chain kernel (one workgroup = one stream x 16 time segments).  517 chunks = 20.2 periods of
    25600 samples -> two column groups with a partial last column and columns behind the block's end; 60 chunks = 3 columns; a ragged single-column block.  With
    and without the float tap (the s16-only runs take the loaders' line-collecting store path).  Every stream +-1 LSB against the oracle's seven stages at its rate."""
    taps = port.firdes_lowpass_f(79, 0.05)
    rates = np.array(WRATES[:S], f32)
    u8 = np.stack([wfm_signal_u8(2000 + s, n, offset=-float(rates[s])) for s in range(S)])
    s16, af = gpu.wfm_chain(u8, rates, 10, taps, want_float=want_float)
    assert gpu.last_wfm_kernel == "k_wfm_mfma_seq"
    for s in range(S):
        ps, pf = port.wfm_chain(u8[s], float(rates[s]), 10, taps)
        m = min(ps.size, s16.shape[1])
        assert 0 <= ps.size - s16.shape[1] <= 2 or 0 <= s16.shape[1] - ps.size <= 2
        assert vc.s16_diff(s16[s, :m], ps[:m]).max() <= 1, "stream %d rate %g: %d" % (s, rates[s], vc.s16_diff(s16[s, :m], ps[:m]).max())
        if want_float:
            assert relrms(af[s, :m], pf[:m]) <= TOL


@pytest.mark.parametrize("out_per_call", [False, True])
def test_wfm_rates_blocks_and_retune(gpu, port, out_per_call):
    """(out_per_call: every call stores at the start of aligned rows -- the line-collecting store path, with calls that start in the middle of a tile.)
    Consecutive calls (tables switched on the side stream, history heads, de-emphasis state across calls and columns), bit-identical to the single call; then
    csdr_amd_wfm_set_rate between calls: new rate from the next block's first sample, phase carried (csdr.c:881-923) -- the oracle retunes its shift stage there."""
    taps = port.firdes_lowpass_f(79, 0.05)
    S = 4
    sizes = [1024 * 100, 1024 * 64, 1024 * 150, 1024 * 30]
    n = sum(sizes)
    rates = np.array(WRATES[:S], f32)
    u8 = np.stack([wfm_signal_u8(2100 + s, n, offset=-float(rates[s])) for s in range(S)])
    a, _ = gpu.wfm_chain(u8, rates, 10, taps, want_float=False)
    b, _ = gpu.wfm_chain(u8, rates, 10, taps, block=sizes, want_float=False, out_per_call=out_per_call)
    m = min(a.shape[1], b.shape[1])
    assert m >= a.shape[1] - 2 and vc.s16_diff(a[:, :m], b[:, :m]).max() <= 1      # (column boundaries differ between the two runs: the warm-up's 5e-8)
    # retune stream 1 in front of the third call: its signal moves with it
    pos = int(np.cumsum([0] + sizes)[2])
    u8r = u8.copy()
    u8r[1] = np.concatenate([wfm_signal_u8(2200, pos, offset=-0.11), wfm_signal_u8(2201, n - pos, offset=0.3)])
    c, _ = gpu.wfm_chain(u8r, rates, 10, taps, block=sizes, retunes={2: [(1, -0.3)]}, want_float=False, out_per_call=out_per_call)
    xf = port.convert_u8_f(u8r[1]).view(c64)
    s1, ph = port.shift_addition_cc(xf[:pos], 0.11)
    s2, _ = port.shift_addition_cc(xf[pos:], -0.3, phase=ph)
    dec = port.fir_decimate_cc(np.concatenate([s1, s2]), 10, taps)
    dem, _ = port.fmdemod_quadri_cf(dec)
    aud = dem[10::5]                                       # fractional_decimator_ff 5 == x[5 k + 10] (exact at an integer rate)
    want = port.convert_f_s16(port.deemphasis_wfm_ff(aud, 50e-6, 48000)[0])
    m = min(want.size, c.shape[1])
    assert m >= want.size - 4 and vc.s16_diff(c[1, :m], want[:m]).max() <= 1, vc.s16_diff(c[1, :m], want[:m]).max()
    assert vc.s16_diff(c[0, :m], b[0, :m]).max() == 0
