"""Golden vectors produced by the UNMODIFIED reference (tests/golden/ref_vectors.npz, generator: tests/golden/make_golden.py, which needs
/root/reference) replayed through (a) the CPU oracle -- runs anywhere, pins the oracle even where oracle/_ref cannot be built -- and
(b) the HIP path through the C ABI on the GPU box.  Gates: bit exact for integer/byte outputs, <= 1e-5 relative RMS for float paths
(the reference build is -ffast-math, so float outputs of the restatement are within ~1e-6, not bit equal)."""
import os

import numpy as np
import pytest
from oracle import relrms

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
c64, f32 = np.complex64, np.float32
TOL = 1e-5


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_vectors.npz"))


def same_bits(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def run_all(impl, G, nfm_taps, stream_fir, is_gpu):
    """impl: oracle.Port or csdr_amd.Context -- the numpy conveniences carry the same names and argument meaning."""
    x, rx = G["cx"], G["rx"]
    # converters: bit exact
    assert same_bits(impl.convert_u8_f(G["u8"]), G["convert_u8_f"])
    assert same_bits(impl.convert_s8_f(G["s8"]), G["convert_s8_f"])
    assert same_bits(impl.convert_s16_f(G["s16"]), G["convert_s16_f"])
    for name in ("convert_f_u8", "convert_f_s8", "convert_f_s16"):
        assert same_bits(getattr(impl, name)(G["flt"]), G[name]), name
    assert same_bits(impl.convert_f_s24(G["flt"], 0), G["convert_f_s24_le"]) and same_bits(impl.convert_f_s24(G["flt"], 1), G["convert_f_s24_be"])
    assert same_bits(impl.convert_s24_f(G["s24"], 0), G["convert_s24_f_le"]) and same_bits(impl.convert_s24_f(G["s24"], 1), G["convert_s24_f_be"])
    # design
    assert relrms(impl.firdes_lowpass_f(79, 0.05), G["firdes_lowpass_79"]) <= 1e-6
    assert relrms(impl.firdes_lowpass_f(801, 0.01, "BLACKMAN"), G["firdes_lowpass_801_blackman"]) <= 5e-6
    assert relrms(impl.firdes_bandpass_c(255, -0.1, 0.2), G["firdes_bandpass_255"]) <= 5e-6
    # shifters
    for name in ("shift_addition_cc", "shift_math_cc", "shift_addfast_cc", "shift_unroll_cc"):
        assert relrms(getattr(impl, name)(x, -0.085)[0], G[name]) <= TOL, name
    assert relrms(impl.shift_table_cc(x, -0.085, 4096)[0], G["shift_table_cc"]) <= TOL
    assert relrms(impl.shift_addition_fc(rx, 0.21)[0], G["shift_addition_fc"]) <= TOL
    y, st = impl.decimating_shift_addition_cc(x, 0.07, 6)
    assert y.size == G["decimating_shift_addition_cc"].size and relrms(y, G["decimating_shift_addition_cc"]) <= TOL
    assert int(st[0]) == int(G["dsa_status"][0]) and int(st[2]) == int(G["dsa_status"][2])
    # FIR / demod / audio
    taps = impl.firdes_lowpass_f(79, 0.05)
    y = stream_fir(x, 10, taps)
    assert y.size == G["fir_decimate_cc"].size and relrms(y, G["fir_decimate_cc"]) <= TOL
    assert relrms(impl.fmdemod_quadri_cf(x, (0.3, -0.2) if not is_gpu else np.array([0.3 - 0.2j], c64))[0], G["fmdemod_quadri_cf"]) <= TOL
    assert relrms(impl.deemphasis_wfm_ff(rx, 50e-6, 48000, 0.1 if not is_gpu else np.array([0.1], f32))[0], G["deemphasis_wfm_ff"]) <= TOL
    y = nfm_taps(rx)
    assert y.size == G["deemphasis_nfm_ff_48000"].size and relrms(y, G["deemphasis_nfm_ff_48000"]) <= TOL
    assert same_bits(impl.limit_ff((rx * 1.7).astype(f32), 1.0), G["limit_ff"])
    assert relrms(impl.gain_ff(rx, 2.5), G["gain_ff"]) <= 1e-7
    assert relrms(impl.fastagc_ff(G["agc_in"], 1024, 0.8), G["fastagc_ff"]) <= TOL
    for key, args in (("fractional_decimator_ff_5", (5.0,)), ("fractional_decimator_ff_2p5_4", (2.5, 4))):
        y = impl.fractional_decimator_ff(rx, *args)
        m = min(y.size, G[key].size)
        assert m >= G[key].size - 1 and relrms(y[:m], G[key][:m]) <= TOL, key
    # FFT paths
    y = impl.bandpass_fir_fft_cc(x, impl.firdes_bandpass_c(255, -0.1, 0.2), 1024)
    assert y.size == G["bandpass_fir_fft_cc_1024"].size and relrms(y, G["bandpass_fir_fft_cc_1024"]) <= TOL
    # f2
    assert relrms(impl.amdemod_cf(x), G["amdemod_cf"]) <= TOL
    assert relrms(impl.amdemod_estimator_cf(x), G["amdemod_estimator_cf"]) <= 1e-7
    assert relrms(impl.fmdemod_atan_cf(x, 0.3 if not is_gpu else np.array([0.3], f32))[0], G["fmdemod_atan_cf"]) <= TOL
    assert relrms(impl.logpower_cf(x + c64(0.01), 3.0), G["logpower_cf"]) <= TOL
    st = (0.1, 0.2) if not is_gpu else np.array([0.1, 0.2], f32)
    assert relrms(impl.dcblock_ff(G["dc_in"], 0, st)[0], G["dcblock_ff"]) <= TOL
    ld = 0.1 if not is_gpu else np.array([0.1], f32)
    assert relrms(impl.fastdcblock_ff(G["dc_in"], 1024, ld)[0], G["fastdcblock_ff"]) <= TOL
    assert relrms(impl.agc_ff(G["agc_in"], 1024)[0], G["agc_ff"]) <= TOL
    assert relrms(impl.precalculate_window(512, "HAMMING"), G["window_hamming_512"]) <= 1e-6
    # f3: bit exact
    st = (5, -100) if not is_gpu else np.array([5, -100], np.int32)
    enc, es = impl.encode_ima_adpcm_i16_u8(G["pcm"], st)
    assert same_bits(enc, G["adpcm_enc"]) and [int(v) for v in es] == [int(v) for v in G["adpcm_enc_state"]]
    st = (3, 77) if not is_gpu else np.array([3, 77], np.int32)
    dec, ds = impl.decode_ima_adpcm_u8_i16(G["adpcm_enc"], st)
    assert same_bits(dec, G["adpcm_dec"]) and [int(v) for v in ds] == [int(v) for v in G["adpcm_dec_state"]]
    assert same_bits(impl.compress_fft_adpcm_f_u8(G["fft_rows_db"], 256), G["compress_fft_adpcm_f_u8_256"])


def check_wfm(s16, af, G):
    n = min(af.size, G["wfm_audio_f"].size)
    assert n >= G["wfm_audio_f"].size - 2 and n > 1200
    assert relrms(af[:n], G["wfm_audio_f"][:n]) <= TOL
    d = np.abs(s16[:n].astype(np.int32) - G["wfm_audio_s16"][:n].astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 0.05


def test_oracle_against_golden(port, G):
    taps48 = np.load(os.path.join(ROOT, "tests", "golden", "nfm_deemph_taps.npz"))["sr48000"]
    run_all(port, G, lambda a: port.deemphasis_nfm_ff(a, taps48), port.fir_decimate_cc, False)
    d, _ = port.fastddc_init(0.05, 16, 0.11)
    geo = [d.pre_decimation, d.post_decimation, d.taps_length, d.overlap_length, d.fft_size, d.fft_inv_size, d.input_size, d.post_input_size, d.startbin, d.offsetbin, d.scrap]
    assert geo == [int(v) for v in G["fastddc_geometry"]]
    spec = port.fastddc_fwd_cc(G["ddc_x"], d)
    assert relrms(spec, G["fastddc_fwd_cc"]) <= TOL
    y = port.fastddc_inv_cc(spec, d, port.fastddc_taps_fft(d, 0.11, 16))
    assert y.size == G["fastddc_inv_cc"].size and relrms(y, G["fastddc_inv_cc"]) <= TOL
    s16, af = port.wfm_chain(G["wfm_iq_u8"], -0.085, 10, port.firdes_lowpass_f(79, 0.05))
    check_wfm(s16, af, G)


@pytest.mark.gpu
def test_hip_path_against_golden(G):
    import torch  # noqa: F401
    import csdr_amd
    gpu = csdr_amd.Context(0)
    try:
        taps48 = gpu.nfm_taps(48000)
        run_all(gpu, G, lambda a: gpu.fir_ff(a, taps48), gpu.fir_decimate_cc, True)
        d, _ = gpu.fastddc_init(0.05, 16, 0.11)
        geo = [d.pre_decimation, d.post_decimation, d.taps_length, d.overlap_length, d.fft_size, d.fft_inv_size, d.input_size, d.post_input_size, d.startbin, d.offsetbin, d.scrap]
        assert geo == [int(v) for v in G["fastddc_geometry"]]
        spec = gpu.fastddc_fwd_cc(G["ddc_x"], d)
        assert relrms(spec, G["fastddc_fwd_cc"]) <= TOL
        y = gpu.fastddc_inv_cc(G["fastddc_fwd_cc"], 0.05, 16, [0.11])[0]
        assert y.size == G["fastddc_inv_cc"].size and relrms(y, G["fastddc_inv_cc"]) <= TOL
        s16, af = gpu.wfm_chain(G["wfm_iq_u8"][None, :], -0.085, 10, gpu.firdes_lowpass_f(79, 0.05))
        check_wfm(s16[0], af[0], G)
    finally:
        gpu.close()


# ---------------------------------------------------------------- the NFM chain (BASELINE config 5) as run by the reference CLI pipeline
def _nfm_golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "nfm_cli_vectors.npz"))


def _check_nfm(pcm, N):
    want = N["nfm_cli_s16"]
    assert pcm.size == want.size
    assert np.all(want[:2048] == 0) and np.any(want[2048:] != 0)          # fastagc's two-block latency, then audio
    d = np.abs(pcm.astype(np.int32) - want.astype(np.int32))
    assert d.max() <= 1 and np.mean(d != 0) < 0.02


def test_oracle_nfm_chain_against_reference_pipeline(port):
    N = _nfm_golden()
    taps48 = np.load(os.path.join(ROOT, "tests", "golden", "nfm_deemph_taps.npz"))["sr48000"]
    pcm, _ = port.nfm_chain(N["nfm_iq_u8"], float(N["shift_rate"]), taps48)
    _check_nfm(pcm, N)


@pytest.mark.gpu
def test_hip_nfm_chain_against_reference_pipeline():
    import torch  # noqa: F401
    import csdr_amd
    N = _nfm_golden()
    gpu = csdr_amd.Context(0)
    try:
        for block in (None, 1024 * 70):
            pcm, _ = gpu.nfm_chain(N["nfm_iq_u8"][None, :], float(N["shift_rate"]), block=block)
            _check_nfm(pcm[0], N)
    finally:
        gpu.close()



# ---------------------------------------------------------------- the reference at the BASELINE geometries (tests/golden/baseline_vectors.npz, `make_golden.py baseline`)
def _baseline():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)          # the input generators (seeded; the inputs themselves are not stored)
    return mg, np.load(os.path.join(ROOT, "tests", "golden", "baseline_vectors.npz"))


def _check_baseline(impl, mg, B, c4_channels_fn, nfm_fn):
    """GPU <-> REFERENCE directly (not via the restatement): the 1e-5 budget is not shared with the oracle's own distance from the reference at these sizes."""
    # C3: apply_fir_fft_cc at fft 65536 (libcsdr.c:814-849), 1023 and 4095 taps, three blocks -- the reference's own taps AND the implementation's own design
    for nt in mg.C3_TAPS:
        x = mg.c3_input(nt); keep = mg.c3_keep(nt)
        for taps in (B["c3_taps_%d" % nt], impl.firdes_bandpass_c(nt, -0.1, 0.2)):
            y = impl.bandpass_fir_fft_cc(x, taps, 65536)
            assert y.size == mg.C3_BLOCKS * (65537 - nt)
            assert relrms(y[keep], B["c3_out_%d" % nt]) <= TOL, nt
    # C4: decimation 256 / transition_bw 0.001 (fft 65536, taps 8193, fft_inv 512), five blocks, three channels (fastddc.c:91-166)
    x = mg.c4_input()
    d0, err = impl.fastddc_init(0.001, 256, 0.0)
    assert err == 0
    spec = impl.fastddc_fwd_cc(x, d0)
    assert relrms(spec[:, mg.c4_spec_keep()], B["c4_spec_subset"]) <= TOL
    rates = {c: float(np.float32(-0.5 + (c + 0.5) / 256)) for c in mg.C4_CHANNELS}
    for c in mg.C4_CHANNELS:
        dc, _ = impl.fastddc_init(0.001, 256, rates[c])
        geo = [dc.pre_decimation, dc.post_decimation, dc.taps_length, dc.overlap_length, dc.fft_size, dc.fft_inv_size, dc.input_size, dc.post_input_size, dc.startbin, dc.offsetbin, dc.scrap]
        assert geo == [int(v) for v in B["c4_geometry_ch%d" % c]]
    for name, outs in c4_channels_fn(x, spec, rates):
        for c in mg.C4_CHANNELS:
            want = B["c4_out_ch%d" % c]
            assert outs[c].size == want.size and relrms(outs[c], want) <= TOL, (name, c)
    # C5: README.md:87 as eight processes of the reference binary at rates 0.25 / 0.05 / -0.4321
    for k, rate in enumerate(mg.C5_RATES):
        pcm = nfm_fn(mg.c5_input(k), rate)
        want = B["c5_s16_rate%d" % k]
        assert pcm.size == want.size and np.any(want[2048:] != 0)
        d = np.abs(pcm.astype(np.int32) - want.astype(np.int32))
        assert d.max() <= 1 and np.mean(d != 0) < 0.02, (rate, int(d.max()))


def test_oracle_against_baseline_golden(port):
    mg, B = _baseline()
    taps48 = np.load(os.path.join(ROOT, "tests", "golden", "nfm_deemph_taps.npz"))["sr48000"]

    def channels(x, spec, rates):
        outs = {}
        for c, r in rates.items():
            dc, _ = port.fastddc_init(0.001, 256, r)
            outs[c] = port.fastddc_inv_cc(spec, dc, port.fastddc_taps_fft(dc, r, 256))
        yield "oracle", outs
    _check_baseline(port, mg, B, channels, lambda iq, r: port.nfm_chain(iq, r, taps48)[0])


@pytest.mark.gpu
def test_hip_path_against_baseline_golden():
    import torch  # noqa: F401
    import csdr_amd
    mg, B = _baseline()
    gpu = csdr_amd.Context(0)
    try:
        def channels(x, spec, rates):
            cs = sorted(rates)
            outs = gpu.fastddc_inv_cc(spec, 0.001, 256, [rates[c] for c in cs])              # the per-channel inverse of the device batch API, three channels
            yield "fastddc_inv", {c: outs[i] for i, c in enumerate(cs)}
            allr = (-0.5 + (np.arange(256) + 0.5) / 256).astype(f32)                           # the bank object bench_fastddc.py times: all 256 channels, five blocks in one call
            outs = gpu.fastddc_bank(x, 0.001, 256, allr, blocks_per_call=mg.C4_BLOCKS)
            yield "bank(256)", {c: outs[c] for c in cs}
            outs = gpu.fastddc_bank(x, 0.001, 256, allr, blocks_per_call=2)                   # ... and in calls of two, two, one blocks (state carried)
            yield "bank(256) in three calls", {c: outs[c] for c in cs}

        def nfm(iq, r):
            a = gpu.nfm_chain(iq[None, :], r)[0][0]
            b = gpu.nfm_chain(np.stack([iq, iq]), np.array([r, r], f32), block=1024 * 70)[0][1]        # the rate-per-channel object, in blocks
            assert np.array_equal(a, b) or np.abs(a.astype(np.int32) - b.astype(np.int32)).max() <= 1
            return a
        _check_baseline(gpu, mg, B, channels, nfm)
    finally:
        gpu.close()
