"""Parity at the EXACT BASELINE.json shapes that the benches time (these were timed but never checked):
  config 4  fastddc D = 256, transition_bw 0.001 -> fft 65536 / inverse 512, 256 channels at -0.5 + (c + 0.5)/256
  config 2  WFM chain at 1024 streams x 2 400 256 samples (64 stream blocks x 4 time segments, > 2^31 byte offsets)
  config 5  NFM chain at 512 channels x 2 400 256 samples
plus the FFT plan layer's real transforms (fft_fftw.c:16-35), which nothing referenced."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import verify_configs as vc  # noqa: E402
from tests_helpers import wfm_signal_u8, nfm_signal_u8  # noqa: E402

pytestmark = pytest.mark.gpu
c64 = np.complex64
f32 = np.float32
TOL = 1e-5


@pytest.fixture(scope="module")
def gpu():
    import torch  # noqa: F401
    import csdr_amd
    ctx = csdr_amd.Context(0)
    assert ctx.arch().startswith("gfx950")
    yield ctx
    ctx.close()


def test_c4_exact_shape(gpu, port):
    """fastddc.c:106-166 / csdr.c:2255-2378 at config 4's geometry: all 256 channels folded, 18 compared (band edges c = 0, 255, the
    middle pair 127 / 128, tile edges of the fold kernels), 5 blocks in calls of 3 + 2 (state carried between calls)."""
    tbw, D, nch, nb = 0.001, 256, 256, 5
    ddc, err = gpu.fastddc_init(tbw, D, 0.0)
    assert err == 0 and (ddc.fft_size, ddc.fft_inv_size, ddc.taps_length, ddc.input_size, ddc.pre_decimation, ddc.post_decimation) == (65536, 512, 8193, 57344, 128, 2)
    rng = np.random.default_rng(4)
    x = (rng.uniform(-1, 1, ddc.input_size * nb) + 1j * rng.uniform(-1, 1, ddc.input_size * nb)).astype(c64)
    rates = vc.c4_rates(nch)
    check = [0, 1, 7, 8, 31, 32, 63, 64, 100, 126, 127, 128, 129, 191, 192, 200, 254, 255]
    pspec, want = vc.fastddc_oracle_channels(x, tbw, D, rates, check)
    spec = gpu.fastddc_fwd_cc(x, ddc, blocks_per_call=3)
    assert vc.relrms(spec, pspec) < TOL                                     # the fft-65536 forward stream
    outs = gpu.fastddc_inv_cc(spec, tbw, D, rates, blocks_per_call=3)
    assert len(outs) == nch
    for c in check:
        assert outs[c].size == want[c].size and want[c].size >= nb * (ddc.post_input_size // ddc.post_decimation) - 1, "channel %d" % c
        assert vc.relrms(outs[c], want[c]) < TOL, "channel %d" % c
    sizes = {o.size for o in outs}
    assert len(sizes) <= 2 and max(sizes) - min(sizes) <= 1                  # +-1 sample between channels (decimation_remain chain)


def test_c4_matrix_core_fold_ragged(gpu, port):
    """the matrix-core path of the fastddc inverse (fastddc_mfma.hip) with ragged tiles on both axes: 37 channels (two channel waves, the second
    partly empty), 40 blocks in ONE call (two 32-block accumulator tiles, the second partly empty) followed by a 3-block call (single tile,
    state carried), every channel compared; and the same call sequence through the general kernels (CSDR_AMD_DDC_MFMA_OFF) gives the same stream."""
    tbw, D, nch, nb = 0.001, 256, 37, 43
    ddc, _ = gpu.fastddc_init(tbw, D, 0.0)
    rng = np.random.default_rng(44)
    x = (rng.uniform(-1, 1, ddc.input_size * nb) + 1j * rng.uniform(-1, 1, ddc.input_size * nb)).astype(c64)
    rates = np.concatenate([vc.c4_rates(256)[::8][:32], np.array([0.0, 0.1234, -0.3711, 0.25, -0.4999], f32)])
    assert rates.size == nch
    pspec, want = vc.fastddc_oracle_channels(x, tbw, D, rates, range(nch))
    outs = gpu.fastddc_inv_cc(pspec, tbw, D, rates, blocks_per_call=40)
    for c in range(nch):
        assert outs[c].size == want[c].size, "channel %d" % c
        assert vc.relrms(outs[c], want[c]) < TOL, "channel %d" % c
    os.environ["CSDR_AMD_DDC_MFMA_OFF"] = "1"
    try:
        general = gpu.fastddc_inv_cc(pspec, tbw, D, rates[:5], blocks_per_call=40)
    finally:
        del os.environ["CSDR_AMD_DDC_MFMA_OFF"]
    for c in range(5):
        assert general[c].size == outs[c].size and vc.relrms(general[c], outs[c]) < TOL


def test_c4_bank_fused_forward(gpu, port):
    """csdr_amd_fastddc_bank_* at config 4's geometry: new samples in -> own 65536 = 512 x 128 forward transform straight into the fold's layout
    (no natural-order spectrum) -> matrix-core fold -> 512-point inverse transforms; 37 channels, 43 blocks in calls of 40 + 3 (overlap tail and
    channel state carried), channel 35 retuned before the second call (csdr.c:2329-2376: status restarts), against the oracle's two-process model."""
    tbw, D, nch, nb = 0.001, 256, 37, 43
    ddc, _ = gpu.fastddc_init(tbw, D, 0.0)
    rng = np.random.default_rng(45)
    x = (rng.uniform(-1, 1, ddc.input_size * nb + 100) + 1j * rng.uniform(-1, 1, ddc.input_size * nb + 100)).astype(c64)
    rates = np.concatenate([vc.c4_rates(256)[3::8][:32], np.array([0.0, -0.2222, 0.3711, 0.125, 0.4999], f32)])
    outs = gpu.fastddc_bank(x, tbw, D, rates, blocks_per_call=40, retune=(1, 35, 0.0517))
    assert gpu.last_ddc_kernel in ("k_ddc_gemm", "k_ddc_gemm3", "k_ddc_gemm3n")          # (37 channels: the fold of few channel rows since round 5)
    pspec, want = vc.fastddc_oracle_channels(x, tbw, D, rates, [c for c in range(nch) if c != 35])
    for c, w in want.items():
        assert outs[c].size == w.size, "channel %d" % c
        assert vc.relrms(outs[c], w) < TOL, "channel %d" % c
    pd_a, _ = port.fastddc_init(tbw, D, float(rates[35])); pd_b, _ = port.fastddc_init(tbw, D, 0.0517)
    part_a = port.fastddc_inv_cc(pspec[:40], pd_a, port.fastddc_taps_fft(pd_a, float(rates[35]), D))
    part_b = port.fastddc_inv_cc(pspec[40:], pd_b, port.fastddc_taps_fft(pd_b, 0.0517, D))
    w35 = np.concatenate([part_a, part_b])
    assert outs[35].size == w35.size and vc.relrms(outs[35], w35) < TOL


@pytest.mark.parametrize("switch", ["CSDR_AMD_DDC_SPEC_OFF=1", "CSDR_AMD_DDC_RIDERS_OFF=1", "CSDR_AMD_DDC_PASS2=1"])
def test_c4_bank_alternative_paths(gpu, monkeypatch, switch):
    """The bank's alternative schedules stay correct: each test hook turns one of the default choices off (chain tables one call ahead, riders, pass 2 inside the
    fold) -- the paths a sharded bank, a change of the call size or more than 4096 channels select; same stream as the default within rounding."""
    tbw, D, nch, nb = 0.001, 256, 37, 11
    ddc, _ = gpu.fastddc_init(tbw, D, 0.0)
    rng = np.random.default_rng(48)
    x = (rng.uniform(-1, 1, ddc.input_size * nb) + 1j * rng.uniform(-1, 1, ddc.input_size * nb)).astype(c64)
    rates = np.ascontiguousarray(vc.c4_rates(256)[2::7][:nch])
    want = gpu.fastddc_bank(x, tbw, D, rates, blocks_per_call=4, retune=(2, 5, 0.0321))
    k, v = switch.split("=")
    monkeypatch.setenv(k, v)
    got = gpu.fastddc_bank(x, tbw, D, rates, blocks_per_call=4, retune=(2, 5, 0.0321))
    for c in range(nch):
        assert got[c].size == want[c].size and vc.relrms(got[c], want[c]) < 2e-6, "channel %d" % c


def test_c4_bank_random_schedule(gpu, monkeypatch):
    """Chain tables computed one call ahead (riders of the previous call's inverse transforms, committed when the size matches and nothing retuned) against the same
    bank with that machinery off, on a random schedule: 26 calls of 1 .. 7 blocks with runs of equal sizes (hits) and changes (misses), retunes of random channels
    before a third of the calls -- every channel's stream equal to rounding, equal sample counts; three channels that are never retuned also against the oracle."""
    tbw, D, nch = 0.001, 256, 23
    ddc, _ = gpu.fastddc_init(tbw, D, 0.0)
    rng = np.random.default_rng(49)
    sizes = []
    while len(sizes) < 26:
        sizes += [int(rng.integers(1, 8))] * int(rng.integers(1, 5))
    sizes = sizes[:26]
    nb = sum(sizes)
    x = (rng.uniform(-1, 1, ddc.input_size * nb) + 1j * rng.uniform(-1, 1, ddc.input_size * nb)).astype(c64)
    rates = np.ascontiguousarray(vc.c4_rates(256)[1::11][:nch])
    retunes = {int(c): [(int(rng.integers(0, nch)), float(rng.uniform(-0.45, 0.45)))] for c in rng.choice(np.arange(1, 26), 8, replace=False)}
    got = gpu.fastddc_bank(x, tbw, D, rates, schedule=sizes, retunes=retunes)
    monkeypatch.setenv("CSDR_AMD_DDC_SPEC_OFF", "1"); monkeypatch.setenv("CSDR_AMD_DDC_RIDERS_OFF", "1")
    want = gpu.fastddc_bank(x, tbw, D, rates, schedule=sizes, retunes=retunes)
    for c in range(nch):
        assert got[c].size == want[c].size and vc.relrms(got[c], want[c]) < 2e-6, "channel %d" % c
    # ... and the oracle on three channels that are never retuned (the whole random schedule, 1e-5)
    touched = {ch for lst in retunes.values() for ch, _ in lst}
    check = [c for c in range(nch) if c not in touched][:3]
    _, ref = vc.fastddc_oracle_channels(x, tbw, D, rates, check)
    for c in check:
        assert got[c].size == ref[c].size and vc.relrms(got[c], ref[c]) < 1e-5, "channel %d vs oracle" % c


def test_bank_pipelined_single_rank_communicator(gpu):
    """submit(N + 1) before collect(N) (two batches staged: the second one's chains and forward transform run on the side stream under the first one's
    fold) gives the stream process() gives; the bank is created through the sharded entry point on a ONE-rank RCCL communicator of the library's own
    (csdr_amd_comm_*: the dlopen of librccl and ncclCommInitRank are exercised; with one rank the slice is everything and nothing moves)."""
    L = gpu.L
    tbw, D, nch, per, calls = 0.001, 256, 20, 9, 4
    ddc, _ = gpu.fastddc_init(tbw, D, 0.0)
    rng = np.random.default_rng(47)
    x = (rng.uniform(-1, 1, ddc.input_size * per * calls) + 1j * rng.uniform(-1, 1, ddc.input_size * per * calls)).astype(c64)
    rates = np.ascontiguousarray(vc.c4_rates(256)[5::13][:nch])
    want = gpu.fastddc_bank(x, tbw, D, rates, blocks_per_call=per)
    idb = (C.c_char * 128)()
    assert L.csdr_amd_comm_unique_id(idb) == 0, gpu.err()
    comm = L.csdr_amd_comm_create(gpu.h, idb, 0, 1)
    assert comm, gpu.err()
    assert (L.csdr_amd_comm_rank(comm), L.csdr_amd_comm_world(comm)) == (0, 1)
    bank = L.csdr_amd_fastddc_bank_create_sharded(gpu.h, tbw, D, rates.ctypes.data_as(C.c_void_p), nch, 2, per, comm)
    assert bank, gpu.err()
    f0 = C.c_int(); c0 = C.c_int(); L.csdr_amd_fastddc_bank_channel_slice(bank, C.byref(f0), C.byref(c0))
    assert (f0.value, c0.value) == (0, nch)
    di = gpu.upload(x)
    pitch = L.csdr_amd_fastddc_bank_max_output(bank, per) + 8
    outs = [[] for _ in range(nch)]
    step = 8 * per * ddc.input_size
    assert L.csdr_amd_fastddc_bank_submit(bank, di.at(0), per) == 0, gpu.err()
    for k in range(calls):
        if k + 1 < calls:
            assert L.csdr_amd_fastddc_bank_submit(bank, di.at((k + 1) * step), per) == 0, gpu.err()
        if k == 0:
            assert L.csdr_amd_fastddc_bank_submit(bank, di.at(0), per) < 0            # a third staged batch is refused
        do = gpu.alloc(8 * nch * pitch)
        counts = np.zeros(nch, np.int32)
        assert L.csdr_amd_fastddc_bank_collect(bank, do.ptr, pitch, counts.ctypes.data_as(C.c_void_p)) == 0, gpu.err()
        y = gpu.download(do, c64, nch * pitch).reshape(nch, pitch)
        for c in range(nch):
            outs[c].append(y[c, :counts[c]].copy())
    assert L.csdr_amd_fastddc_bank_collect(bank, do.ptr, pitch, None) < 0                  # nothing staged any more
    L.csdr_amd_fastddc_bank_destroy(bank); L.csdr_amd_comm_destroy(comm)
    for c in range(nch):
        got = np.concatenate(outs[c])
        # (process() runs the second forward pass inside the fold, the staged form as a kernel of its own: the same butterflies, but the compiler contracts
        # the twiddle products differently in the two contexts -- equal to rounding, not to the bit)
        assert got.size == want[c].size and vc.relrms(got, want[c]) < 2e-6, "channel %d" % c


def test_bank_general_geometry(gpu, port):
    """the same object where the matrix-core path does not apply (fft_inv_size 128): forward and inverse halves chained through a spectrum buffer"""
    tbw, D = 0.05, 16
    rates = [-0.1, 0.2, 0.33, 0.05, -0.41]
    ddc, _ = gpu.fastddc_init(tbw, D, 0.0)
    rng = np.random.default_rng(46)
    x = (rng.uniform(-1, 1, ddc.input_size * 23) + 1j * rng.uniform(-1, 1, ddc.input_size * 23)).astype(c64)
    outs = gpu.fastddc_bank(x, tbw, D, rates, blocks_per_call=7)
    assert gpu.last_ddc_kernel == "k_ddc_fold_ct"
    _, want = vc.fastddc_oracle_channels(x, tbw, D, np.array(rates, f32), range(len(rates)))
    for c in range(len(rates)):
        assert outs[c].size == want[c].size and vc.relrms(outs[c], want[c]) < TOL


def test_c1_fir_decimate_at_256_streams(gpu, port):
    """bench_fir.py's timed configuration (BASELINE configs[0] batched): 256 streams x 2 400 256 complexf samples, decimation 10, 79 taps; rows spread over the
    batch against the oracle's fir_decimate_cc (libcsdr.c:528-549), and the long-filter shape (50 / 801 taps, the matrix-core kernel) on 64 streams."""
    import torch
    L = gpu.L
    for S, D, tbw, kern in ((256, 10, 0.05, "poly"), (64, 50, 0.005, "mfma")):
        T = 2344 * 1024
        nt = gpu.firdes_filter_len(tbw)
        taps_h = gpu.firdes_lowpass_f(nt, 0.5 / D, "HAMMING")
        taps = gpu.upload(taps_h)
        g = torch.Generator(device="cuda"); g.manual_seed(77 + D)
        x = (torch.rand((S, T, 2), device="cuda", generator=g) * 2 - 1).contiguous()
        opitch = T // D + 8
        y = torch.empty((S, opitch, 2), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        n = L.csdr_amd_fir_decimate_cc(gpu.h, x.data_ptr(), y.data_ptr(), S, T, T, opitch, D, taps.ptr, nt)
        assert n == (T - nt) // D + 1, gpu.err()
        gpu.sync()
        for r in vc.pick_rows(S, want=5):
            want = port.fir_decimate_cc(x[r].cpu().numpy().view(np.complex64).ravel(), D, taps_h)
            got = y[r, :n].cpu().numpy().view(np.complex64).ravel()
            assert want.size == got.size and vc.relrms(got, want) < TOL, (kern, r)
        taps.free(); del x, y


def test_c3_fft_filter_at_64_streams(gpu, port):
    """bench_fftfilt.py's timed configuration (BASELINE configs[2]): 64 streams x 16 blocks at the 65536-point framing, 1023 taps -- the one-pass kernel and, with
    CSDR_AMD_FFTFILT_LDS_OFF, the three-pass 65536-point transform -- and the sweep's longest filter (4095 taps: 16384-point windows), rows against the oracle's
    block-by-block overlap-add (libcsdr.c:814-849)."""
    import torch
    L = gpu.L
    S, nb, FFT = 64, 16, 65536
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    x = (torch.rand((S, nb * FFT, 2), device="cuda", generator=g) * 2 - 1).contiguous()
    y = torch.empty((S, nb * FFT, 2), dtype=torch.float32, device="cuda")
    pitch = nb * FFT
    for ntaps, full in ((1023, False), (1023, True), (4095, False)):
        taps = port.firdes_bandpass_c(ntaps, -0.1, 0.2)
        if full:
            os.environ["CSDR_AMD_FFTFILT_LDS_OFF"] = "1"
        try:
            f = L.csdr_amd_fftfilt_create(gpu.h, FFT, taps.ctypes.data_as(C.c_void_p), ntaps, S, nb)
        finally:
            os.environ.pop("CSDR_AMD_FFTFILT_LDS_OFF", None)
        assert f, gpu.err()
        assert (L.csdr_amd_fftfilt_window(f) == 0) == full
        inp = L.csdr_amd_fftfilt_input_size(f)
        assert L.csdr_amd_fftfilt_process(f, x.data_ptr(), y.data_ptr(), nb, pitch, pitch) >= 0, gpu.err()
        gpu.sync()
        for r in vc.pick_rows(S, want=3):
            want = port.bandpass_fir_fft_cc(x[r, :nb * inp].cpu().numpy().view(np.complex64).ravel(), taps, FFT)
            got = y[r, :nb * inp].cpu().numpy().view(np.complex64).ravel()
            assert vc.relrms(got[:want.size], want) < TOL, (ntaps, full, r)
        L.csdr_amd_fftfilt_destroy(f)


def test_c2_wfm_at_1024_streams(gpu):
    """bench.py's timed configuration: 1024 streams x 2 400 256 samples (4.9 GB of u8 IQ), 16 full audio rows of the bench's noise input vs
    the oracle (statistical gate, see verify_configs.verify_wfm) and 6 rows carrying a real FM signal, spread over other stream blocks,
    held to +-1 LSB on every one of their 48 001 samples."""
    import torch
    S, T = 1024, 2344 * 1024
    L = gpu.L
    taps = gpu.firdes_lowpass_f(gpu.firdes_filter_len(0.05), 0.5 / 10, "HAMMING")
    g = torch.Generator(device="cuda"); g.manual_seed(42)
    x = torch.randint(0, 256, (S, 2 * T), dtype=torch.uint8, device="cuda", generator=g)
    n_audio_max = (T // 50 + 64 + 63) // 64 * 64
    out = torch.zeros((S, n_audio_max), dtype=torch.int16, device="cuda")
    strict = [3, 17, 300, 515, 777, 1022]
    for k, r in enumerate(strict):
        x[r] = torch.from_numpy(wfm_signal_u8(2000 + k, T)).cuda()
    torch.cuda.synchronize()
    w = L.csdr_amd_wfm_create(gpu.h, S, -0.085, 10, taps.ctypes.data_as(C.c_void_p), taps.size, 5, 50e-6, 48000, T)
    assert w, gpu.err()
    try:
        res = vc.verify_wfm(gpu, w, x, out, S, T, 2 * T, n_audio_max, taps, strict_rows=strict)
    finally:
        L.csdr_amd_wfm_destroy(w)
    assert res["kernel"] == "k_wfm_mfma_seq"
    assert res["rows_expected_len"] >= 48000 and res["ok"], res


def test_c5_nfm_at_512_channels(gpu):
    """bench_nfm.py's timed configuration: 512 channels x 2 400 256 samples, 16 full s16 rows vs the oracle's stage-by-stage chain."""
    import torch
    S, T, D = 512, 2344 * 1024, 50
    L = gpu.L
    taps = gpu.firdes_lowpass_f(gpu.firdes_filter_len(0.005), 0.5 / D, "HAMMING")
    g = torch.Generator(device="cuda"); g.manual_seed(5000)
    x = torch.randint(0, 256, (S, 2 * T), dtype=torch.uint8, device="cuda", generator=g)
    n_out_max = (T // D + 2048 + 63) // 64 * 64
    out = torch.zeros((S, n_out_max), dtype=torch.int16, device="cuda")
    strict = [5, 18, 250, 400, 510]
    for k, r in enumerate(strict):
        x[r] = torch.from_numpy(nfm_signal_u8(3000 + k, T, offset=0.05)).cuda()
    torch.cuda.synchronize()
    obj = L.csdr_amd_nfm_create(gpu.h, S, -0.05, D, taps.ctypes.data_as(C.c_void_p), taps.size, 48000, 1024, 1.0, 1.0, T)
    assert obj, gpu.err()
    try:
        res = vc.verify_nfm(gpu, obj, x, out, S, T, 2 * T, n_out_max, strict_rows=strict)
    finally:
        L.csdr_amd_nfm_destroy(obj)
    assert res["kernel"] == "k_ddc_mfma"
    assert res["ok"], res


def _replicated(signals, S, T):
    """x[s] = signals[(s + s // 16) % 16]: every stream of the batch carries one of 16 signals, and every signal visits every column of a 16-stream group"""
    import torch
    base = torch.from_numpy(np.stack(signals)).cuda()
    idx = (torch.arange(S, device="cuda") + torch.arange(S, device="cuda") // 16) % 16
    return base[idx].contiguous(), idx.cpu().numpy()


def test_c2_wfm_every_stream_block_and_segment(gpu, port):
    """bench.py's timed shape, 100 % of it: all 1024 streams x 2 400 256 samples carry one of 16 FM signals.  Every replica of a signal must be BIT-identical
    (the 64 stream blocks x 4 time segments of the launch, every column of the matrix product), and the 16 distinct rows are held to +-1 LSB against the
    oracle on every one of their 48 001 samples -- together: every sample of every stream, at the cost of 16 oracle rows."""
    import torch
    S, T = 1024, 2344 * 1024
    L = gpu.L
    taps = gpu.firdes_lowpass_f(gpu.firdes_filter_len(0.05), 0.5 / 10, "HAMMING")
    sigs = [wfm_signal_u8(2100 + k, T) for k in range(16)]
    x, idx = _replicated(sigs, S, T)
    n_audio_max = (T // 50 + 64 + 63) // 64 * 64
    out = torch.zeros((S, n_audio_max), dtype=torch.int16, device="cuda")
    w = L.csdr_amd_wfm_create(gpu.h, S, -0.085, 10, taps.ctypes.data_as(C.c_void_p), taps.size, 5, 50e-6, 48000, T)
    assert w, gpu.err()
    try:
        n = L.csdr_amd_wfm_process(w, x.data_ptr(), 2 * T, T, out.data_ptr(), None, n_audio_max)
        assert n >= 48000, gpu.err()
        gpu.sync()
        assert L.csdr_amd_wfm_kernel_name(w).decode() == "k_wfm_mfma_seq"
    finally:
        L.csdr_amd_wfm_destroy(w)
    first = {int(k): int(np.nonzero(idx == k)[0][0]) for k in range(16)}
    ref_rows = out[torch.tensor([first[int(k)] for k in idx], device="cuda")]
    assert torch.equal(out[:, :n], ref_rows[:, :n]), "replicas of one signal differ between stream blocks / columns"
    for k in range(16):
        ps, _ = port.wfm_chain(sigs[k], -0.085, 10, taps)
        got = out[first[k], :n].cpu().numpy()
        m = min(ps.size, got.size)
        assert 0 <= n - ps.size <= 2 and vc.s16_diff(got[:m], ps[:m]).max() <= 1, "signal %d" % k


def test_wfm_65536_streams_in_10_ms_blocks(gpu, port):
    """The many-streams x short-block operating point (the reference moves 1024 / 16384-sample blocks per process, csdr.c:189-193; a receiver bank that serves 65536
    clients with 10 ms of latency hands over 24576 samples per stream and call): 65536 streams x 24576 samples per call, three consecutive calls with the state
    carried (history heads, chunk seeds, de-emphasis), every stream one of 16 FM signals.  Replicas bit-identical -- all 4096 stream blocks of the launch, every
    column --, the 16 distinct rows +-1 LSB against the oracle's stream on every sample.  (bench.py's operating_points times this shape.)"""
    import torch
    S, T, calls = 65536, 24576, 3
    L = gpu.L
    taps = gpu.firdes_lowpass_f(gpu.firdes_filter_len(0.05), 0.5 / 10, "HAMMING")
    sigs = [wfm_signal_u8(2300 + k, calls * T) for k in range(16)]
    n_audio_max = (T // 50 + 64 + 63) // 64 * 64
    out = torch.zeros((S, n_audio_max), dtype=torch.int16, device="cuda")
    w = L.csdr_amd_wfm_create(gpu.h, S, -0.085, 10, taps.ctypes.data_as(C.c_void_p), taps.size, 5, 50e-6, 48000, T)
    assert w, gpu.err()
    got = [[] for _ in range(16)]
    try:
        for c in range(calls):
            x, idx = _replicated([sg[2 * c * T:2 * (c + 1) * T] for sg in sigs], S, T)
            n = L.csdr_amd_wfm_process(w, x.data_ptr(), 2 * T, T, out.data_ptr(), None, n_audio_max)
            assert n > 0, gpu.err()
            gpu.sync()
            assert L.csdr_amd_wfm_kernel_name(w).decode() == "k_wfm_mfma_seq" and not L.csdr_amd_wfm_fallback(w)
            first = {int(k): int(np.nonzero(idx == k)[0][0]) for k in range(16)}
            ref_rows = out[torch.tensor([first[int(k)] for k in idx], device="cuda")]
            assert torch.equal(out[:, :n], ref_rows[:, :n]), "call %d: replicas of one signal differ between stream blocks / columns" % c
            for k in range(16): got[k].append(out[first[k], :n].cpu().numpy().copy())
            del x
    finally:
        L.csdr_amd_wfm_destroy(w)
    for k in range(16):
        ps, _ = port.wfm_chain(sigs[k], -0.085, 10, taps)
        g = np.concatenate(got[k])
        m = min(ps.size, g.size)
        assert 0 <= g.size - ps.size <= 2 and vc.s16_diff(g[:m], ps[:m]).max() <= 1, "signal %d" % k


def test_c2_wfm_every_stream_in_three_calls(gpu, port):
    """The timed shape as a STREAM: the same 1024 x 2 400 256 batch handed over in three calls of unequal size (1139 + 600 + 605 chunks), s16 only, every call into
    its own aligned rows -- what bench.py's timed loop does from its second step on: the later calls start in the history, their first audio sample falls anywhere in a
    128-byte line of the output row, and every one of the 4 long time segments of a launch fills and flushes the loader waves' line registers several times.  Replicas
    bit-identical; the 16 distinct rows +-1 LSB against the oracle's stream on every sample."""
    import torch
    S, T = 1024, 2344 * 1024
    L = gpu.L
    taps = gpu.firdes_lowpass_f(gpu.firdes_filter_len(0.05), 0.5 / 10, "HAMMING")
    sigs = [wfm_signal_u8(2100 + k, T) for k in range(16)]
    x, idx = _replicated(sigs, S, T)
    calls = [1139 * 1024, 600 * 1024, 605 * 1024]
    n_audio_max = (max(calls) // 50 + 64 + 63) // 64 * 64
    out = torch.zeros((S, n_audio_max), dtype=torch.int16, device="cuda")
    w = L.csdr_amd_wfm_create(gpu.h, S, -0.085, 10, taps.ctypes.data_as(C.c_void_p), taps.size, 5, 50e-6, 48000, max(calls))
    assert w, gpu.err()
    parts, pos = [], 0
    try:
        for k in calls:
            n = L.csdr_amd_wfm_process(w, x.data_ptr() + 2 * pos, 2 * T, k, out.data_ptr(), None, n_audio_max)
            assert n > 0, gpu.err()
            gpu.sync()
            parts.append(out[:, :n].clone())
            pos += k
        assert L.csdr_amd_wfm_kernel_name(w).decode() == "k_wfm_mfma_seq"
    finally:
        L.csdr_amd_wfm_destroy(w)
    got_all = torch.cat(parts, dim=1)
    n = got_all.shape[1]
    first = {int(k): int(np.nonzero(idx == k)[0][0]) for k in range(16)}
    ref_rows = got_all[torch.tensor([first[int(k)] for k in idx], device="cuda")]
    assert torch.equal(got_all, ref_rows), "replicas of one signal differ between stream blocks / columns"
    for k in range(16):
        ps, _ = port.wfm_chain(sigs[k], -0.085, 10, taps)
        got = got_all[first[k]].cpu().numpy()
        m = min(ps.size, got.size)
        assert 0 <= n - ps.size <= 2 and vc.s16_diff(got[:m], ps[:m]).max() <= 1, "signal %d" % k


def test_c5_nfm_every_channel(gpu, port):
    """bench_nfm.py's timed shape, all of it, as SURVEY.md section 8d defines config 5: 512 (stream, shift_rate) PAIRS x 2 400 256 samples -- every channel its own
    rate (verify_configs.c5_rates: 512 distinct values incl. the drifting 0.05 / 0.25 / -0.05 / -0.25, 0 and +-0.4999), every channel its own narrow-band FM signal
    at -rate.  32 channels spread over the batch (the special rates among them) +-1 LSB on every sample against the oracle's stage-by-stage chain; every channel
    must have produced the same count and a live signal (RMS within 3 dB of the checked ones: a channel demodulated at a wrong rate is noise at full scale)."""
    import torch
    S, T, D = 512, 2344 * 1024, 50
    L = gpu.L
    taps = gpu.firdes_lowpass_f(gpu.firdes_filter_len(0.005), 0.5 / D, "HAMMING")
    rates = vc.c5_rates(S)
    assert len(set(rates.tolist())) == S
    # one modulating signal per channel would cost 512 x 2.4 M numpy samples: 16 base signals at offset 0, moved to -rate of the channel by a phase ramp before quantising
    base = [_nfm_baseband(3100 + k, T) for k in range(16)]
    n_out_max = (T // D + 2048 + 63) // 64 * 64
    x = torch.empty((S, 2 * T), dtype=torch.uint8, device="cuda")
    tt = torch.arange(T, device="cuda", dtype=torch.float64)
    for c in range(S):
        sig = torch.from_numpy(base[c % 16]).cuda() * torch.exp(2j * np.pi * (-float(rates[c])) * tt)
        iq = torch.view_as_real(sig).reshape(-1).to(torch.float32)
        x[c] = torch.clamp(torch.round(127.5 * (iq + 1)), 0, 255).to(torch.uint8)
    out = torch.zeros((S, n_out_max), dtype=torch.int16, device="cuda")
    obj = L.csdr_amd_nfm_create_rates(gpu.h, S, rates.ctypes.data_as(C.c_void_p), D, taps.ctypes.data_as(C.c_void_p), taps.size, 48000, 1024, 1.0, 1.0, T)
    assert obj, gpu.err()
    try:
        n = L.csdr_amd_nfm_process(obj, x.data_ptr(), 2 * T, T, out.data_ptr(), None, n_out_max)
        assert n > 0, gpu.err()
        gpu.sync()
        assert L.csdr_amd_ddc_kernel_name(L.csdr_amd_nfm_front_end(obj)) == b"k_ddc_mfma" and L.csdr_amd_ddc_fallback(L.csdr_amd_nfm_front_end(obj)) == 0
    finally:
        L.csdr_amd_nfm_destroy(obj)
    nfm_taps = gpu.nfm_taps(48000)
    check = sorted(set(vc.pick_rows(S, want=25)) | {5, S // 3 + 1, (2 * S) // 3 + 2, S - 2, 7, 1, S - 5})
    for c in check:
        ps, _ = port.nfm_chain(x[c].cpu().numpy(), float(rates[c]), nfm_taps, D, 0.005, 1024)
        got = out[c, :n].cpu().numpy()
        assert n == ps.size and vc.s16_diff(got, ps).max() <= 1, "channel %d rate %g: %d" % (c, rates[c], vc.s16_diff(got, ps).max())
    rms = out[:, 4096:n].to(torch.float32).pow(2).mean(dim=1).sqrt().cpu().numpy()
    ref = np.median(rms[check])
    assert (rms > ref / 1.41).all() and (rms < ref * 1.41).all(), (int(rms.argmin()), float(rms.min()), int(rms.argmax()), float(rms.max()), float(ref))


def _nfm_baseband(seed, n, deviation=5e3 / 2.4e6):
    """complex128 narrow-band FM signal at offset 0 (tests_helpers.nfm_signal_u8 before the frequency offset and the quantiser)"""
    rng = np.random.default_rng(seed)
    msg = np.sin(2 * np.pi * 1e3 / 2.4e6 * np.arange(n)) + 0.3 * np.convolve(rng.uniform(-1, 1, n + 199), np.ones(200) / 200, "valid")
    return 0.7 * np.exp(2j * np.pi * np.cumsum(deviation * msg)) + 0.01 * (rng.normal(size=n) + 1j * rng.normal(size=n))


def _wfm_baseband(seed, n):
    """complex128 FM broadcast-like signal at offset 0 (tests_helpers.wfm_signal_u8 before the frequency offset and the quantiser)"""
    rng = np.random.default_rng(seed)
    msg = np.sin(2 * np.pi * 1e3 / 2.4e6 * np.arange(n)) + 0.3 * rng.uniform(-1, 1, n)
    return 0.7 * np.exp(2j * np.pi * np.cumsum(0.03125 * msg)) + 0.01 * (rng.normal(size=n) + 1j * rng.normal(size=n))


def c2_rates(n_streams=1024):
    """bench.py::operating_points' 1024 distinct rates (-0.45 + 0.9 (s + 0.5) / S) with the awkward ones of verify_configs.c5_rates at fixed streams:
    +-0.05 / +-0.25 (the reference's float phasor recurrence drifts systematically there), 0 and +-0.4999 (the most phase wraps per chunk)."""
    r = (-0.45 + 0.9 * (np.arange(n_streams) + 0.5) / n_streams).astype(f32)
    special = {5: 0.05, n_streams // 3 + 1: 0.25, (2 * n_streams) // 3 + 2: -0.05, n_streams - 2: -0.25, 7: 0.0, 1: 0.4999, n_streams - 5: -0.4999}
    for k, v in special.items():
        r[k] = v
    return r, sorted(special)


def _stage_chain(port, u8, rates_at, taps):
    """The WFM chain stage by stage with the shift rate changing at the given samples (phase carried: `csdr shift_addition_cc --fifo`, csdr.c:881-923):
    (s16, float audio).  fractional_decimator_ff 5 == x[5 k + 10] (exact at an integer rate)."""
    xf = port.convert_u8_f(u8).view(c64)
    parts, ph = [], 0.0
    for i, (pos, r) in enumerate(rates_at):
        end = rates_at[i + 1][0] if i + 1 < len(rates_at) else xf.size
        y, ph = port.shift_addition_cc(xf[pos:end], r, phase=ph)
        parts.append(y)
    dec = port.fir_decimate_cc(np.concatenate(parts), 10, taps)
    dem, _ = port.fmdemod_quadri_cf(dec)
    aud = port.deemphasis_wfm_ff(dem[10::5], 50e-6, 48000)[0]
    return port.convert_f_s16(aud), aud


def test_c2_wfm_every_stream_with_its_own_rate(gpu, port):
    """The per-stream-rate WFM kernel (`k_wfm_mfma_seq<true>`, csdr_amd_wfm_create_rates) at the size bench.py::operating_points times and README advertises
    (VERDICT r4 item 1): 1024 (stream, shift_rate) pairs x 2 400 256 samples -- the reference's unit of work, one `csdr shift_addition_cc <rate> | ...` chain per
    client (csdr.c:876-923, libcsdr_gpl.c:27-52, ddcd_old.h:51-61).  1024 distinct rates (c2_rates), every stream one of 16 FM signals re-centred at -rate.
    THREE consecutive calls on one object: the full size twice (2344-chunk seed chains, seed tables switched on the side stream, state carried over a full-size
    boundary, the line-collecting store path at full row length) and a 600-chunk call, with csdr_amd_wfm_set_rate on three checked streams between the first and the
    second call (their signals move with them).  40 rows -- spread over every workgroup round of the 1024 one-stream workgroups, the special rates and the retuned
    streams among them -- are held on EVERY sample of all three calls to +-1 LSB (s16) and to <= 1e-5 relative RMS (float audio) against the oracle's seven stages;
    every other stream must have produced a live demodulated signal of the same level (a stream mixed with a wrong rate is noise at full scale)."""
    import torch
    S, T = 1024, 2344 * 1024
    T3 = 600 * 1024
    L = gpu.L
    taps = gpu.firdes_lowpass_f(gpu.firdes_filter_len(0.05), 0.5 / 10, "HAMMING")
    rates, special = c2_rates(S)
    assert len(set(rates.tolist())) == S
    base = [torch.from_numpy(_wfm_baseband(2500 + k, T)).cuda() for k in range(16)]
    tt = torch.arange(T, device="cuda", dtype=torch.float64)

    def row_bytes(sig, rate, n=T):
        z = sig[:n] * torch.exp(2j * np.pi * (-float(rate)) * tt[:n])
        iq = torch.view_as_real(z).reshape(-1).to(torch.float32)
        return torch.clamp(torch.round(127.5 * (iq + 1)), 0, 255).to(torch.uint8)

    x = torch.empty((S, 2 * T), dtype=torch.uint8, device="cuda")
    for c in range(S):
        x[c] = row_bytes(base[c % 16], rates[c])
    n_audio_max = (T // 50 + 64 + 63) // 64 * 64
    out = torch.zeros((S, n_audio_max), dtype=torch.int16, device="cuda")
    outf = torch.zeros((S, n_audio_max), dtype=torch.float32, device="cuda")
    retuned = {40: 0.3, 517: -0.11, S - 9: 0.05}                     # stream -> new rate from the second call's first sample
    check = sorted(set(vc.pick_rows(S, want=30)) | set(special) | set(retuned))
    assert len(check) >= 36
    kept = {c: [x[c].cpu().numpy().copy()] for c in check}            # the bytes each checked row was fed, call by call
    got_s = {c: [] for c in check}; got_f = {c: [] for c in check}
    w = L.csdr_amd_wfm_create_rates(gpu.h, S, rates.ctypes.data_as(C.c_void_p), 10, taps.ctypes.data_as(C.c_void_p), taps.size, 5, 50e-6, 48000, T)
    assert w, gpu.err()
    counts = []
    try:
        for call, k in enumerate((T, T, T3)):
            if call == 1:
                for st, r in retuned.items():
                    assert L.csdr_amd_wfm_set_rate(w, st, r) == 0, gpu.err()
                    assert abs(L.csdr_amd_wfm_get_rate(w, st) - r) < 1e-7
                    x[st] = row_bytes(base[(st + 3) % 16], r)
                torch.cuda.synchronize()
            if call >= 1:
                for c in check:
                    kept[c].append(x[c, :2 * k].cpu().numpy().copy())
            n = L.csdr_amd_wfm_process(w, x.data_ptr(), 2 * T, k, out.data_ptr(), outf.data_ptr(), n_audio_max)
            assert n > 0, gpu.err()
            gpu.sync()
            assert L.csdr_amd_wfm_kernel_name(w).decode() == "k_wfm_mfma_seq" and not L.csdr_amd_wfm_fallback(w)
            counts.append(n)
            for c in check:
                got_s[c].append(out[c, :n].cpu().numpy().copy()); got_f[c].append(outf[c, :n].cpu().numpy().copy())
            if call == 0:
                rms = out[:, 64:n].to(torch.float32).pow(2).mean(dim=1).sqrt().cpu().numpy()
    finally:
        L.csdr_amd_wfm_destroy(w)
    assert counts[0] >= 48000 and counts[1] >= 48000 and counts[2] >= T3 // 50 - 2
    worst_lsb, worst_rms = 0, 0.0
    for c in check:
        u8 = np.concatenate(kept[c])
        r0 = float(rates[c])
        sched = [(0, r0)] + ([(T, float(retuned[c]))] if c in retuned else [])
        ps, pf = _stage_chain(port, u8, sched, taps)
        if c not in retuned:                                          # the oracle's one-call chain must say the same as its stages (pins _stage_chain itself)
            ps2, _ = port.wfm_chain(u8, r0, 10, taps)
            m2 = min(ps.size, ps2.size)
            assert m2 >= ps.size - 12 and np.array_equal(ps[:m2], ps2[:m2])
        g = np.concatenate(got_s[c]); gf = np.concatenate(got_f[c])
        m = min(ps.size, g.size)
        assert -12 <= g.size - ps.size <= 2, (c, g.size, ps.size)
        d = int(vc.s16_diff(g[:m], ps[:m]).max()); e = vc.relrms(gf[:m], pf[:m])
        worst_lsb = max(worst_lsb, d); worst_rms = max(worst_rms, e)
        assert d <= 1, "stream %d rate %g: %d LSB (first at audio sample %d)" % (c, r0, d, int(np.nonzero(vc.s16_diff(g[:m], ps[:m]) > 1)[0][0]))
        assert e <= TOL, "stream %d rate %g: float audio rel. RMS %g" % (c, r0, e)
    ref = np.median(rms[check])
    assert (rms > ref / 1.41).all() and (rms < ref * 1.41).all(), (int(rms.argmin()), float(rms.min()), int(rms.argmax()), float(rms.max()), float(ref))
    print("c2 per-stream rates: %d rows x %d audio samples, worst %d LSB, worst rel. RMS %.2e" % (len(check), sum(counts), worst_lsb, worst_rms))


def test_c5_nfm_every_channel_uniform_rate(gpu, port):
    """The same shape with ONE rate for all channels (the shared-weights kernel, csdr_amd_nfm_create): 512 channels carry one of 16 narrow-band FM signals; replicas
    bit-identical, the 16 distinct rows +-1 LSB against the oracle's stage-by-stage chain on every sample."""
    import torch
    S, T, D = 512, 2344 * 1024, 50
    L = gpu.L
    taps = gpu.firdes_lowpass_f(gpu.firdes_filter_len(0.005), 0.5 / D, "HAMMING")
    sigs = [nfm_signal_u8(3100 + k, T, offset=0.05) for k in range(16)]
    x, idx = _replicated(sigs, S, T)
    n_out_max = (T // D + 2048 + 63) // 64 * 64
    out = torch.zeros((S, n_out_max), dtype=torch.int16, device="cuda")
    obj = L.csdr_amd_nfm_create(gpu.h, S, -0.05, D, taps.ctypes.data_as(C.c_void_p), taps.size, 48000, 1024, 1.0, 1.0, T)
    assert obj, gpu.err()
    try:
        n = L.csdr_amd_nfm_process(obj, x.data_ptr(), 2 * T, T, out.data_ptr(), None, n_out_max)
        assert n > 0, gpu.err()
        gpu.sync()
    finally:
        L.csdr_amd_nfm_destroy(obj)
    first = {int(k): int(np.nonzero(idx == k)[0][0]) for k in range(16)}
    ref_rows = out[torch.tensor([first[int(k)] for k in idx], device="cuda")]
    assert torch.equal(out[:, :n], ref_rows[:, :n]), "replicas of one signal differ between channel blocks / columns"
    nfm_taps = gpu.nfm_taps(48000)
    for k in range(16):
        ps, _ = port.nfm_chain(sigs[k], -0.05, nfm_taps, D, 0.005, 1024)
        got = out[first[k], :n].cpu().numpy()
        assert n == ps.size and vc.s16_diff(got, ps).max() <= 1, "signal %d" % k


def test_c4_bank_at_the_timed_size(gpu, port):
    """bench_fastddc.py's timed call: the bank object, 256 channels x 64 blocks in ONE process() (the fold kernel with the second forward pass inside, two
    32-block accumulator tiles, chain riders) -- 40 channels spread over every 32-channel wave tile against the oracle"""
    tbw, D, nch, nb = 0.001, 256, 256, 64
    ddc, _ = gpu.fastddc_init(tbw, D, 0.0)
    rng = np.random.default_rng(64)
    x = (rng.uniform(-1, 1, ddc.input_size * nb) + 1j * rng.uniform(-1, 1, ddc.input_size * nb)).astype(c64)
    rates = vc.c4_rates(nch)
    check = sorted(set(list(range(0, 256, 8)) + [1, 31, 33, 127, 128, 129, 254, 255]))
    _, want = vc.fastddc_oracle_channels(x, tbw, D, rates, check)
    outs = gpu.fastddc_bank(x, tbw, D, rates, blocks_per_call=nb)
    assert gpu.last_ddc_kernel == "k_ddc_gemm3" and len(check) >= 36
    for c in check:
        assert outs[c].size == want[c].size and vc.relrms(outs[c], want[c]) < TOL, "channel %d" % c


@pytest.mark.parametrize("nch,nb", [(32, 64), (5, 64), (33, 40), (64, 64), (100, 7), (128, 96)])
def test_c4_bank_of_few_channels(gpu, port, nch, nb):
    """config 4's geometry with few channels -- a small bank, or what one rank of a channel-sharded bank folds (32 of 256 at eight ranks): the fold kernel of one /
    two / four waves per residue and 32 blocks (k_ddc_gemm3n), ragged channel and block tiles, every channel against the oracle"""
    tbw, D = 0.001, 256
    ddc, _ = gpu.fastddc_init(tbw, D, 0.0)
    rng = np.random.default_rng(1000 * nch + nb)
    x = (rng.uniform(-1, 1, ddc.input_size * nb) + 1j * rng.uniform(-1, 1, ddc.input_size * nb)).astype(c64)
    rates = vc.c4_rates(256)[rng.permutation(256)[:nch]]
    check = list(range(nch)) if nch <= 40 else sorted(set(list(range(0, nch, 5)) + [1, 31, 32, 33, 63, 64, 65, nch - 2, nch - 1]) & set(range(nch)))
    _, want = vc.fastddc_oracle_channels(x, tbw, D, rates, check)
    outs = gpu.fastddc_bank(x, tbw, D, rates, blocks_per_call=nb)
    assert gpu.last_ddc_kernel == "k_ddc_gemm3n"
    for c in check:
        assert outs[c].size == want[c].size and vc.relrms(outs[c], want[c]) < TOL, "channel %d of %d" % (c, nch)


class _Plan(C.Structure):        # fft_fftw.h:14-20
    _fields_ = [("size", C.c_int), ("input", C.c_void_p), ("output", C.c_void_p), ("plan", C.c_void_p)]


@pytest.mark.parametrize("n", [16, 1024, 4096])
def test_fft_plan_layer_real_transforms(gpu, n):
    """make_fft_r2c / make_fft_c2r (fft_fftw.c:16-35: FFTW's r2c half spectrum of n/2+1 bins, c2r unnormalised inverse) through the drop-in symbols."""
    import csdr_amd
    L = C.CDLL(csdr_amd.LIB_PATH)
    L.make_fft_r2c.restype = C.POINTER(_Plan); L.make_fft_r2c.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.make_fft_c2r.restype = C.POINTER(_Plan); L.make_fft_c2r.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.fft_execute.argtypes = [C.POINTER(_Plan)]; L.fft_destroy.argtypes = [C.POINTER(_Plan)]
    rng = np.random.default_rng(n)
    x = rng.uniform(-1, 1, n).astype(f32)
    X = np.zeros(n // 2 + 1, c64)
    p = L.make_fft_r2c(n, x.ctypes.data, X.ctypes.data, 0)
    assert p.contents.size == n and p.contents.input == x.ctypes.data and p.contents.output == X.ctypes.data
    L.fft_execute(p)
    want = np.fft.rfft(x.astype(np.float64))
    assert vc.relrms(X, want) < TOL
    y = np.zeros(n, f32)
    Xin = want.astype(c64)
    q = L.make_fft_c2r(n, Xin.ctypes.data, y.ctypes.data, 0)
    L.fft_execute(q)
    assert vc.relrms(y, np.fft.irfft(want, n) * n) < TOL                      # FFTW's c2r is unnormalised
    L.fft_destroy(p); L.fft_destroy(q)
