"""Checker side of the `--verify` legs of bench.py / bench_nfm.py / bench_fastddc.py and of tests/test_configs_gpu.py: the outputs the
HIP path produced at the FULL BASELINE.json sizes (1024 WFM streams x 2 400 256 samples, 512 NFM channels, 256 fastddc channels at
fft 65536) against the CPU oracle on the same bytes.  TEST INFRASTRUCTURE: imports oracle/, never on the timed path (the benches call
it after the timed region; nothing here is measured)."""
import ctypes as C
import numpy as np

f32 = np.float32
c64 = np.complex64


def pick_rows(n_rows, group=16, want=16):
    """Rows spread over every region of the batch: first / last row of the first, middle and last `group`-row stream block, the very
    last row, and evenly spaced rows in between (each lands in a different stream block when n_rows / want >= group)."""
    if n_rows <= want:
        return list(range(n_rows))
    rows = {0, group - 1, group, n_rows // 2 - 1, n_rows // 2, n_rows - group, n_rows - 1}
    k = 1
    while len(rows) < want:
        rows.add(min(n_rows - 1, (k * n_rows) // want + (k * 5) % group))
        k += 1
    return sorted(r for r in rows if 0 <= r < n_rows)[:want]


def s16_diff(a, b):
    """Difference modulo 2^16 (convert_f_s16 wraps out-of-range values, libcsdr.c:2378-2381: a +-1 LSB difference next to the wrap
    point is a 65535 difference of the int16 values)."""
    d = (a.astype(np.int64) - b.astype(np.int64) + 32768) % 65536 - 32768
    return np.abs(d)


def summarize(diffs, n_expected, n_got):
    d = np.concatenate(diffs) if diffs else np.zeros(0, np.int64)
    return {"samples_compared": int(d.size), "max_abs_diff_lsb": int(d.max()) if d.size else 0,
            "frac_nonzero": float((d > 0).mean()) if d.size else 0.0, "frac_over_1_lsb": float((d > 1).mean()) if d.size else 0.0,
            "rows_expected_len": int(n_expected), "rows_got_len": int(n_got)}


def verify_wfm(ctx, w, x, out_s16, S, T, pitch, n_audio_max, taps, shift_rate=-0.085, decimation=10, rows=None, strict_rows=(), out_f32=None):
    """Fresh-state pass of the SAME object / buffers / launch configuration as the timed steps (csdr_amd_wfm_reset, one
    csdr_amd_wfm_process over all S streams x T samples), then `rows` full audio rows against oracle.port().wfm_chain on the same bytes.
    Gate on the bench's own input (i.i.d. uniform u8 = band-limited noise after the FIR): < 5 % of the s16 samples differ at all, < 0.5 % by
    more than one LSB.  Why not "max <= 1 LSB": fmdemod_quadri_cf divides by I^2+Q^2 (libcsdr.c:1040-1071); for a noise input that
    denominator is exponentially distributed, P(|y|^2 < t) = t / E|y|^2, and an absolute error delta of y becomes K |y_prev| delta / |y|^2
    in the output: ANY two float implementations that agree to 1e-7 (the compiled reference's own -ffast-math rcpps path vs its plain-C
    path included) differ by more than one s16 LSB on about K delta / (3e-5 / 0.29 x sigma) ~ 4e-4 of the samples, and the one-pole
    de-emphasis spreads each such event over the next ~10-20 samples.  `strict_rows` (rows whose bytes the caller has replaced by a real
    FM signal, where the denominator stays near 0.49) are held to max <= 1 LSB.
    The HIP path may emit the last one or two audio samples of a block EARLIER than the reference's fractional_decimator_ff, whose loop
    waits for num_poly_points samples of look-ahead it never uses at an integer rate (libcsdr.c:763): a stream sees identical samples, only
    the block boundary moves, so 0 <= got - expected <= 2 is accepted and the common prefix compared.
    shift_rate: one float, or one per stream (an object made by csdr_amd_wfm_create_rates; the caller's strict rows then carry a signal at -rate of that stream).
    out_f32 (a [S, n_audio_max] float32 device buffer): the pass also writes the float audio (the chain's output in front of convert_f_s16) and the strict rows are
    ALSO held to north_star's float gate, relative RMS <= 1e-5 over the whole row against the oracle's float audio (`strict_rows_max_rel_rms`)."""
    import oracle
    port = oracle.port()
    L = ctx.L
    rc = L.csdr_amd_wfm_reset(w)
    assert rc == 0, ctx.err()
    n = L.csdr_amd_wfm_process(w, x.data_ptr(), pitch, T, out_s16.data_ptr(), out_f32.data_ptr() if out_f32 is not None else None, n_audio_max)
    assert n >= 0, ctx.err()
    ctx.sync()
    rows = pick_rows(S) if rows is None else rows
    diffs = []; n_ref = -1; strict_max = 0; strict_rms = 0.0; strict_n = 0
    for r in list(rows) + list(strict_rows):
        u8 = x[r, :2 * T].cpu().numpy()
        ps, pf = port.wfm_chain(u8, float(shift_rate[r]) if np.ndim(shift_rate) else shift_rate, decimation, taps)
        got = out_s16[r, :n].cpu().numpy()
        n_ref = ps.size
        m = min(ps.size, got.size)
        d = s16_diff(got[:m], ps[:m])
        if r in strict_rows:
            strict_max = max(strict_max, int(d.max()) if d.size else 0); strict_n += int(d.size)
            if out_f32 is not None and m:
                strict_rms = max(strict_rms, relrms(out_f32[r, :m].cpu().numpy(), pf[:m]))
        else:
            diffs.append(d)
    res = summarize(diffs, n_ref, n)
    res["rows"] = list(rows); res["strict_rows"] = list(strict_rows); res["strict_rows_max_abs_diff_lsb"] = strict_max; res["strict_rows_samples_compared"] = strict_n
    if out_f32 is not None:
        res["strict_rows_max_rel_rms"] = float("%.3g" % strict_rms)
    res["kernel"] = L.csdr_amd_wfm_kernel_name(w).decode()
    res["gate"] = ("frac_nonzero < 0.05, frac_over_1_lsb < 0.005 (noise input: ill-conditioned demodulator, see tests/verify_configs.py), strict rows max <= 1 LSB"
                   + (" and float audio rel. RMS <= 1e-5" if out_f32 is not None else "") + ", 0 <= got_len - expected_len <= 2")
    res["ok"] = bool(0 <= n - n_ref <= 2 and res["frac_over_1_lsb"] < 5e-3 and res["frac_nonzero"] < 0.05 and strict_max <= 1 and strict_rms <= 1e-5
                     and (strict_n > 0 or not strict_rows))
    return res


def verify_wfm_ring(ctx, ring, S, T, taps, shift_rate=-0.085, decimation=10, n_blocks=6, noise_rows=12, seed=7200):
    """The resident form (csdr_amd_wfm_ring_*) at the bench's shape: reset, n_blocks consecutive blocks of S streams -- four rows a real FM signal (held to +-1 LSB on every
    sample), the rest i.i.d. noise of which `noise_rows` rows spread over the stream groups go through verify_wfm's statistical gate -- written into the input ring slot by
    slot, posted, collected, against oracle.port().wfm_chain on each checked row's n_blocks * T samples."""
    import oracle
    import torch
    from tests_helpers import wfm_signal_u8
    port = oracle.port()
    L = ctx.L
    assert n_blocks <= L.csdr_amd_wfm_ring_slots(ring) - 2
    assert L.csdr_amd_wfm_ring_reset(ring) == 0, ctx.err()
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    x = torch.randint(0, 256, (S, 2 * T * n_blocks), dtype=torch.uint8, device="cuda", generator=g)
    strict_rows = sorted({r for r in (5, S // 3 + 1, (2 * S) // 3 + 2, S - 2) if 0 <= r < S})
    for k, r in enumerate(strict_rows):
        x[r] = torch.from_numpy(wfm_signal_u8(seed + 1 + k, T * n_blocks, offset=-shift_rate)).cuda()
    rows = [r for r in pick_rows(S, want=noise_rows) if r not in strict_rows]
    pitch = C.c_size_t(0); opitch = C.c_size_t(0)
    for k in range(n_blocks):
        assert L.csdr_amd_wfm_ring_acquire(ring, k, 0.0) == 0, ctx.err()
        pi = L.csdr_amd_wfm_ring_input(ring, k, C.byref(pitch))
        blk = torch.zeros((S, pitch.value), dtype=torch.uint8, device="cuda"); blk[:, :2 * T] = x[:, 2 * T * k:2 * T * (k + 1)]
        torch.cuda.synchronize()
        assert L.csdr_amd_d2d(ctx.h, pi, blk.data_ptr(), blk.numel()) == 0 and L.csdr_amd_ctx_sync(ctx.h) == 0, ctx.err()
        assert L.csdr_amd_wfm_ring_submit(ring) == k, ctx.err()
    outs = []
    for k in range(n_blocks):
        na = L.csdr_amd_wfm_ring_wait(ring, k, 0.0)
        assert na > 0, ctx.err()
        po = L.csdr_amd_wfm_ring_output(ring, k, C.byref(opitch))
        buf = np.empty((S, opitch.value), np.int16)
        assert L.csdr_amd_d2h(ctx.h, buf.ctypes.data_as(C.c_void_p), po, buf.nbytes) == 0, ctx.err()
        outs.append(buf[:, :na].copy())
    y = np.concatenate(outs, axis=1)
    diffs = []; strict_max = 0; strict_n = 0; n_ref = -1
    for r in rows + strict_rows:
        ps, _ = port.wfm_chain(x[r].cpu().numpy(), shift_rate, decimation, taps)
        n_ref = ps.size
        m = min(ps.size, y.shape[1])
        d = s16_diff(y[r, :m], ps[:m])
        if r in strict_rows:
            strict_max = max(strict_max, int(d.max())); strict_n += int(d.size)
        else:
            diffs.append(d)
    res = summarize(diffs, n_ref, y.shape[1])
    res["rows"] = rows; res["strict_rows"] = strict_rows; res["strict_rows_max_abs_diff_lsb"] = strict_max; res["strict_rows_samples_compared"] = strict_n
    res["blocks"] = n_blocks
    res["gate"] = "frac_nonzero < 0.05, frac_over_1_lsb < 0.005 (noise rows), strict rows max <= 1 LSB, 0 <= got_len - expected_len <= 2"
    res["ok"] = bool(0 <= y.shape[1] - n_ref <= 2 and res["frac_over_1_lsb"] < 5e-3 and res["frac_nonzero"] < 0.05 and strict_max <= 1 and strict_n > 0)
    return res


def verify_nfm(ctx, obj, x, out_s16, S, T, pitch, n_out_max, shift_rate=-0.05, decimation=50, tbw=0.005, audio_rate=48000, agc_block=1024, rows=None, strict_rows=()):
    """Same for the NFM chain object: reset, one pass over all S channels, `rows` full s16 rows against oracle.port().nfm_chain (same gates
    and the same reason as verify_wfm; limit_ff bounds the ill-conditioned samples, so the largest differences are tens of LSB, not thousands)."""
    import oracle
    port = oracle.port()
    L = ctx.L
    rc = L.csdr_amd_nfm_reset(obj)
    assert rc == 0, ctx.err()
    n = L.csdr_amd_nfm_process(obj, x.data_ptr(), pitch, T, out_s16.data_ptr(), None, n_out_max)
    assert n >= 0, ctx.err()
    ctx.sync()
    rows = pick_rows(S) if rows is None else rows
    nfm_taps = ctx.nfm_taps(audio_rate)
    diffs = []; n_ref = -1; strict_max = 0
    for r in list(rows) + list(strict_rows):
        u8 = x[r, :2 * T].cpu().numpy()
        ps, _ = port.nfm_chain(u8, float(shift_rate[r]) if np.ndim(shift_rate) else shift_rate, nfm_taps, decimation, tbw, agc_block)   # (a rate per channel: an array)
        n_ref = ps.size
        got = out_s16[r, :n].cpu().numpy()
        m = min(ps.size, got.size)
        d = s16_diff(got[:m], ps[:m])
        if r in strict_rows:
            strict_max = max(strict_max, int(d.max()))
        else:
            diffs.append(d)
    res = summarize(diffs, n_ref, n)
    res["rows"] = list(rows); res["strict_rows"] = list(strict_rows); res["strict_rows_max_abs_diff_lsb"] = strict_max
    res["kernel"] = L.csdr_amd_ddc_kernel_name(L.csdr_amd_nfm_front_end(obj)).decode()
    res["gate"] = "frac_nonzero < 0.05, frac_over_1_lsb < 0.005 (noise input), strict rows max <= 1 LSB, got_len == expected_len"
    res["ok"] = bool(n == n_ref and n > 0 and res["frac_over_1_lsb"] < 5e-3 and res["frac_nonzero"] < 0.05 and strict_max <= 1)
    return res


def c5_rates(n_channels=512):
    """Config 5 as SURVEY.md section 8d states it: one (stream, shift_rate) pair per channel.  Rates spread over -0.45 .. 0.45 (none equal), with the awkward ones a
    receiver bank meets put at fixed channels: 0.05 / 0.25 / -0.05 / -0.25 (the reference's float phasor recurrence drifts systematically there, ddc_mfma.hip), 0 (no
    shift) and the largest rates (410 phase wraps per chunk)."""
    r = (-0.45 + 0.9 * (np.arange(n_channels) + 0.5) / n_channels).astype(f32)
    special = {5: 0.05, n_channels // 3 + 1: 0.25, (2 * n_channels) // 3 + 2: -0.05, n_channels - 2: -0.25, 7: 0.0, 1: 0.4999, n_channels - 5: -0.4999}
    for k, v in special.items():
        if 0 <= k < n_channels:
            r[k] = v
    return r


def c4_rates(n_channels=256):
    """channel c at shift_rate = -0.5 + (c + 0.5)/C"""
    return (-0.5 + (np.arange(n_channels) + 0.5) / n_channels).astype(f32)


def fastddc_oracle_channels(x, tbw, decimation, rates, channels):
    """Oracle outputs of the listed channels for the wideband input x (whole blocks only): (spectra, {channel: samples})."""
    import oracle
    port = oracle.port()
    pd, err = port.fastddc_init(tbw, decimation, 0.0)
    assert err == 0
    spec = port.fastddc_fwd_cc(x, pd)
    outs = {}
    for c in channels:
        pdc, _ = port.fastddc_init(tbw, decimation, float(rates[c]))
        outs[c] = port.fastddc_inv_cc(spec, pdc, port.fastddc_taps_fft(pdc, float(rates[c]), decimation))
    return spec, outs


def relrms(a, b):
    a = np.asarray(a).astype(np.complex128).ravel(); b = np.asarray(b).astype(np.complex128).ravel()
    den = np.sqrt((np.abs(b) ** 2).sum())
    return float(np.sqrt((np.abs(a - b) ** 2).sum()) / den) if den else float(np.abs(a).max() if a.size else 0.0)
