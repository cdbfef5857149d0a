#!/usr/bin/env python3
"""tests/golden/make_golden.py -- golden input/output vectors of the hot path, produced by the UNMODIFIED reference.

Runs only where /root/reference is mounted: it drives oracle/_ref/libcsdr_ref.so (the reference's own sources compiled with the
reference's own flags by oracle/Makefile, FFTW replaced by the double-precision shim) through oracle.Ref, and the reference CLI
binary oracle/_ref/csdr for the two CLI-only loops.  Output: tests/golden/ref_vectors.npz (seeded inputs + reference outputs,
a few hundred KB).  tests/test_golden.py replays the inputs through the CPU oracle (anywhere) and through the HIP path (GPU box,
where neither /root/reference nor a toolchain for it is needed) and compares with these outputs.

    python tests/golden/make_golden.py        # rewrites ref_vectors.npz
    python tests/golden/make_golden.py nfm    # rewrites nfm_cli_vectors.npz (the NFM chain as a process pipeline of the reference binary)
    python tests/golden/make_golden.py baseline   # rewrites baseline_vectors.npz: the reference at the BASELINE geometries (C3 fft 65536, C4 D = 256 / tbw 0.001, C5 rates)
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

c64, f32 = np.complex64, np.float32


def crand(rng, n):
    return (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(c64)


def make_nfm_cli():
    """tests/golden/nfm_cli_vectors.npz: README.md:87 (NFM, BASELINE config 5) as EIGHT processes of the reference binary connected by pipes."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tests_helpers import nfm_signal_u8
    cli = os.path.join(ROOT, "oracle", "_ref", "csdr")
    assert os.path.exists(cli), "oracle/_ref/csdr is missing"
    iq = nfm_signal_u8(2026, 1024 * 200, offset=-0.11)
    cmds = ("convert_u8_f", "shift_addition_cc 0.11", "fir_decimate_cc 50 0.005 HAMMING", "fmdemod_quadri_cf", "limit_ff", "deemphasis_nfm_ff 48000", "fastagc_ff", "convert_f_s16")
    pipe = " | ".join("%s %s" % (cli, c) for c in cmds)
    out = subprocess.run(pipe, shell=True, input=iq.tobytes(), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=120).stdout
    s16 = np.frombuffer(out[:len(out) // 2 * 2], np.int16)
    # the processes repeat / hold back their last blocks at EOF: keep the part every complete run agrees on
    keep = ((1024 * 200 - 801) // 50 + 1 + 1024 - 201) // 1024 * 1024
    assert s16.size >= keep
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "nfm_cli_vectors.npz"), nfm_iq_u8=iq, nfm_cli_s16=s16[:keep].copy(), shift_rate=np.float32(0.11))
    print("wrote nfm_cli_vectors.npz: %d input samples, %d audio samples" % (iq.size // 2, keep))


# ---------------------------------------------------------------- the reference at the BASELINE geometries (inputs are regenerated from their seeds, never stored)
C3_TAPS = (1023, 4095)
C3_BLOCKS = 3
C4_CHANNELS = (0, 97, 255)          # of 256, channel c at shift_rate = -0.5 + (c + 0.5)/256
C4_BLOCKS = 5
C5_RATES = (0.25, 0.05, -0.4321)
C5_SAMPLES = 1024 * 200


def c3_input(ntaps):
    return crand(np.random.default_rng(3000 + ntaps), C3_BLOCKS * (65537 - ntaps))


def c3_keep(ntaps):
    """Output indexes kept per block: the whole overlap region (where the previous block's tail is added, libcsdr.c:843-848), the block's end, every 61st sample between."""
    inp = 65537 - ntaps
    one = np.unique(np.concatenate([np.arange(0, ntaps + 105), np.arange(ntaps + 105, inp - 256, 61), np.arange(inp - 256, inp)]))
    return np.concatenate([b * inp + one for b in range(C3_BLOCKS)])


def c4_input():
    return crand(np.random.default_rng(4), C4_BLOCKS * 57344)


def c4_spec_keep():
    return np.arange(0, 65536, 97)


def c5_input(k):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tests_helpers import nfm_signal_u8
    return nfm_signal_u8(5000 + k, C5_SAMPLES, offset=-C5_RATES[k])


def make_baseline():
    """tests/golden/baseline_vectors.npz: outputs of the UNMODIFIED reference at the sizes BASELINE.json names.
    C3: apply_fir_fft_cc at fft 65536 with 1023 / 4095 taps (libcsdr.c:814-849 driven like csdr.c:1846-1880), three blocks, a subset of the output samples.
    C4: fastddc at decimation 256 / transition_bw 0.001 (fft 65536, taps 8193, fft_inv 512; fastddc.c:38-166, csdr.c:2255-2378): every 97th bin of five forward spectra,
        the complete output of three channels over five blocks.
    C5: README.md:87 as eight processes of the reference binary at three shift rates; the audio every complete run agrees on."""
    assert oracle.Ref.available(), "oracle/_ref/libcsdr_ref.so is missing (needs /root/reference; run `make -C oracle`)"
    R = oracle.ref()
    g = {}
    for nt in C3_TAPS:
        taps = R.firdes_bandpass_c(nt, -0.1, 0.2)
        y = R.bandpass_fir_fft_cc(c3_input(nt), taps, 65536)
        assert y.size == C3_BLOCKS * (65537 - nt)
        g["c3_taps_%d" % nt] = taps
        g["c3_out_%d" % nt] = y[c3_keep(nt)].copy()
    x = c4_input()
    d0, err = R.fastddc_init(0.001, 256, 0.0)
    assert err == 0 and d0.fft_size == 65536 and d0.input_size == 57344 and d0.fft_inv_size == 512
    spec = R.fastddc_fwd_cc(x, d0)
    g["c4_spec_subset"] = spec[:, c4_spec_keep()].copy()
    for c in C4_CHANNELS:
        rate = float(np.float32(-0.5 + (c + 0.5) / 256))
        dc, err = R.fastddc_init(0.001, 256, rate)
        assert err == 0
        g["c4_out_ch%d" % c] = R.fastddc_inv_cc(spec, dc, R.fastddc_taps_fft(dc, rate, 256))
        g["c4_geometry_ch%d" % c] = np.array([dc.pre_decimation, dc.post_decimation, dc.taps_length, dc.overlap_length, dc.fft_size, dc.fft_inv_size, dc.input_size,
                                               dc.post_input_size, dc.startbin, dc.offsetbin, dc.scrap], np.int64)
    cli = os.path.join(ROOT, "oracle", "_ref", "csdr")
    keep = ((C5_SAMPLES - 801) // 50 + 1 + 1024 - 201) // 1024 * 1024
    for k, rate in enumerate(C5_RATES):
        cmds = ("convert_u8_f", "shift_addition_cc %r" % rate, "fir_decimate_cc 50 0.005 HAMMING", "fmdemod_quadri_cf", "limit_ff", "deemphasis_nfm_ff 48000", "fastagc_ff", "convert_f_s16")
        pipe = " | ".join("%s %s" % (cli, c) for c in cmds)
        out = subprocess.run(pipe, shell=True, input=c5_input(k).tobytes(), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=120).stdout
        s16 = np.frombuffer(out[:len(out) // 2 * 2], np.int16)
        assert s16.size >= keep
        g["c5_s16_rate%d" % k] = s16[:keep].copy()
    path = os.path.join(ROOT, "tests", "golden", "baseline_vectors.npz")
    np.savez_compressed(path, **g)
    print("wrote %s: %d arrays, %.0f KB" % (path, len(g), os.path.getsize(path) / 1024))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "nfm":
        return make_nfm_cli()
    if len(sys.argv) > 1 and sys.argv[1] == "baseline":
        return make_baseline()
    assert oracle.Ref.available(), "oracle/_ref/libcsdr_ref.so is missing (needs /root/reference; run `make -C oracle`)"
    R = oracle.ref()
    rng = np.random.default_rng(20260924)
    g = {}
    # ---- converters (bit exact)
    g["u8"] = np.arange(256, dtype=np.uint8)
    g["convert_u8_f"] = R.convert_u8_f(g["u8"])
    g["s8"] = np.arange(-128, 128, dtype=np.int8)
    g["convert_s8_f"] = R.convert_s8_f(g["s8"])
    g["s16"] = rng.integers(-32768, 32768, 4096).astype(np.int16)
    g["convert_s16_f"] = R.convert_s16_f(g["s16"])
    fl = np.concatenate([np.linspace(-1.2, 1.2, 4001), [0.0, -0.0, 1.0, -1.0, 0.999999, 1e-30, 3.0, -3.0]]).astype(f32)
    g["flt"] = fl
    for name in ("convert_f_u8", "convert_f_s8", "convert_f_s16"):
        g[name] = getattr(R, name)(fl)
    g["convert_f_s24_le"] = R.convert_f_s24(fl, 0); g["convert_f_s24_be"] = R.convert_f_s24(fl, 1)
    g["s24"] = rng.integers(0, 256, 3 * 1000, dtype=np.uint8)
    g["convert_s24_f_le"] = R.convert_s24_f(g["s24"], 0); g["convert_s24_f_be"] = R.convert_s24_f(g["s24"], 1)
    # ---- design
    g["firdes_lowpass_79"] = R.firdes_lowpass_f(79, 0.05)
    g["firdes_lowpass_801_blackman"] = R.firdes_lowpass_f(801, 0.01, "BLACKMAN")
    g["firdes_bandpass_255"] = R.firdes_bandpass_c(255, -0.1, 0.2)
    # ---- shifters with the CLI's 1024-chunk framing
    x = crand(rng, 5 * 1024 + 300)
    g["cx"] = x
    for name in ("shift_addition_cc", "shift_math_cc", "shift_addfast_cc", "shift_unroll_cc"):
        y, ph = getattr(R, name)(x, -0.085)
        g[name] = y; g[name + "_phase"] = np.float32(ph)
    y, ph = R.shift_table_cc(x, -0.085, 4096); g["shift_table_cc"] = y
    g["rx"] = rng.uniform(-1, 1, 3000).astype(f32)
    g["shift_addition_fc"] = R.shift_addition_fc(g["rx"], 0.21)[0]
    y, st = R.decimating_shift_addition_cc(x, 0.07, 6); g["decimating_shift_addition_cc"] = y; g["dsa_status"] = np.array(st, np.float64)
    # ---- FIR / demod / audio
    taps = R.firdes_lowpass_f(79, 0.05)
    g["fir_decimate_cc"] = R.fir_decimate_cc_block(x, 10, taps)
    y, last = R.fmdemod_quadri_cf(x, (0.3, -0.2)); g["fmdemod_quadri_cf"] = y
    y, last = R.deemphasis_wfm_ff(g["rx"], 50e-6, 48000, 0.1); g["deemphasis_wfm_ff"] = y
    g["deemphasis_nfm_ff_48000"] = R.deemphasis_nfm_ff(g["rx"], 48000)
    g["limit_ff"] = R.limit_ff((g["rx"] * 1.7).astype(f32), 1.0)
    g["gain_ff"] = R.gain_ff(g["rx"], 2.5)
    g["agc_in"] = (rng.uniform(-1, 1, 6 * 1024) * np.repeat(rng.uniform(0.01, 1, 6 * 8), 128)).astype(f32)
    g["fastagc_ff"] = R.fastagc_ff(g["agc_in"], 1024, 0.8)
    g["fractional_decimator_ff_5"] = R.fractional_decimator_ff(g["rx"], 5.0)
    g["fractional_decimator_ff_2p5_4"] = R.fractional_decimator_ff(g["rx"], 2.5, 4)
    # ---- FFT paths
    bt = R.firdes_bandpass_c(255, -0.1, 0.2)
    g["bandpass_fir_fft_cc_1024"] = R.bandpass_fir_fft_cc(x, bt, 1024)
    ddc, err = R.fastddc_init(0.05, 16, 0.11)
    g["fastddc_geometry"] = np.array([ddc.pre_decimation, ddc.post_decimation, ddc.taps_length, ddc.overlap_length, ddc.fft_size, ddc.fft_inv_size,
                                      ddc.input_size, ddc.post_input_size, ddc.startbin, ddc.offsetbin, ddc.scrap], np.int64)
    xd = crand(rng, 5 * ddc.input_size)
    g["ddc_x"] = xd
    spec = R.fastddc_fwd_cc(xd, ddc); g["fastddc_fwd_cc"] = spec
    g["fastddc_inv_cc"] = R.fastddc_inv_cc(spec, ddc, R.fastddc_taps_fft(ddc, 0.11, 16))
    # ---- the README.md:66 chain, stage by stage through the reference library with the CLI's framing
    n = 60 * 1024
    t = np.arange(n)
    msg = np.sin(2 * np.pi * 1e3 / 2.4e6 * t) + 0.3 * rng.uniform(-1, 1, n)
    sig = 0.7 * np.exp(1j * (2 * np.pi * np.cumsum(0.03125 * msg) + 2 * np.pi * 0.085 * t)) + 0.01 * (rng.normal(size=n) + 1j * rng.normal(size=n))
    iq = np.empty(2 * n, f32); iq[0::2] = sig.real; iq[1::2] = sig.imag
    iq = np.clip(np.round(127.5 * (iq + 1)), 0, 255).astype(np.uint8)
    g["wfm_iq_u8"] = iq
    s16, af = R.wfm_chain(iq, -0.085, 10, taps)
    g["wfm_audio_f"] = af; g["wfm_audio_s16"] = s16
    # ---- f2 blocks
    g["amdemod_cf"] = R.amdemod_cf(x); g["amdemod_estimator_cf"] = R.amdemod_estimator_cf(x)
    g["fmdemod_atan_cf"] = R.fmdemod_atan_cf(x, 0.3)[0]
    g["logpower_cf"] = R.logpower_cf(x + c64(0.01), 3.0)
    g["dc_in"] = (g["rx"] + 0.3).astype(f32)
    g["dcblock_ff"] = R.dcblock_ff(g["dc_in"], 0, (0.1, 0.2))[0]
    g["fastdcblock_ff"] = R.fastdcblock_ff(g["dc_in"], 1024, 0.1)[0]
    g["agc_ff"] = R.agc_ff(g["agc_in"], 1024)[0]
    g["window_hamming_512"] = R.precalculate_window(512, "HAMMING")
    # ---- f3: ADPCM (bit exact); the waterfall compressor only exists as a CLI loop
    pcm = (8000 * np.sin(np.arange(4001) * 0.01) + rng.integers(-3000, 3000, 4001)).astype(np.int16)
    g["pcm"] = pcm
    enc, st = R.encode_ima_adpcm_i16_u8(pcm, (5, -100)); g["adpcm_enc"] = enc; g["adpcm_enc_state"] = np.array(st, np.int64)
    dec, st = R.decode_ima_adpcm_u8_i16(enc, (3, 77)); g["adpcm_dec"] = dec; g["adpcm_dec_state"] = np.array(st, np.int64)
    rows = (rng.uniform(-120, 10, 3 * 256) + 20 * np.sin(np.arange(3 * 256) * 0.05)).astype(f32)
    g["fft_rows_db"] = rows
    cli = os.path.join(ROOT, "oracle", "_ref", "csdr")
    out = subprocess.run([cli, "compress_fft_adpcm_f_u8", "256"], input=rows.tobytes(), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=30).stdout
    g["compress_fft_adpcm_f_u8_256"] = np.frombuffer(out, np.uint8)[:3 * 133].copy()     # the reference repeats its last block at EOF
    path = os.path.join(ROOT, "tests", "golden", "ref_vectors.npz")
    np.savez_compressed(path, **g)
    print("wrote %s: %d arrays, %.0f KB" % (path, len(g), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
