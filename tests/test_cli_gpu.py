"""`csdr <function>` front end (csdr_amd/csdr, built from csdr_amd/csrc/csdr_cli.cpp) on the GPU: raw streams in and out through real
pipes, compared with the oracle's stream models of the reference CLI loops (csdr.c, cited per test) and -- when the compiled reference
CLI travelled with the snapshot (oracle/_ref/csdr) -- with the reference processes themselves in the README.md:66 pipeline.
Small CSDR_AMD_BLOCK values force several host iterations so that every carry path (refeed, overlap, phase, AGC history) is crossed."""
import os
import subprocess

import numpy as np
import pytest
from oracle import relrms

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "csdr_amd", "csdr")
REF_CLI = os.path.join(ROOT, "oracle", "_ref", "csdr")
c64 = np.complex64
f32 = np.float32
TOL = 1e-5


def run(args, data, block=8192, cli=CLI):
    assert os.path.exists(cli), "csdr_amd/csdr not built (make -C csdr_amd/csrc)"
    env = dict(os.environ, CSDR_AMD_BLOCK=str(block))
    p = subprocess.run([cli] + [str(a) for a in args], input=np.ascontiguousarray(data).tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout


def crand(rng, n):
    return (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(c64)


def test_cli_converters_bit_exact(port):
    rng = np.random.default_rng(1)
    u8 = rng.integers(0, 256, 50001, dtype=np.uint8)
    assert run(["convert_u8_f"], u8) == port.convert_u8_f(u8).tobytes()
    x = rng.uniform(-1.2, 1.2, 30011).astype(f32)
    assert run(["convert_f_s16"], x) == port.convert_f_s16(x).tobytes()
    assert run(["convert_f_i16"], x) == port.convert_f_s16(x).tobytes()
    assert run(["convert_f_u8"], x) == port.convert_f_u8(x).tobytes()
    assert run(["convert_f_s8"], x) == port.convert_f_s8(x).tobytes()
    s16 = rng.integers(-32768, 32768, 20001, dtype=np.int16)
    assert run(["convert_s16_f"], s16) == port.convert_s16_f(s16).tobytes()
    s8 = rng.integers(-128, 128, 20001, dtype=np.int8)
    assert run(["convert_s8_f"], s8) == port.convert_s8_f(s8).tobytes()
    assert run(["convert_f_s24"], x) == port.convert_f_s24(x).tobytes()
    assert run(["convert_f_s24", "--bigendian"], x) == port.convert_f_s24(x, 1).tobytes()
    s24 = rng.integers(0, 256, 3 * 10001, dtype=np.uint8)
    assert run(["convert_s24_f"], s24) == port.convert_s24_f(s24).tobytes()


@pytest.mark.parametrize("cmd,fn,extra", [("shift_addition_cc", "shift_addition_cc", []), ("shift_math_cc", "shift_math_cc", []),
                                          ("shift_addfast_cc", "shift_addfast_cc", []), ("shift_unroll_cc", "shift_unroll_cc", []),
                                          ("shift_table_cc", "shift_table_cc", [4096])])
def test_cli_shift(port, cmd, fn, extra):
    rng = np.random.default_rng(2)
    x = crand(rng, 5 * 8192 + 1024 + 300)                  # several host blocks, a whole chunk and a ragged tail
    got = np.frombuffer(run([cmd, -0.085] + extra, x), c64)
    want = getattr(port, fn)(x, -0.085, *extra)[0]
    assert got.size == x.size
    assert relrms(got, want) <= TOL


def test_cli_shift_addition_fc(port):
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, 3 * 8192 + 77).astype(f32)
    got = np.frombuffer(run(["shift_addition_fc", 0.21], x), c64)
    assert relrms(got, port.shift_addition_fc(x, 0.21)[0]) <= TOL


def test_cli_fir_decimate_refeed(port):
    """csdr.c:1114-1177: every output of the stream appears exactly once, whatever the block size."""
    rng = np.random.default_rng(4)
    x = crand(rng, 100000)
    taps = port.firdes_lowpass_f(port.firdes_filter_len(0.05), 0.5 / 10)
    want = port.fir_decimate_cc(x, 10, taps)
    for block in (4096, 8192, 65536):
        got = np.frombuffer(run(["fir_decimate_cc", 10, 0.05, "HAMMING"], x, block), c64)
        assert got.size == want.size
        assert relrms(got, want) <= TOL


def test_cli_long_streams_small_blocks(port):
    """Millions of samples through operators that leave an unconsumed tail every pass (csdr.c:1114-1177, 1511-1524), input arriving in full blocks: the
    carry in front of the first operator must stay bounded (round 2 took a new buffer every pass and died with 'block too small' after ~225 k samples at
    block 4096 / 81 taps, ~5 M samples at block 65536 / 801 taps)."""
    rng = np.random.default_rng(44)
    x = crand(rng, 2_500_000)
    taps = port.firdes_lowpass_f(port.firdes_filter_len(0.05), 0.5 / 10)
    got = np.frombuffer(run(["fir_decimate_cc", 10, 0.05, "HAMMING"], x, 4096), c64)
    want = port.fir_decimate_cc(x, 10, taps)
    assert got.size == want.size and relrms(got, want) <= TOL
    x2 = crand(rng, 6_000_000)
    taps50 = port.firdes_lowpass_f(port.firdes_filter_len(0.005), 0.5 / 50)
    got = np.frombuffer(run(["fir_decimate_cc", 50, 0.005, "HAMMING"], x2, 65536), c64)
    want = port.fir_decimate_cc(x2, 50, taps50)
    assert got.size == want.size and relrms(got, want) <= TOL
    a = rng.uniform(-1, 1, 2_200_000).astype(f32)
    got = np.frombuffer(run(["fractional_decimator_ff", 5], a, 4096), f32)
    want = port.fractional_decimator_ff(a, 5.0)
    m = min(got.size, want.size)
    assert abs(got.size - want.size) <= 1 and relrms(got[:m], want[:m]) <= TOL


def test_cli_fm_audio_stages(port):
    rng = np.random.default_rng(5)
    x = crand(rng, 40000)
    got = np.frombuffer(run(["fmdemod_quadri_cf"], x), f32)
    assert relrms(got, port.fmdemod_quadri_cf(x)[0]) <= TOL
    a = rng.uniform(-1.5, 1.5, 50000).astype(f32)
    assert run(["limit_ff"], a) == port.limit_ff(a, 1.0).tobytes()
    assert run(["limit_ff", 0.5], a) == port.limit_ff(a, 0.5).tobytes()
    got = np.frombuffer(run(["deemphasis_wfm_ff", 48000, 50e-6], a), f32)
    assert relrms(got, port.deemphasis_wfm_ff(a, 50e-6, 48000)[0]) <= TOL
    got = np.frombuffer(run(["fractional_decimator_ff", 5], a), f32)
    want = port.fractional_decimator_ff(a, 5.0)
    assert abs(got.size - want.size) <= 1
    m = min(got.size, want.size)
    assert relrms(got[:m], want[:m]) <= TOL
    got = np.frombuffer(run(["fractional_decimator_ff", 2.5, 4], a), f32)
    want = port.fractional_decimator_ff(a, 2.5, 4, bufsize=1024)
    m = min(got.size, want.size)
    assert m >= want.size - 2 and relrms(got[:m], want[:m]) <= TOL
    got = np.frombuffer(run(["fractional_decimator_ff", 3.3], a), f32)      # a rate that is not exact in float: the window loop of csdr.c:1511-1524 matters
    want = port.fractional_decimator_ff(a, 3.3, bufsize=1024)
    m = min(got.size, want.size)
    assert m >= want.size - 2 and relrms(got[:m], want[:m]) <= TOL
    if os.path.exists(REF_CLI):
        ref = np.frombuffer(run(["fractional_decimator_ff", 3.3], a, cli=REF_CLI), f32)
        m = min(ref.size, got.size)
        assert m >= got.size - 1024 and relrms(got[:m], ref[:m]) <= TOL
    got = np.frombuffer(run(["fastagc_ff", 1024, 0.8], a), f32)
    want = port.fastagc_ff(a, 1024, 0.8)
    assert got.size == want.size and relrms(got, want) <= TOL


def test_cli_deemphasis_nfm(port):
    rng = np.random.default_rng(6)
    a = rng.uniform(-1, 1, 30000).astype(f32)
    taps = np.load(os.path.join(ROOT, "tests", "golden", "nfm_deemph_taps.npz"))["sr48000"]
    got = np.frombuffer(run(["deemphasis_nfm_ff", 48000], a), f32)
    want = port.deemphasis_nfm_ff_cli(a, taps)             # the CLI loop filters  the_bufsize zeros ++ stream  (csdr.c:1076-1081)
    assert got.size == want.size and relrms(got, want) <= TOL
    if os.path.exists(REF_CLI):                             # ... as the reference binary itself does (it may repeat its last block at EOF)
        ref = np.frombuffer(run(["deemphasis_nfm_ff", 48000], a, cli=REF_CLI), f32)
        m = min(ref.size, got.size)
        assert m >= got.size - 1024 and relrms(got[:m], ref[:m]) <= TOL
    p = subprocess.run([CLI, "deemphasis_nfm_ff", "12345"], input=b"", stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode != 0 and b"sample rate" in p.stderr


def test_cli_bandpass_fir_fft(port):
    """csdr.c:1810-1886 with the CLI's own fft_size rule (next_pow2(taps), doubled when the padding is < 200)."""
    rng = np.random.default_rng(7)
    x = crand(rng, 60000)
    nt = port.firdes_filter_len(0.02)
    fft = port.next_pow2(nt)
    if fft - nt < 200:
        fft *= 2
    taps = port.firdes_bandpass_c(nt, -0.1, 0.2)
    want = port.bandpass_fir_fft_cc(x, taps, fft)
    got = np.frombuffer(run(["bandpass_fir_fft_cc", -0.1, 0.2, 0.02], x, 4096), c64)
    assert got.size == want.size and relrms(got, want) <= TOL


def test_cli_fastddc_pipe(port):
    """`fastddc_fwd_cc D tbw | fastddc_inv_cc shift D tbw` (csdr.c:2255-2378) as two processes joined by a pipe."""
    rng = np.random.default_rng(8)
    D, tbw, shift = 16, 0.02, 0.11
    ddc, err = port.fastddc_init(tbw, D, shift)
    assert err == 0
    x = crand(rng, 9 * ddc.input_size + 100)
    spectra = port.fastddc_fwd_cc(x, ddc)
    want = port.fastddc_inv_cc(spectra, ddc, port.fastddc_taps_fft(ddc, shift, D))
    mid = run(["fastddc_fwd_cc", D, tbw], x, 4096)
    assert relrms(np.frombuffer(mid, c64), spectra.ravel()) <= TOL
    got = np.frombuffer(run(["fastddc_inv_cc", shift, D, tbw], np.frombuffer(mid, np.uint8), 4096), c64)
    assert got.size == want.size and relrms(got, want) <= TOL


def fm_iq(rng, n):
    """u8 IQ of an FM signal 0.085 fs above centre (the WFM chain shifts it down): well conditioned for the demodulator, unlike noise."""
    t = np.arange(n)
    msg = np.sin(2 * np.pi * 1e3 / 2.4e6 * t) + 0.3 * rng.uniform(-1, 1, n)
    sig = 0.7 * np.exp(1j * (2 * np.pi * np.cumsum(0.03125 * msg) + 2 * np.pi * 0.085 * t)) + 0.01 * (rng.normal(size=n) + 1j * rng.normal(size=n))
    iq = np.empty(2 * n, f32); iq[0::2] = sig.real; iq[1::2] = sig.imag
    return np.clip(np.round(127.5 * (iq + 1)), 0, 255).astype(np.uint8)


def wfm_pipeline(cli, iq, block=None, env_extra=None, stderr_dir=None):
    stages = [["convert_u8_f"], ["shift_addition_cc", "-0.085"], ["fir_decimate_cc", "10", "0.05", "HAMMING"], ["fmdemod_quadri_cf"],
              ["fractional_decimator_ff", "5"], ["deemphasis_wfm_ff", "48000", "50e-6"], ["convert_f_s16"]]
    env = dict(os.environ)
    if block:
        env["CSDR_AMD_BLOCK"] = str(block)
    env.update(env_extra or {})
    procs = []
    prev = subprocess.PIPE
    for i, st in enumerate(stages):
        err = open(os.path.join(stderr_dir, "err%d.txt" % i), "wb") if stderr_dir else subprocess.DEVNULL
        p = subprocess.Popen([cli] + st, stdin=prev if i else subprocess.PIPE, stdout=subprocess.PIPE, stderr=err, env=env)
        if i:
            procs[-1].stdout.close()
        prev = p.stdout
        procs.append(p)
    import threading
    t = threading.Thread(target=lambda: (procs[0].stdin.write(iq.tobytes()), procs[0].stdin.close()))
    t.start()
    out = procs[-1].stdout.read()
    t.join()
    for p in procs:
        p.wait(timeout=60)
    return np.frombuffer(out, np.int16)


def test_cli_device_handoff_between_processes(port, tmp_path):
    """north_star: "the stdin->stdout pipe never round-trips to host between stages".  The unchanged seven-process shell pipeline of README.md:66: adjacent
    csdr processes find each other through a socket named after the pipe between them and hand blocks over in HBM (HIP IPC ring + tokens; csdr_cli.cpp "device
    hand-off"); the first process still reads bytes from a foreign producer, the last still writes bytes.  Same samples as with CSDR_AMD_IPC=0, bit for bit, and
    as the oracle's chain; every inner link reports the hand-off (or, where HIP IPC is not available to the container, its refusal and the byte fallback)."""
    iq = fm_iq(np.random.default_rng(19), 1200000)
    want_s16, _ = port.wfm_chain(iq, -0.085, 10, port.firdes_lowpass_f(port.firdes_filter_len(0.05), 0.05))
    d1 = tmp_path / "on"; d1.mkdir()
    on = wfm_pipeline(CLI, iq, 65536, {"CSDR_AMD_IPC_VERBOSE": "1", "CSDR_AMD_IPC_WAIT_MS": "3000"}, str(d1))     # (a loaded box may start a consumer late: look for its socket for 3 s)
    off = wfm_pipeline(CLI, iq, 65536, {"CSDR_AMD_IPC": "0"})
    assert np.array_equal(on, off)
    m = min(on.size, want_s16.size)
    assert m >= want_s16.size - 2
    d = np.abs(on[:m].astype(np.int32) - want_s16[:m].astype(np.int32))
    assert d.max() <= 1 and np.mean(d != 0) < 0.01
    errs = [open(d1 / ("err%d.txt" % i)).read() for i in range(7)]
    handed = [("output leaves by device hand-off" in errs[i], "input arrives by device hand-off" in errs[i + 1]) for i in range(6)]
    refused = ["refused" in e for e in errs]
    assert all(a == b for a, b in handed), errs
    assert all(a for a, _ in handed) or any(refused), errs         # every link hands over in HBM, or says why not
    assert "hand-off" not in errs[0].split("output leaves")[0] and "output leaves" not in errs[6]      # the ends of the pipeline talk bytes
    if not all(a for a, _ in handed):
        pytest.xfail("HIP IPC refused on this box (the byte fallback gave the same samples): " + " | ".join(e.strip() for e in errs if "refused" in e))


def test_cli_device_handoff_refused_falls_back_to_bytes(tmp_path):
    """The other half of the negotiation: the producer connects, the consumer cannot take the handle (another device, HIP IPC not available to it: forced here with
    CSDR_AMD_IPC_TEST_NAK) and answers NAK -- both sides go on with bytes through the pipe they already share; same samples as without any hand-off."""
    iq = fm_iq(np.random.default_rng(20), 300000)
    d1 = tmp_path / "nak"; d1.mkdir()
    nak = wfm_pipeline(CLI, iq, 65536, {"CSDR_AMD_IPC_VERBOSE": "1", "CSDR_AMD_IPC_WAIT_MS": "3000", "CSDR_AMD_IPC_TEST_NAK": "1"}, str(d1))
    off = wfm_pipeline(CLI, iq, 65536, {"CSDR_AMD_IPC": "0"})
    assert nak.size > 5000 and np.array_equal(nak, off)
    errs = [open(d1 / ("err%d.txt" % i)).read() for i in range(7)]
    assert all("hand-off to the next process refused" in errs[i] and "hand-off from the previous process refused" in errs[i + 1] for i in range(6)), errs


def test_cli_device_handoff_consumer_dies_midstream():
    """`csdr a | csdr b` with device hand-off, b killed in the middle of an endless stream while all ring slots are in flight -- a byte pipe would
    end `a` with SIGPIPE; the hand-off's credit socket must do the same (EOF on the credit side BEFORE the producer's own shutdown = the consumer is gone) instead of
    leaving `a` blocked on a free slot forever."""
    import signal
    import threading
    import time
    env = dict(os.environ, CSDR_AMD_BLOCK="65536", CSDR_AMD_IPC_WAIT_MS="3000", CSDR_AMD_IPC_VERBOSE="1")
    a = subprocess.Popen([CLI, "convert_u8_f"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    b = subprocess.Popen([CLI, "shift_addition_cc", "0.1"], stdin=a.stdout, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env)
    a.stdout.close()
    chunk = np.random.default_rng(31).integers(0, 256, 1 << 20, dtype=np.uint8).tobytes()

    def feed():
        try:
            while True:
                a.stdin.write(chunk)
        except (BrokenPipeError, OSError, ValueError):
            pass
    t = threading.Thread(target=feed, daemon=True); t.start()
    got = 0
    while got < (8 << 20):                                           # the stream is flowing: 8 MB of output seen
        d = b.stdout.read(1 << 20)
        assert d, "consumer ended early"
        got += len(d)
    b.send_signal(signal.SIGKILL); b.wait(timeout=10)
    try:
        a.wait(timeout=20)                                           # (before the fix: never)
    except subprocess.TimeoutExpired:
        a.kill(); a.wait()
        raise AssertionError("the producer stayed blocked after its hand-off consumer was killed: " + a.stderr.read().decode()[-400:])
    t.join(timeout=10)
    assert not t.is_alive()


def test_cli_handoff_then_bytes_from_a_second_writer(port, tmp_path):
    """`(csdr a < f1; csdr a < f2) | csdr b` -- the first producer hands over in HBM, the second finds no listener any more and writes bytes into the
    same pipe: b must carry on in byte mode after the hand-off socket's EOF until stdin itself ends (the reference's b sees one stream, csdr.c:232-247)."""
    rng = np.random.default_rng(32)
    x1 = rng.integers(0, 256, 300000, dtype=np.uint8); x2 = rng.integers(0, 256, 200000, dtype=np.uint8)
    f1 = tmp_path / "f1.u8"; f2 = tmp_path / "f2.u8"; x1.tofile(f1); x2.tofile(f2)
    env = dict(os.environ, CSDR_AMD_BLOCK="65536", CSDR_AMD_IPC_WAIT_MS="3000")
    cmd = "(%s convert_u8_f < %s; %s convert_u8_f < %s) | %s convert_f_u8" % (CLI, f1, CLI, f2, CLI)
    p = subprocess.run(["bash", "-c", cmd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
    assert p.returncode == 0, p.stderr.decode()
    want = port.convert_f_u8(port.convert_u8_f(np.concatenate([x1, x2])))
    got = np.frombuffer(p.stdout, np.uint8)
    assert got.size == want.size and np.array_equal(got, want)


def test_cli_regular_file_input_with_parallel_readers(port, tmp_path):
    """`csdr ... < file`: the block's bytes come from CSDR_AMD_READERS threads with pread() on disjoint slices (one reader thread copied 7 GS/s out of the page cache,
    a third of `cat`: VERDICT r4 #7).  A file whose length is no multiple of the block, the slice or the page; blocks of 4 Mi and of 300 k elements (the second below
    the 4 MiB threshold: plain read() on the same descriptor); 1 / 3 / 4 readers: the same bytes out, and the oracle's."""
    rng = np.random.default_rng(33)
    x = rng.integers(0, 256, 9 * (1 << 20) + 12345, dtype=np.uint8)
    f = tmp_path / "in.u8"; x.tofile(f)
    want = port.convert_u8_f(x)
    for block in (4 << 20, 300000):
        outs = []
        for readers in ("1", "3", "4"):
            env = dict(os.environ, CSDR_AMD_BLOCK=str(block), CSDR_AMD_READERS=readers)
            with open(f, "rb") as fin:
                p = subprocess.run([CLI, "convert_u8_f"], stdin=fin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
            assert p.returncode == 0, p.stderr.decode()
            outs.append(np.frombuffer(p.stdout, f32))
        for o in outs:
            assert o.size == want.size and np.array_equal(o.view(np.uint32), want.view(np.uint32)), (block, o.size, want.size)


def test_cli_wfm_shell_pipeline(port):
    """README.md:66 as seven processes and as the fused `wfm_chain_u8_s16`, against the oracle chain and the reference's own CLI."""
    iq = fm_iq(np.random.default_rng(9), 480000)
    want_s16, _ = port.wfm_chain(iq, -0.085, 10, port.firdes_lowpass_f(port.firdes_filter_len(0.05), 0.05))
    piped = wfm_pipeline(CLI, iq, 16384)
    fused = np.frombuffer(run(["wfm_chain_u8_s16", -0.085], iq, 65536), np.int16)
    for got in (piped, fused):
        m = min(got.size, want_s16.size)
        assert m >= want_s16.size - 2
        d = np.abs(got[:m].astype(np.int32) - want_s16[:m].astype(np.int32))
        assert d.max() <= 1 and np.mean(d != 0) < 0.01            # float path then a truncating s16 cast: isolated +-1 LSB at most
    if os.path.exists(REF_CLI):
        ref = wfm_pipeline(REF_CLI, iq)
        m = min(ref.size, piped.size, want_s16.size)               # the reference CLI's EOF handling differs in the tail
        assert m > 9000
        d = np.abs(ref[:m].astype(np.int32) - piped[:m].astype(np.int32))
        assert d.max() <= 1 and np.mean(d != 0) < 0.01


def test_cli_chain_commands_at_large_blocks(port):
    """From 1 Mi-sample blocks on (the default is 4 Mi) the single-stream chain commands run the rate-per-stream objects (one stream as 16 time segments per tile,
    chunk seeds from a host-side phase chain): `wfm_chain_u8_s16` and `nfm_chain_u8_s16` over 2.6 M samples in blocks of 1 Mi (two full blocks and a ragged one,
    state and seed tables carried) against the oracle chains; CSDR_AMD_CLI_SHARED=1 (the shared-rate object) must give the same samples within the same gate."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tests_helpers import nfm_signal_u8
    n = 2 * 1048576 + 500000 + 333
    iq = fm_iq(np.random.default_rng(19), n)
    want, _ = port.wfm_chain(iq, -0.085, 10, port.firdes_lowpass_f(port.firdes_filter_len(0.05), 0.05))
    for extra in ({}, {"CSDR_AMD_CLI_SHARED": "1"}):
        env = dict(os.environ, CSDR_AMD_BLOCK=str(1048576), **extra)
        p = subprocess.run([CLI, "wfm_chain_u8_s16", "-0.085"], input=iq.tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
        assert p.returncode == 0, p.stderr.decode()
        got = np.frombuffer(p.stdout, np.int16)
        m = min(got.size, want.size)
        assert m >= want.size - 2 and got.size <= want.size + 2
        d = np.abs(got[:m].astype(np.int32) - want[:m].astype(np.int32))
        assert d.max() <= 1 and np.mean(d != 0) < 0.01, (extra, int(d.max()))
    iqn = nfm_signal_u8(78, n, offset=-0.11)
    dtaps = np.load(os.path.join(ROOT, "tests", "golden", "nfm_deemph_taps.npz"))["sr48000"]
    wantn, _ = port.nfm_chain(iqn, 0.11, dtaps)
    gotn = np.frombuffer(run(["nfm_chain_u8_s16", 0.11], iqn, 1048576), np.int16)
    assert gotn.size == wantn.size
    dn = np.abs(gotn.astype(np.int32) - wantn.astype(np.int32))
    assert dn.max() <= 1 and np.mean(dn != 0) < 0.01


_DEFAULT_BLOCK_DATA = {}


def _default_block_inputs():
    """One ragged 5.25 M-element stream per input type (what tools/probes/cli_default_block_sweep.py feeds), built once per session."""
    if not _DEFAULT_BLOCK_DATA:
        rng = np.random.default_rng(5)
        n = 5 * 1048576 + 12345 + 3
        t = np.arange(n)
        sig = 0.6 * np.exp(1j * (2 * np.pi * 0.085 * t + 3 * np.sin(2 * np.pi * 1e-3 * t))) + 0.02 * (rng.normal(size=n) + 1j * rng.normal(size=n))
        iq = np.empty(2 * n, f32); iq[0::2] = sig.real; iq[1::2] = sig.imag
        _DEFAULT_BLOCK_DATA["cf"] = sig.astype(c64)
        _DEFAULT_BLOCK_DATA["u8"] = np.clip(np.round(127.5 * (iq + 1)), 0, 255).astype(np.uint8)
        _DEFAULT_BLOCK_DATA["fl"] = (0.5 * np.sin(2 * np.pi * 1e-3 * t) + 0.1 * rng.uniform(-1, 1, n)).astype(f32)
        _DEFAULT_BLOCK_DATA["fm"] = fm_iq(np.random.default_rng(23), n)
        import sys
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from tests_helpers import nfm_signal_u8
        _DEFAULT_BLOCK_DATA["nfm"] = nfm_signal_u8(79, n, offset=-0.11)
    return _DEFAULT_BLOCK_DATA


def _nfm_dtaps():
    return np.load(os.path.join(ROOT, "tests", "golden", "nfm_deemph_taps.npz"))["sr48000"]


def _bpf_want(port, x):
    nt = port.firdes_filter_len(0.05)
    fft = port.next_pow2(nt)
    if fft - nt < 200:
        fft *= 2
    return port.bandpass_fir_fft_cc(x, port.firdes_bandpass_c(nt, -0.1, 0.1), fft)


WFM_CHAIN = "convert_u8_f | shift_addition_cc -0.085 | fir_decimate_cc 10 0.05 HAMMING | fmdemod_quadri_cf | fractional_decimator_ff 5 | deemphasis_wfm_ff 48000 50e-6 | convert_f_s16"
# (argv, input key, output dtype, oracle stream model, gate): "bits" = identical bytes, "rms" = equal length and rel. RMS <= 1e-5, "lsb" = s16 after a float chain (+-1 LSB,
# < 1 % of the samples), "lsb2" = the same with the fused WFM chain's early-emission rule (up to two audio samples more than the reference's count)
DEFAULT_BLOCK_CASES = [
    (["convert_u8_f"], "u8", f32, lambda o, x: o.convert_u8_f(x), "bits"),
    (["wfm_chain_u8_s16", "-0.085"], "fm", np.int16, lambda o, x: o.wfm_chain(x, -0.085, 10, o.firdes_lowpass_f(o.firdes_filter_len(0.05), 0.05))[0], "lsb2"),
    (["nfm_chain_u8_s16", "0.11"], "nfm", np.int16, lambda o, x: o.nfm_chain(x, 0.11, _nfm_dtaps())[0], "lsb"),
    (["ddc_u8_cc", "0.11", "50", "0.005", "HAMMING"], "u8", c64,
     lambda o, x: o.fir_decimate_cc(o.shift_addition_cc(o.convert_u8_f(x).view(c64), 0.11)[0], 50, o.firdes_lowpass_f(o.firdes_filter_len(0.005), 0.5 / 50)), "rms"),
    (["shift_addition_cc", "0.1"], "cf", c64, lambda o, x: o.shift_addition_cc(x, 0.1)[0], "rms"),
    (["shift_math_cc", "0.1"], "cf", c64, lambda o, x: o.shift_math_cc(x, 0.1)[0], "rms"),
    (["fir_decimate_cc", "10", "0.05", "HAMMING"], "cf", c64, lambda o, x: o.fir_decimate_cc(x, 10, o.firdes_lowpass_f(o.firdes_filter_len(0.05), 0.5 / 10)), "rms"),
    (["fir_decimate_cc", "50", "0.005", "HAMMING"], "cf", c64, lambda o, x: o.fir_decimate_cc(x, 50, o.firdes_lowpass_f(o.firdes_filter_len(0.005), 0.5 / 50)), "rms"),
    (["fmdemod_quadri_cf"], "cf", f32, lambda o, x: o.fmdemod_quadri_cf(x)[0], "rms"),
    (["bandpass_fir_fft_cc", "-0.1", "0.1", "0.05"], "cf", c64, _bpf_want, "rms"),
    (["amdemod_cf"], "cf", f32, lambda o, x: o.amdemod_cf(x), "rms"),
    (["realpart_cf"], "cf", f32, lambda o, x: o.realpart_cf(x), "bits"),
    (["fractional_decimator_ff", "5"], "fl", f32, lambda o, x: o.fractional_decimator_ff(x, 5.0), "frac"),
    (["deemphasis_wfm_ff", "48000", "50e-6"], "fl", f32, lambda o, x: o.deemphasis_wfm_ff(x, 50e-6, 48000)[0], "rms"),
    (["deemphasis_nfm_ff", "48000"], "fl", f32, lambda o, x: o.deemphasis_nfm_ff_cli(x, _nfm_dtaps()), "rms"),
    (["convert_f_s16"], "fl", np.int16, lambda o, x: o.convert_f_s16(x), "bits"),
    (["fastagc_ff"], "fl", f32, lambda o, x: o.fastagc_ff(x, 1024, 1.0), "rms"),
    (["limit_ff"], "fl", f32, lambda o, x: o.limit_ff(x, 1.0), "bits"),
    (["fastdcblock_ff"], "fl", f32, lambda o, x: o.fastdcblock_ff(x)[0], "rms"),
    (["dcblock_ff"], "fl", f32, lambda o, x: o.dcblock_ff(x)[0], "rms"),
    (["gain_ff", "0.5"], "fl", f32, lambda o, x: o.gain_ff(x, 0.5), "bits"),
    (["chain", WFM_CHAIN], "fm", np.int16, lambda o, x: o.wfm_chain(x, -0.085, 10, o.firdes_lowpass_f(o.firdes_filter_len(0.05), 0.05))[0], "lsb2"),
]


@pytest.mark.parametrize("case", DEFAULT_BLOCK_CASES, ids=[" ".join(c[0])[:40].replace(" ", "_").replace("|", "") for c in DEFAULT_BLOCK_CASES])
def test_cli_every_command_at_the_default_block(port, case):
    """The 22 hot-path commands on a ragged 5.25 M-element input at the DEFAULT block (CSDR_AMD_BLOCK unset: 4 Mi elements per read -- what a user gets; the tests above
    stream in small blocks), each against the ORACLE's stream model of the reference CLI loop: identical bytes for integer / pass-through outputs, equal length and
    rel. RMS <= 1e-5 for float outputs, +-1 LSB for s16 behind a float chain.  (tools/probes/cli_default_block_sweep.py remains as the self-consistency probe.)"""
    argv, key, ot, model, gate = case
    data = _default_block_inputs()[key]
    env = dict(os.environ); env.pop("CSDR_AMD_BLOCK", None)
    p = subprocess.run([CLI] + argv, input=data.tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
    assert p.returncode == 0, p.stderr.decode()[-500:]
    assert len(p.stdout) % np.dtype(ot).itemsize == 0
    got = np.frombuffer(p.stdout, ot)
    want = np.asarray(model(port, data))
    if gate == "bits":
        assert got.tobytes() == want.astype(ot, copy=False).tobytes()
    elif gate == "rms":
        assert got.size == want.size and want.size > 0
        assert relrms(got, want) <= TOL
    elif gate == "frac":
        m = min(got.size, want.size)
        assert abs(got.size - want.size) <= 1 and relrms(got[:m], want[:m]) <= TOL
    else:
        if gate == "lsb":
            assert got.size == want.size
        else:
            assert 0 <= got.size - want.size <= 2
        m = min(got.size, want.size)
        d = np.abs(got[:m].astype(np.int32) - want[:m].astype(np.int32))
        assert d.max() <= 1 and np.mean(d != 0) < 0.01, int(d.max())


# ---------------------------------------------------------------- f1: protocol, control channel, in-process chains
def test_cli_decimating_shift_addition(port):
    """csdr.c:851-875: one library call per the_bufsize (16384 by default) samples, status carried."""
    rng = np.random.default_rng(10)
    x = crand(rng, 3 * 16384 + 5000)
    got = np.frombuffer(run(["decimating_shift_addition_cc", 0.07, 6], x, 32768), c64)
    outs, st = [], (0, 0.0, 0)
    for at in range(0, x.size, 16384):
        y, st = port.decimating_shift_addition_cc(x[at:at + 16384], 0.07, 6, st)
        outs.append(y)
    want = np.concatenate(outs)
    assert got.size == want.size and relrms(got, want) <= TOL


def test_cli_chain_equals_pipeline(port):
    """`csdr chain "a | b | c"`: same bytes as the process-per-command pipeline, intermediates never leave HBM."""
    rng = np.random.default_rng(11)
    iq = rng.integers(0, 256, 2 * 300000, dtype=np.uint8)
    nfm = "convert_u8_f | shift_addition_cc 0.11 | fir_decimate_cc 50 0.005 HAMMING | fmdemod_quadri_cf | limit_ff | deemphasis_nfm_ff 48000 | fastagc_ff | convert_f_s16"
    got = np.frombuffer(run(["chain", nfm], iq, 65536), np.int16)
    taps = np.load(os.path.join(ROOT, "tests", "golden", "nfm_deemph_taps.npz"))["sr48000"]
    want, _ = port.nfm_chain(iq, 0.11, taps)
    assert got.size == want.size
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert d.max() <= 1 and np.mean(d != 0) < 0.01
    # the README.md:66 pattern is replaced by the fused kernel
    wfm = "csdr convert_u8_f | csdr shift_addition_cc -0.085 | csdr fir_decimate_cc 10 0.05 HAMMING | csdr fmdemod_quadri_cf | csdr fractional_decimator_ff 5 | csdr deemphasis_wfm_ff 48000 50e-6 | csdr convert_f_s16"
    env = dict(os.environ, CSDR_AMD_BLOCK="65536")
    iq = fm_iq(rng, 300000)
    p = subprocess.run([CLI, "chain", wfm], input=iq.tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
    assert p.returncode == 0 and b"fused" in p.stderr
    got = np.frombuffer(p.stdout, np.int16)
    want, _ = port.wfm_chain(iq, -0.085, 10, port.firdes_lowpass_f(port.firdes_filter_len(0.05), 0.05))
    m = min(got.size, want.size)
    assert m >= want.size - 2
    d = np.abs(got[:m].astype(np.int32) - want[:m].astype(np.int32))
    assert d.max() <= 1 and np.mean(d != 0) < 0.01


def test_cli_fused_front_end_and_nfm_commands(port):
    """Extensions: `csdr ddc_u8_cc r D tbw window` (= convert_u8_f | shift_addition_cc | fir_decimate_cc in one pass on the matrix cores) and
    `csdr nfm_chain_u8_s16 r` (README.md:87 in one process), streaming in several host blocks; a chain that merely starts with the three
    front-end commands gets them fused."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tests_helpers import nfm_signal_u8
    n = 1024 * 330 + 517
    iq = nfm_signal_u8(77, n, offset=-0.11)
    taps = port.firdes_lowpass_f(port.firdes_filter_len(0.005), 0.5 / 50)
    want_y = port.fir_decimate_cc(port.shift_addition_cc(port.convert_u8_f(iq).view(c64), 0.11)[0], 50, taps)
    got_y = np.frombuffer(run(["ddc_u8_cc", 0.11, 50, 0.005, "HAMMING"], iq, 1024 * 100), c64)
    assert got_y.size == want_y.size and relrms(got_y, want_y) <= TOL
    dtaps = np.load(os.path.join(ROOT, "tests", "golden", "nfm_deemph_taps.npz"))["sr48000"]
    want, _ = port.nfm_chain(iq, 0.11, dtaps)
    got = np.frombuffer(run(["nfm_chain_u8_s16", 0.11], iq, 1024 * 100), np.int16)
    assert got.size == want.size
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert d.max() <= 1 and np.mean(d != 0) < 0.01
    am = "convert_u8_f | shift_addition_cc 0.11 | fir_decimate_cc 50 0.005 HAMMING | amdemod_cf | fastdcblock_ff | agc_ff | limit_ff | convert_f_s16"
    env = dict(os.environ, CSDR_AMD_BLOCK=str(1024 * 100))
    p = subprocess.run([CLI, "chain", am], input=iq.tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
    assert p.returncode == 0 and b"fused matrix-core front end" in p.stderr, p.stderr.decode()
    assert len(p.stdout) // 2 >= want_y.size - 2048


def test_cli_dynamic_bufsize_preamble(port):
    """csdr.c:330-391: with CSDR_DYNAMIC_BUFSIZE_ON=1 every command eats the "csdr"+int preamble and sends its own (size rule per command)."""
    import struct
    rng = np.random.default_rng(12)
    x = crand(rng, 50000)
    env = dict(os.environ, CSDR_DYNAMIC_BUFSIZE_ON="1", CSDR_AMD_BLOCK="8192")

    def one(cli, args, data):
        p = subprocess.run([cli] + args, input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=60)
        assert p.returncode == 0, p.stderr.decode()
        return p.stdout
    head = one(CLI, ["setbuf", "2048"], x.tobytes())
    assert head[:8] == b"csdr" + struct.pack("i", 2048) and head[8:] == x.tobytes()
    out = one(CLI, ["fir_decimate_cc", "10", "0.05", "HAMMING"], head)
    assert out[:8] == b"csdr" + struct.pack("i", 204)                     # the_bufsize / factor
    taps = port.firdes_lowpass_f(port.firdes_filter_len(0.05), 0.05)
    want = port.fir_decimate_cc(x, 10, taps)
    got = np.frombuffer(out[8:], c64)
    assert got.size == want.size and relrms(got, want) <= TOL
    out2 = one(CLI, ["fmdemod_quadri_cf"], out)
    assert out2[:8] == b"csdr" + struct.pack("i", 204)
    if os.path.exists(REF_CLI):                                            # the reference CLI sends the same preambles
        r1 = one(REF_CLI, ["fir_decimate_cc", "10", "0.05", "HAMMING"], head)
        assert r1[:8] == out[:8]
    # a stream without the preamble: warning, 8 bytes consumed, default size proposed (csdr.c:335-336)
    p = subprocess.run([CLI, "convert_u8_f"], input=bytes(range(64)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=60)
    assert b"preamble" in p.stderr and p.stdout[:8] == b"csdr" + struct.pack("i", 1024) and len(p.stdout) == 8 + 4 * 56


def test_cli_fifo_retune(port, tmp_path):
    """csdr.c:252-323, 883-921: `shift_addition_cc --fifo <path>`: first rate from the fifo, a later line retunes between blocks, phase carries on."""
    rng = np.random.default_rng(13)
    x = crand(rng, 2 * 4096)
    fifo = str(tmp_path / "ctl")
    os.mkfifo(fifo)
    env = dict(os.environ, CSDR_AMD_BLOCK="4096")
    p = subprocess.Popen([CLI, "shift_addition_cc", "--fifo", fifo], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    ctl = open(fifo, "w")
    ctl.write("0.05\n"); ctl.flush()
    p.stdin.write(x[:4096].tobytes()); p.stdin.flush()
    first = b""
    while len(first) < 8 * 4096:                                           # block 1 has come out: the process is now waiting for block 2
        chunk = p.stdout.read(8 * 4096 - len(first))
        assert chunk, p.stderr.read().decode()
        first += chunk
    ctl.write("-0.2\n"); ctl.flush()                                        # in the fifo before block 2 is even sent: applied to block 2
    p.stdin.write(x[4096:].tobytes()); p.stdin.close()
    rest = p.stdout.read()
    ctl.close()
    assert p.wait(timeout=30) == 0
    got = np.frombuffer(first + rest, c64)
    a, ph = port.shift_addition_cc(x[:4096], 0.05)
    b, _ = port.shift_addition_cc(x[4096:], -0.2, phase=ph)
    assert got.size == x.size
    assert relrms(got[:4096], a) <= TOL and relrms(got[4096:], b) <= TOL


def _read_until_quiet(f, quiet=1.0, limit=60.0):
    """what a running process has written so far: reads until nothing has arrived for `quiet` seconds"""
    import select, time
    out = b""; t_end = time.time() + limit
    while time.time() < t_end:
        r, _, _ = select.select([f], [], [], quiet if out else max(t_end - time.time(), quiet))      # (the first bytes may take a process start-up on a cold box)
        if not r: break
        chunk = os.read(f.fileno(), 1 << 20)
        if not chunk: break
        out += chunk
    return out


@pytest.mark.parametrize("spec", ["wfm", "ddc", "plain"])
def test_cli_chain_with_control_channel(port, tmp_path, spec):
    """`--fifo` INSIDE `csdr chain` (fusion and retune together; each command of the reference has its own control channel, csdr.c:252-323): the README.md:66
    pattern with `shift_addition_cc --fifo <path>` still becomes the fused WFM kernel, the three-command front end the fused DDC, and a chain without a fused
    form polls the stage's channel in front of every pass.  First rate from the fifo; a line written while the process waits for input is applied from the next
    block's first sample, phase carried (csdr.c:881-923)."""
    from tests_helpers import wfm_signal_u8
    fifo = str(tmp_path / "ctl"); os.mkfifo(fifo)
    n1, n2 = 65536, 65536 + 3072
    r1, r2 = -0.085, 0.21
    rng = np.random.default_rng(77)
    if spec == "plain":
        x = crand(rng, n1 + n2); raw = x.tobytes(); cut = 8 * n1
        chain = "shift_addition_cc --fifo %s | fmdemod_quadri_cf" % fifo
    else:
        u8 = np.concatenate([wfm_signal_u8(31, n1, offset=-r1), wfm_signal_u8(32, n2, offset=-r2)]); raw = u8.tobytes(); cut = 2 * n1
        chain = "convert_u8_f | shift_addition_cc --fifo %s | fir_decimate_cc 10 0.05 HAMMING" % fifo
        if spec == "wfm": chain += " | fmdemod_quadri_cf | fractional_decimator_ff 5 | deemphasis_wfm_ff 48000 50e-6 | convert_f_s16"
    env = dict(os.environ, CSDR_AMD_BLOCK="65536")
    p = subprocess.Popen([CLI, "chain", chain], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    ctl = open(fifo, "w")
    ctl.write("%g\n" % r1); ctl.flush()
    p.stdin.write(raw[:cut]); p.stdin.flush()
    first = _read_until_quiet(p.stdout)
    assert first, p.stderr.read().decode()
    ctl.write("%g\n" % r2); ctl.flush()
    p.stdin.write(raw[cut:]); p.stdin.close()
    rest = p.stdout.read(); err = p.stderr.read().decode()
    ctl.close()
    assert p.wait(timeout=30) == 0, err
    assert "reinitialized to 0.21" in err
    if spec == "wfm": assert "fused matrix-core kernel" in err
    if spec == "ddc": assert "fused matrix-core front end" in err
    # the oracle: the shift stage retuned at sample n1, everything behind it as one stream
    xf = x if spec == "plain" else port.convert_u8_f(u8).view(c64)
    a, ph = port.shift_addition_cc(xf[:n1], r1)
    b, _ = port.shift_addition_cc(xf[n1:], r2, phase=ph)
    sh = np.concatenate([a, b])
    if spec == "plain":
        want = port.fmdemod_quadri_cf(sh)[0]; got = np.frombuffer(first + rest, f32)
        assert got.size == want.size and relrms(got, want) <= TOL
        return
    dec = port.fir_decimate_cc(sh, 10, port.firdes_lowpass_f(79, 0.05))
    if spec == "ddc":
        got = np.frombuffer(first + rest, c64)
        assert abs(got.size - dec.size) <= 1
        m = min(got.size, dec.size); assert relrms(got[:m], dec[:m]) <= TOL
        return
    dem, _ = port.fmdemod_quadri_cf(dec)
    want = port.convert_f_s16(port.deemphasis_wfm_ff(dem[10::5], 50e-6, 48000)[0])      # fractional_decimator_ff 5 == x[5 k + 10] (exact at an integer rate)
    got = np.frombuffer(first + rest, np.int16)
    m = min(got.size, want.size)
    assert m >= want.size - 4 and np.abs(got[:m].astype(np.int32) - want[:m].astype(np.int32)).max() <= 1


@pytest.mark.parametrize("cmd,fn", [("shift_addfast_cc", "shift_addfast_cc"), ("shift_unroll_cc", "shift_unroll_cc"), ("shift_addition_fc", "shift_addition_fc")])
def test_cli_fifo_retune_other_shifters(port, tmp_path, cmd, fn):
    """csdr.c:757-792 (shift_addfast_cc), 808-843 (shift_unroll_cc), 3373-3407 (shift_addition_fc): the same --fifo protocol as shift_addition_cc --
    the rate comes from the control channel, a later line re-initialises the shifter between blocks, the phase carries on."""
    rng = np.random.default_rng(14)
    real = cmd.endswith("_fc")
    x = rng.uniform(-1, 1, 2 * 4096).astype(f32) if real else crand(rng, 2 * 4096)
    esz = 4 if real else 8
    fifo = str(tmp_path / "ctl"); os.mkfifo(fifo)
    env = dict(os.environ, CSDR_AMD_BLOCK="4096")
    p = subprocess.Popen([CLI, cmd, "--fifo", fifo], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    ctl = open(fifo, "w")
    ctl.write("0.05\n"); ctl.flush()
    p.stdin.write(x[:4096].tobytes()); p.stdin.flush()
    first = b""
    while len(first) < 8 * 4096:
        chunk = p.stdout.read(8 * 4096 - len(first))
        assert chunk, p.stderr.read().decode()
        first += chunk
    ctl.write("-0.2\n"); ctl.flush()
    p.stdin.write(x[4096:].tobytes()); p.stdin.close()
    rest = p.stdout.read()
    ctl.close()
    assert p.wait(timeout=30) == 0
    got = np.frombuffer(first + rest, c64)
    a, ph = getattr(port, fn)(x[:4096], 0.05)
    b, _ = getattr(port, fn)(x[4096:], -0.2, phase=ph)
    assert got.size == x.size and esz
    assert relrms(got[:4096], a) <= TOL and relrms(got[4096:], b) <= TOL


def test_cli_ddcd_time_domain_command_line(port):
    """ddcd's per-client pipeline, exactly as it launches it (ddcd_old.h:51-57, ddcd_old.cpp:474-492): `csdr shift_unroll_cc --fd <n> | csdr fir_decimate_cc
    <D> <tbw>` with the shift rate arriving on an inherited pipe fd."""
    rng = np.random.default_rng(15)
    D, tbw = 8, 0.05
    x = crand(rng, 40000)
    r, w = os.pipe()
    os.set_inheritable(r, True)
    env = dict(os.environ, CSDR_AMD_BLOCK="8192")
    p1 = subprocess.Popen([CLI, "shift_unroll_cc", "--fd", str(r)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, pass_fds=(r,))
    p2 = subprocess.Popen([CLI, "fir_decimate_cc", str(D), str(tbw)], stdin=p1.stdout, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    p1.stdout.close(); os.close(r)
    os.write(w, b"0.1234\n")
    p1.stdin.write(x.tobytes()); p1.stdin.close()
    out = p2.stdout.read()
    os.close(w)
    assert p1.wait(timeout=60) == 0 and p2.wait(timeout=60) == 0, (p1.stderr.read().decode(), p2.stderr.read().decode())
    sh, _ = port.shift_unroll_cc(x, 0.1234)
    want = port.fir_decimate_cc(sh, D, port.firdes_lowpass_f(port.firdes_filter_len(tbw), 0.5 / D))
    got = np.frombuffer(out, c64)
    assert got.size == want.size and relrms(got, want) <= TOL


def test_cli_live_stream_latency_and_ragged_writes(port):
    """The reader hands on whatever has arrived once the reference's the_bufsize is there (csdr.c:232-247, 332) instead of waiting for a
    whole CSDR_AMD_BLOCK: with the default (1 Mi element) block, the output of the first 1024 floats must come back while stdin stays open;
    writes that end in the middle of a sample are reassembled."""
    rng = np.random.default_rng(16)
    x = rng.uniform(-1.5, 1.5, 1024 * 3 + 100).astype(f32)
    env = dict(os.environ); env.pop("CSDR_AMD_BLOCK", None)
    p = subprocess.Popen([CLI, "limit_ff", "0.7"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    raw = x.tobytes()
    p.stdin.write(raw[:4096 + 3]); p.stdin.flush()                          # 1024 floats and 3 bytes of the next one
    first = b""
    while len(first) < 4096:                                              # must arrive without EOF: a blocking full-block read would hang here
        chunk = p.stdout.read(4096 - len(first))
        assert chunk, p.stderr.read().decode()
        first += chunk
    p.stdin.write(raw[4096 + 3:]); p.stdin.close()
    rest = p.stdout.read()
    assert p.wait(timeout=30) == 0
    got = np.frombuffer(first + rest, f32)
    assert np.array_equal(got, port.limit_ff(x, 0.7))


def test_cli_resident_chain_on_a_live_stream(port):
    """CSDR_AMD_RESIDENT=1 csdr wfm_chain_u8_s16: the chain through the resident ring (one persistent grid, no launch per block) on a LIVE 2.4 MS/s stream -- a
    16384-sample block every 6.8 ms, as the reference's stages read them (csdr.c:189-193, 330-392): every block's audio comes back before the next block is due (the
    reference's own latency is a block), the grid stays on the GPU between blocks, the stream equals the oracle's."""
    import threading
    import time
    T, nb = 16384, 60
    iq = fm_iq(np.random.default_rng(31), nb * T + 5000)                   # (5000 samples behind the last whole block: dropped at EOF like a partial read, csdr.c:232-247)
    env = dict(os.environ, CSDR_AMD_RESIDENT="1"); env.pop("CSDR_AMD_BLOCK", None)
    p = subprocess.Popen([CLI, "wfm_chain_u8_s16", "-0.085"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    got = []; stamps = []

    def reader():
        while True:
            chunk = p.stdout.read1(65536) if hasattr(p.stdout, "read1") else p.stdout.read(4096)
            if not chunk:
                return
            got.append(chunk); stamps.append((time.perf_counter(), sum(len(c) for c in got) // 2))
    th = threading.Thread(target=reader); th.start()
    sent = []
    raw = iq.tobytes()
    need_of = lambda k: (((k + 1) * T - 79) // 10 - 10) // 5 + 1           # audio samples that exist once block k is in: j_hi(k) + 1
    n_head = 3                                                              # the process starts up on these (context, ring, first launch: a second or more on a busy box) ...
    for k in range(n_head):
        p.stdin.write(raw[2 * T * k:2 * T * (k + 1)]); p.stdin.flush()
        sent.append(time.perf_counter())
    t_lim = time.perf_counter() + 60
    while not (stamps and stamps[-1][1] >= need_of(n_head - 1)):           # ... and only when their audio is back does the live part begin
        assert time.perf_counter() < t_lim and p.poll() is None, "no audio from the first blocks"
        time.sleep(0.002)
    t_next = time.perf_counter() + 0.02
    for k in range(n_head, nb):
        while time.perf_counter() < t_next:
            time.sleep(0.0005)
        p.stdin.write(raw[2 * T * k:2 * T * (k + 1)]); p.stdin.flush()
        sent.append(time.perf_counter())
        t_next += T / 2.4e6
    time.sleep(0.05)
    tail_at = len(stamps)
    p.stdin.write(raw[2 * T * nb:]); p.stdin.close()
    th.join(timeout=60)
    assert p.wait(timeout=60) == 0
    err = p.stderr.read().decode()
    assert "resident grid" in err, err
    out = np.frombuffer(b"".join(got), np.int16)
    want, _ = port.wfm_chain(iq[:2 * T * nb], -0.085, 10, port.firdes_lowpass_f(port.firdes_filter_len(0.05), 0.05))
    m = min(out.size, want.size)
    assert 0 <= out.size - want.size <= 2
    d = np.abs(out[:m].astype(np.int32) - want[:m].astype(np.int32))
    assert d.max() <= 1 and np.mean(d != 0) < 0.01
    # latency: block k's last audio sample is j_hi(k) = ((k + 1) T - 79) // 10 - 10) // 5; when did the output reach that count?
    lat = []
    for k in range(5, nb):
        need = need_of(k)
        t_out = next((t for t, n in stamps[:tail_at] if n >= need), None)
        assert t_out is not None, "block %d's audio did not arrive while the stream was live" % k
        lat.append(t_out - sent[k])
    assert np.median(lat) < T / 2.4e6, "median latency %.2f ms" % (1e3 * np.median(lat))      # before the next block is due: the reference's own latency is a block
    print("resident chain on a live stream: median %.2f ms, max %.2f ms from a block's last byte to its audio" % (1e3 * np.median(lat), 1e3 * max(lat)))


def test_cli_argument_validation():
    """bad arguments end with the reference's badsyntax exit status instead of a crash (SIGFPE on a zero block / decimation)"""
    for args in (["fastagc_ff", "0"], ["fir_decimate_cc", "0"], ["fir_decimate_cc", "x"]):
        p = subprocess.run([CLI] + args, input=b"", stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
        assert p.returncode == 255 and p.stderr, args


# ---------------------------------------------------------------- f2 commands
def test_cli_f2_commands(port):
    rng = np.random.default_rng(14)
    x = crand(rng, 30000)
    assert relrms(np.frombuffer(run(["amdemod_cf"], x), f32), port.amdemod_cf(x)) <= TOL
    assert run(["amdemod_estimator_cf"], x) == port.amdemod_estimator_cf(x).tobytes()
    assert run(["realpart_cf"], x) == port.realpart_cf(x).tobytes()
    assert relrms(np.frombuffer(run(["logpower_cf", -70], x), f32), port.logpower_cf(x, -70)) <= TOL
    assert relrms(np.frombuffer(run(["fmdemod_atan_cf"], x), f32), port.fmdemod_atan_cf(x)[0]) <= TOL
    a = (rng.uniform(-1, 1, 50000) + 0.2).astype(f32)
    assert run(["gain_ff", 2.5], a) == port.gain_ff(a, 2.5).tobytes()
    assert relrms(np.frombuffer(run(["dcblock_ff"], a), f32), port.dcblock_ff(a)[0]) <= TOL
    got = np.frombuffer(run(["fastdcblock_ff"], a), f32); want = port.fastdcblock_ff(a)[0]
    assert got.size == want.size and relrms(got, want) <= TOL
    sig = (a * np.repeat(rng.uniform(0.01, 1, 500), 100)).astype(f32)
    assert relrms(np.frombuffer(run(["agc_ff"], sig), f32), port.agc_ff(sig)[0]) <= TOL
    assert relrms(np.frombuffer(run(["agc_ff", 20, 0.5, 0.05, 0.001, 100, 5, 0.99], sig), f32),
                  port.agc_ff(sig, 1024, 20, 0.5, 0.05, 0.001, 100.0, 5, 0.99)[0]) <= TOL
    got = np.frombuffer(run(["fft_cc", 1024, 300, "HAMMING"], x, 4096), c64); want = port.fft_cc(x, 1024, 300)
    assert got.size == want.size and relrms(got, want) <= TOL
    got = np.frombuffer(run(["fft_cc", 256, 5000], x, 8192), c64); want = port.fft_cc(x, 256, 5000)
    assert got.size == want.size and relrms(got, want) <= TOL


def test_cli_am_and_ssb_chains(port):
    """README.md:95 (AM) and :110 (SSB) as in-process chains against the oracle, stage by stage stream models."""
    rng = np.random.default_rng(15)
    n = 400000
    t = np.arange(n)
    audio = 0.5 * np.sin(2 * np.pi * 700 / 2.4e6 * t)
    am = (0.5 * (1 + audio) * np.exp(2j * np.pi * 0.25 * t) + 0.01 * (rng.normal(size=n) + 1j * rng.normal(size=n)))
    iq = np.empty(2 * n, f32); iq[0::2] = am.real; iq[1::2] = am.imag
    iq = np.clip(np.round(127.5 * (iq + 1)), 0, 255).astype(np.uint8)
    xf = port.convert_u8_f(iq).view(c64)
    sh, _ = port.shift_addition_cc(xf, -0.25)
    taps = port.firdes_lowpass_f(port.firdes_filter_len(0.005), 0.5 / 50)
    dec = port.fir_decimate_cc(sh, 50, taps)
    # AM
    d, _ = port.fastdcblock_ff(port.amdemod_cf(dec))
    want = port.convert_f_s16(port.limit_ff(port.agc_ff(d)[0], 1.0))
    got = np.frombuffer(run(["chain", "convert_u8_f | shift_addition_cc -0.25 | fir_decimate_cc 50 0.005 HAMMING | amdemod_cf | fastdcblock_ff | agc_ff | limit_ff | convert_f_s16"], iq, 65536), np.int16)
    assert got.size == want.size
    dd = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert dd.max() <= 2 and np.mean(dd != 0) < 0.02
    # SSB
    nt = port.firdes_filter_len(0.05); fft = port.next_pow2(nt)
    if fft - nt < 200:
        fft *= 2
    bp = port.bandpass_fir_fft_cc(dec, port.firdes_bandpass_c(nt, 0.0, 0.1), fft)
    ssb = "convert_u8_f | shift_addition_cc -0.25 | fir_decimate_cc 50 0.005 HAMMING | bandpass_fir_fft_cc 0 0.1 0.05 | realpart_cf | %slimit_ff | convert_f_s16"
    want = port.convert_f_s16(port.limit_ff(port.gain_ff(port.realpart_cf(bp), 3.0), 1.0))
    got = np.frombuffer(run(["chain", ssb % "gain_ff 3 | "], iq, 65536), np.int16)
    assert got.size == want.size
    dd = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert dd.max() <= 2 and np.mean(dd != 0) < 0.02
    # with agc_ff in the chain only a loose comparison is meaningful: its hang/attack logic branches on float comparisons, so inputs that
    # differ in the last bit (here: the FFT filter's rounding) legitimately give gain tracks that differ by a fraction of a percent for
    # a while.  agc_ff itself is compared on identical input in test_cli_f2_commands / tests/test_gpu_parity.py::test_f2_agc.
    want = port.convert_f_s16(port.limit_ff(port.agc_ff(port.realpart_cf(bp))[0], 1.0))
    got = np.frombuffer(run(["chain", ssb % "agc_ff | "], iq, 65536), np.int16)
    assert got.size == want.size and relrms(got.astype(f32), want.astype(f32)) < 2e-2


def test_cli_f3_adpcm(port):
    """test_ima_adpcm.grc:721-722 round trip as CLI processes, and the waterfall compressor, byte for byte."""
    rng = np.random.default_rng(16)
    x = (8000 * np.sin(np.arange(50000) * 0.01) + rng.integers(-3000, 3000, 50000)).astype(np.int16)
    enc = run(["encode_ima_adpcm_i16_u8"], x)
    assert enc == port.encode_ima_adpcm_i16_u8(x)[0].tobytes()
    dec = run(["decode_ima_adpcm_u8_i16"], np.frombuffer(enc, np.uint8))
    assert dec == port.decode_ima_adpcm_u8_i16(np.frombuffer(enc, np.uint8))[0].tobytes()
    rows = (rng.uniform(-120, 10, 9 * 1024) + 20 * np.sin(np.arange(9 * 1024) * 0.05)).astype(f32)
    assert run(["compress_fft_adpcm_f_u8", 1024], rows, 4096) == port.compress_fft_adpcm_f_u8(rows, 1024).tobytes()
    # waterfall path of openwebrx in one process: fft_cc | logpower_cf | compress_fft_adpcm_f_u8
    sig = crand(rng, 40000)
    want = port.compress_fft_adpcm_f_u8(port.logpower_cf(port.fft_cc(sig, 1024, 3000), -70), 1024)
    got = np.frombuffer(run(["chain", "fft_cc 1024 3000 HAMMING | logpower_cf -70 | compress_fft_adpcm_f_u8 1024"], sig, 16384), np.uint8)
    assert got.size == want.size and np.mean(got != want) < 0.02      # float dB values x100 truncated to short: a last-bit difference can flip a code


# ---------------------------------------------------------------- f4: the ddcd topology (one forward transform, N channels, per-channel retune)
def test_cli_fastddc_bank_multi_rank_mode(tmp_path):
    """`csdr fastddc_bank_cc` started the way one rank of a multi-GPU bank is (CSDR_AMD_RANK / CSDR_AMD_WORLD / CSDR_AMD_COMM_FILE: the library's own RCCL communicator,
    the sharded bank entry point, the per-batch header broadcast, only this rank's outputs opened) with a world of ONE -- all a single-GPU box can run -- must write
    what the plain command writes, retune included (config 4's geometry: the sharded bank needs the matrix-core path)."""
    rng = np.random.default_rng(18)
    D, tbw, nch, nblk = 256, 0.001, 5, 5
    inp = 57344
    x = crand(rng, nblk * inp + 11)
    rates = [0.11, -0.2, 0.3, 0.0, -0.4321]
    res = {}
    for tag, extra in (("plain", {}), ("rank", {"CSDR_AMD_RANK": "0", "CSDR_AMD_WORLD": "1", "CSDR_AMD_COMM_FILE": str(tmp_path / "comm.id")})):
        outs = [str(tmp_path / ("%s_ch%d.bin" % (tag, k))) for k in range(nch)]
        fifo = str(tmp_path / ("ctl_" + tag)); os.mkfifo(fifo)
        args = ["fastddc_bank_cc", D, tbw, "HAMMING", fifo]
        for o, r in zip(outs, rates):
            args += [o, r]
        env = dict(os.environ, CSDR_AMD_BLOCK=str(2 * inp), **extra)
        p = subprocess.Popen([CLI] + [str(a) for a in args], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        ctl = open(fifo, "w")
        ctl.write("3 0.25\n"); ctl.flush()                            # applied before the first batch
        p.stdin.write(x.tobytes()); p.stdin.close()
        assert p.wait(timeout=120) == 0, p.stderr.read().decode()
        err = p.stderr.read().decode()
        ctl.close()
        if extra:
            assert "rank 0 of 1 serves channels 0 .. 4" in err and os.path.getsize(extra["CSDR_AMD_COMM_FILE"]) == 128
        res[tag] = [np.fromfile(o, c64) for o in outs]
    for a, b in zip(res["plain"], res["rank"]):
        assert a.size == b.size and a.size > 0 and relrms(b, a) <= 2e-6


@pytest.mark.parametrize("world,shard", [(2, "channels"), (2, "blocks"), (3, "auto")])
def test_cli_fastddc_bank_as_separate_processes(tmp_path, world, shard):
    """The multi-GPU form of `csdr fastddc_bank_cc` as it is deployed -- the SAME command line started once per rank with CSDR_AMD_RANK / CSDR_AMD_WORLD / CSDR_AMD_COMM_FILE
    (ddcd_old.cpp:238-252, 474-492: one process per client there) -- as `world` real PROCESSES on the one GPU of this box, joined by the inter-process transport
    (CSDR_AMD_COMM=ipc: unix sockets + HIP IPC; RCCL refuses two ranks per device): per-rank bootstrap, the per-batch header broadcast with a retune in it, each rank opening
    only its clients' outputs, both schedules (channel shards with the spectrum exchange, time slices with the output exchange; "auto" = the library's choice for the
    world size).  Every channel's file must hold what the single-process command writes."""
    rng = np.random.default_rng(28)
    D, tbw, nch, nblk = 256, 0.001, 7, 6
    inp = 57344
    x = crand(rng, nblk * inp + 5)
    rates = [0.11, -0.2, 0.3, 0.0, -0.4321, 0.05, 0.25]
    base = ["fastddc_bank_cc", D, tbw, "HAMMING"]

    def run_one(tag, envs):
        outs = [str(tmp_path / ("%s_ch%d.bin" % (tag, k))) for k in range(nch)]
        fifo = str(tmp_path / ("ctl_" + tag)); os.mkfifo(fifo)
        procs = []
        for r, extra in enumerate(envs):
            args = base + [fifo if r == 0 else "-"]
            for o, rt in zip(outs, rates):
                args += [o, rt]
            env = dict(os.environ, CSDR_AMD_BLOCK=str(2 * inp), **extra)
            procs.append(subprocess.Popen([CLI] + [str(a) for a in args], stdin=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env))
        ctl = open(fifo, "w")
        ctl.write("3 0.25\n5 -0.125\n"); ctl.flush()                    # applied before the first batch, by every rank
        procs[0].stdin.write(x.tobytes()); procs[0].stdin.close()
        errs = []
        for pr in procs:
            rc = pr.wait(timeout=180)
            errs.append(pr.stderr.read().decode())
            assert rc == 0, errs[-1][-1500:]
        ctl.close()
        return [np.fromfile(o, c64) for o in outs], errs

    want, _ = run_one("plain", [{}])
    comm = str(tmp_path / "comm")
    envs = [{"CSDR_AMD_RANK": str(r), "CSDR_AMD_WORLD": str(world), "CSDR_AMD_COMM": "ipc", "CSDR_AMD_COMM_FILE": comm, **({} if shard == "auto" else {"CSDR_AMD_SHARD": shard})} for r in range(world)]
    got, errs = run_one("w%d_%s" % (world, shard), envs)
    expect = {"channels": "channel shards", "blocks": "time slices", "auto": "time slices" if world > 2 else "channel shards"}[shard]
    served = []
    for r, e in enumerate(errs):
        assert "rank %d of %d, ipc transport, schedule: %s" % (r, world, expect) in e, e[-800:]
        m = [ln for ln in e.splitlines() if "serves channels" in ln]
        assert m, e[-800:]
        lo, hi = [int(v) for v in m[0].split("serves channels")[1].split("..")]
        served += list(range(lo, hi + 1))
    assert sorted(served) == list(range(nch))                          # every client is served by exactly one rank
    for k, (a, b) in enumerate(zip(want, got)):
        assert a.size == b.size and a.size > 0 and relrms(b, a) <= 2e-6, "channel %d" % k


def test_cli_fastddc_bank(port, tmp_path):
    import time
    rng = np.random.default_rng(17)
    D, tbw = 16, 0.02
    rates = [0.11, -0.2, 0.3]
    ddc, err = port.fastddc_init(tbw, D, rates[0])
    assert err == 0
    nblk = 6
    x = crand(rng, nblk * ddc.input_size + 50)
    outs = [str(tmp_path / ("ch%d.bin" % k)) for k in range(3)]
    args = ["fastddc_bank_cc", D, tbw, "HAMMING", "-"]
    for o, r in zip(outs, rates):
        args += [o, r]
    run(args, x, 2 * ddc.input_size)
    spectra = port.fastddc_fwd_cc(x, ddc)
    for k, r in enumerate(rates):
        dk, _ = port.fastddc_init(tbw, D, r)
        want = port.fastddc_inv_cc(spectra, dk, port.fastddc_taps_fft(dk, r, D))
        got = np.fromfile(outs[k], c64)
        assert got.size == want.size and relrms(got, want) <= TOL
    # retune channel 1 through the control fifo between two calls (two blocks per call here)
    fifo = str(tmp_path / "ctl"); os.mkfifo(fifo)
    args[4] = fifo
    for o in outs:
        os.remove(o)                                                  # the wait below must not see the first run's files
    env = dict(os.environ, CSDR_AMD_BLOCK=str(2 * ddc.input_size))
    p = subprocess.Popen([CLI] + [str(a) for a in args], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    ctl = open(fifo, "w")
    first = 2 * ddc.input_size
    p.stdin.write(x[:first].tobytes()); p.stdin.flush()
    d1, _ = port.fastddc_init(tbw, D, rates[1])
    part1 = port.fastddc_inv_cc(spectra[:2], d1, port.fastddc_taps_fft(d1, rates[1], D))
    deadline = time.time() + 60
    while not (os.path.exists(outs[1]) and os.path.getsize(outs[1]) >= 8 * part1.size):    # the first call (two blocks) is through
        assert time.time() < deadline and p.poll() is None
        time.sleep(0.05)
    ctl.write("1 0.05\n"); ctl.flush()
    p.stdin.write(x[first:].tobytes()); p.stdin.close()
    assert p.wait(timeout=60) == 0, p.stderr.read().decode()
    ctl.close()
    d2, _ = port.fastddc_init(tbw, D, 0.05)
    part2 = port.fastddc_inv_cc(spectra[2:], d2, port.fastddc_taps_fft(d2, 0.05, D))       # status restarts from zero like the reference's rebuild
    got = np.fromfile(outs[1], c64)
    want = np.concatenate([part1, part2])
    assert got.size == want.size
    assert relrms(got[:part1.size], part1) <= TOL, "before the retune"
    assert relrms(got[part1.size:], part2) <= TOL, "after the retune"
    got0 = np.fromfile(outs[0], c64)                                                         # the other channels are untouched by the retune
    d0, _ = port.fastddc_init(tbw, D, rates[0])
    assert relrms(got0, port.fastddc_inv_cc(spectra, d0, port.fastddc_taps_fft(d0, rates[0], D))) <= TOL


# ---------------------------------------------------------------- f4, first half: N-stream ingest into the batch API
@pytest.mark.parametrize("cmd", ["wfm_bank_u8_s16", "nfm_bank_u8_s16"])
def test_cli_stream_bank(port, tmp_path, cmd):
    """nmux-style producer for the batch API (nmux.cpp:177-283 fans one source out to N single-stream pipelines; here N streams enter ONE chain object):
    three u8 IQ files -> one process -> three audio files, each equal to the oracle's single-stream chain."""
    from tests_helpers import wfm_signal_u8, nfm_signal_u8
    nfm = cmd.startswith("nfm")
    n = 3 * 65536 + 7 * 1024
    sig = [(nfm_signal_u8(900 + k, n, offset=0.05) if nfm else wfm_signal_u8(900 + k, n)) for k in range(3)]
    args = [cmd, "-0.05" if nfm else "-0.085"]
    outs = []
    for k in range(3):
        fi = tmp_path / ("in%d.u8" % k); fo = tmp_path / ("out%d.s16" % k)
        sig[k].tofile(fi); outs.append(fo); args += [str(fi), str(fo)]
    env = dict(os.environ, CSDR_AMD_BANK_BLOCK="65536")
    p = subprocess.run([CLI] + args, stdin=subprocess.DEVNULL, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
    assert p.returncode == 0, p.stderr.decode()
    taps48 = np.load(os.path.join(ROOT, "tests", "golden", "nfm_deemph_taps.npz"))["sr48000"]
    for k in range(3):
        got = np.fromfile(outs[k], np.int16)
        if nfm:
            want, _ = port.nfm_chain(sig[k], -0.05, taps48)
        else:
            want, _ = port.wfm_chain(sig[k], -0.085, 10, port.firdes_lowpass_f(79, 0.05))
        m = min(got.size, want.size)
        assert m > 0 and abs(got.size - want.size) <= (1024 if nfm else 2)
        d = np.abs(got[:m].astype(np.int32) - want[:m].astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 0.05, "stream %d" % k


@pytest.mark.parametrize("cmd", ["nfm_bank_u8_s16", "wfm_bank_u8_s16"])
def test_cli_bank_rate_per_stream_and_control_channel(port, tmp_path, cmd):
    """`csdr nfm_bank_u8_s16 | wfm_bank_u8_s16 --ctl <fifo> r0,r1,r2 ...`: a shift rate per stream (ddcd tunes every client on its own, ddcd_old.h:51-61) and
    retunes over a control channel, lines "<stream> <rate>" (the `--fifo` protocol of shift_addition_cc, csdr.c:881-893, with a stream number in front).  The line
    is in the fifo before the process starts, so it is applied in front of the first pass: stream 1 runs at the NEW rate from its first sample (the mid-stream case,
    with its two-rate window, is tests/test_rates_gpu.py's)."""
    from tests_helpers import nfm_signal_u8, wfm_signal_u8
    nfm = cmd.startswith("nfm")
    n = 3 * 65536 + 5 * 1024
    rates = [-0.05, 0.3, 0.1234]
    new1 = -0.2
    eff = [rates[0], new1, rates[2]]
    sig = [(nfm_signal_u8 if nfm else wfm_signal_u8)(950 + k, n, offset=-eff[k]) for k in range(3)]
    ctl = tmp_path / "ctl.fifo"; os.mkfifo(ctl)
    keep = os.open(ctl, os.O_RDWR)                      # keeps the fifo open for writing while the command runs
    os.write(keep, b"1 %g\n" % new1)
    args = [cmd, "--ctl", str(ctl), ",".join("%g" % r for r in rates)]
    outs = []
    for k in range(3):
        fi = tmp_path / ("in%d.u8" % k); fo = tmp_path / ("out%d.s16" % k)
        sig[k].tofile(fi); outs.append(fo); args += [str(fi), str(fo)]
    env = dict(os.environ, CSDR_AMD_BANK_BLOCK="65536")
    try:
        p = subprocess.run([CLI] + args, stdin=subprocess.DEVNULL, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
    finally:
        os.close(keep)
    assert p.returncode == 0, p.stderr.decode()
    assert b"stream 1 reinitialized to -0.2" in p.stderr
    taps48 = np.load(os.path.join(ROOT, "tests", "golden", "nfm_deemph_taps.npz"))["sr48000"]
    for k in range(3):
        got = np.fromfile(outs[k], np.int16)
        if nfm:
            want, _ = port.nfm_chain(sig[k], eff[k], taps48)
        else:
            want, _ = port.wfm_chain(sig[k], eff[k], 10, port.firdes_lowpass_f(79, 0.05))
        m = min(got.size, want.size)
        assert m > 0 and abs(got.size - want.size) <= (1024 if nfm else 2)
        d = np.abs(got[:m].astype(np.int32) - want[:m].astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 0.05, "stream %d" % k
