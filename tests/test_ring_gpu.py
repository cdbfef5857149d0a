"""The RESIDENT form of the fused WFM chain (csdr_amd_wfm_ring_*, csdr_amd/csrc/wfm_ring.cpp + k_wfm_mfma_seq<false, true>): a persistent grid that walks a ring of
16384-sample blocks -- the reference's unit of work (csdr.c:189-193, 232-247, 330-392), retune between two blocks as csdr.c:881-923 -- against the CPU oracle's chain on
the same bytes and against the per-call chain object.  Also: the grid leaves by itself (idle / age) and is relaunched without a sample changing; a killed host
process does not keep the GPU."""
import os
import signal
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests_helpers import wfm_signal_u8  # noqa: E402

pytestmark = pytest.mark.gpu
c64, f32 = np.complex64, np.float32
BLOCK = 16384


@pytest.fixture(scope="module")
def gpu():
    import torch  # noqa: F401
    import csdr_amd
    g = csdr_amd.Context(0)
    yield g
    g.close()


def _taps(port):
    return port.firdes_lowpass_f(port.firdes_filter_len(0.05), 0.05)


def _lsb_check(got, want, what=""):
    m = min(got.size, want.size)
    assert 0 <= got.size - want.size <= 2, (what, got.size, want.size)          # the fused chain emits what is computable
    d = np.abs(got[:m].astype(np.int32) - want[:m].astype(np.int32))
    assert d.max() <= 1 and np.mean(d != 0) < 0.01, (what, int(d.max()), float(np.mean(d != 0)))


def test_ring_against_oracle_and_chain_object(gpu, port):
    """20 streams (two stream groups, the second one partial) x 40 blocks: every stream against the oracle's one-shot chain, and the same audio as the chain object
    called block by block."""
    nb, S = 40, 20
    taps = _taps(port)
    sigs = [wfm_signal_u8(9100 + s, nb * BLOCK) for s in range(4)]
    x = np.stack([sigs[s % 4] for s in range(S)])
    # idle_us 20 ms: the grid (12 workgroups: the copies run beside it) stays resident across the collects -- the audio the host reads is what the grid's write-through
    # stores put into memory, not what a kernel end flushed
    y = gpu.wfm_ring_chain(x, -0.085, 10, taps, block=BLOCK, n_slots=8, idle_us=20000.0)
    assert gpu.last_ring["grid"] == 2 * 6 and gpu.last_ring["launches"] <= 3
    for s in range(4):
        want, _ = port.wfm_chain(sigs[s], -0.085, 10, taps)
        _lsb_check(y[s], want, "stream %d" % s)
    for s in range(4, S):
        assert np.array_equal(y[s], y[s % 4]), "replica %d differs" % s        # same bytes in another row / stream group / workgroup: identical samples
    per_call, _ = gpu.wfm_chain(x[:4], -0.085, 10, taps, block=BLOCK, want_float=False)
    assert per_call.shape == y[:4].shape
    d = np.abs(per_call.astype(np.int32) - y[:4].astype(np.int32))
    assert d.max() <= 1 and np.mean(d != 0) < 0.01


def _stage_chain(port, u8, rates_at, taps):
    """the chain stage by stage with the shift rate changing at the given samples, phase carried (csdr.c:881-923); fractional_decimator_ff 5 == x[5 k + 10]"""
    xf = port.convert_u8_f(u8).view(c64)
    parts, ph = [], 0.0
    for i, (pos, r) in enumerate(rates_at):
        end = rates_at[i + 1][0] if i + 1 < len(rates_at) else xf.size
        yv, ph = port.shift_addition_cc(xf[pos:end], r, phase=ph)
        parts.append(yv)
    dec = port.fir_decimate_cc(np.concatenate(parts), 10, taps)
    dem, _ = port.fmdemod_quadri_cf(dec)
    aud = port.deemphasis_wfm_ff(dem[10::5], 50e-6, 48000)[0]
    return port.convert_f_s16(aud)


def _two_rate_signal(seed, n, cut, r1, r2):
    """an FM signal that sits at -r1 before sample `cut` and at -r2 behind it (the chain shifts it to the centre on both sides of the retune)"""
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    msg = np.sin(2 * np.pi * 1e3 / 2.4e6 * t) + 0.3 * rng.uniform(-1, 1, n)
    off = np.where(t < cut, -r1, -r2)
    sig = 0.7 * np.exp(2j * np.pi * (np.cumsum(0.03125 * msg) + np.cumsum(off))) + 0.01 * (rng.normal(size=n) + 1j * rng.normal(size=n))
    iq = np.empty(2 * n, f32); iq[0::2] = sig.real; iq[1::2] = sig.imag
    return np.clip(np.round(127.5 * (iq + 1)), 0, 255).astype(np.uint8)


def test_ring_1024_consecutive_blocks_with_a_retune(gpu, port):
    """1024 consecutive 16384-sample blocks (16.8 M samples per stream) through ONE ring, the shift rate changed between blocks 511 and 512 -- new rate from block 512's
    first sample, phase carried, the audio samples whose windows straddle the two rates included -- against the oracle's stages on the same bytes, every sample."""
    nb, cut = 1024, 512
    r1, r2 = -0.085, 0.1234
    taps = _taps(port)
    u8 = _two_rate_signal(77, nb * BLOCK, cut * BLOCK, r1, r2)
    x = np.stack([u8, u8, u8])
    t0 = time.time()
    y = gpu.wfm_ring_chain(x, r1, 10, taps, block=BLOCK, n_slots=8, retunes={cut: r2})
    wall = time.time() - t0
    want = _stage_chain(port, u8, [(0, r1), (cut * BLOCK, r2)], taps)
    _lsb_check(y[0], want, "retuned stream")
    assert np.array_equal(y[1], y[0]) and np.array_equal(y[2], y[0])
    # around the retune: audio sample j needs input up to 10 (5 j + 10) + 78; the first sample of block 512 is j0
    j0 = (cut * BLOCK - 79) // 50 - 1
    seg = slice(j0 - 8, j0 + 16)
    assert np.abs(y[0][seg].astype(np.int32) - want[seg].astype(np.int32)).max() <= 1
    assert gpu.last_ring["launches"] >= 2          # the retune stops the grid; the copies between blocks let it idle out now and then
    print("ring: 1024 blocks in %.2f s, %d launches of the grid" % (wall, gpu.last_ring["launches"]))


def test_ring_leaves_when_the_host_is_silent_and_resumes(gpu, port):
    """A host that stops posting does not keep the GPU: the grid leaves after idle_us; the next block relaunches it and the stream continues as if nothing had happened."""
    import ctypes as C
    nb = 24
    taps = _taps(port)
    u8 = wfm_signal_u8(9200, nb * BLOCK)
    ref = gpu.wfm_ring_chain(u8[None, :], -0.085, 10, taps, block=BLOCK, n_slots=6)[0]
    # the same with pauses of 30 ms every 5 blocks (idle 100 us) and with a grid that may live 0.05 ms only (it leaves after every few blocks)
    paused = gpu.wfm_ring_chain(u8[None, :], -0.085, 10, taps, block=BLOCK, n_slots=6, idle_us=100.0, pause_every=5, pause_s=0.03)[0]
    assert gpu.last_ring["launches"] >= 4
    assert np.array_equal(paused, ref)
    short = gpu.wfm_ring_chain(u8[None, :], -0.085, 10, taps, block=BLOCK, n_slots=6, life_ms=0.05)[0]
    assert gpu.last_ring["launches"] >= 3
    assert np.array_equal(short, ref)
    want, _ = port.wfm_chain(u8, -0.085, 10, taps)
    _lsb_check(ref, want)
    # residency as the host sees it
    L = gpu.L
    t = np.ascontiguousarray(taps, f32)
    r = L.csdr_amd_wfm_ring_create(gpu.h, 1, -0.085, 10, t.ctypes.data_as(C.c_void_p), t.size, 5, 50e-6, 48000, BLOCK, 6)
    assert r, gpu.err()
    try:
        assert L.csdr_amd_wfm_ring_set_timeouts(r, 2000.0, 250.0) == 0
        assert L.csdr_amd_wfm_ring_resident(r) == 0
        assert L.csdr_amd_wfm_ring_submit(r) == 0                      # (slot 0 holds 0x80 bytes: silence)
        assert L.csdr_amd_wfm_ring_wait(r, 0, 5.0) > 0
        assert L.csdr_amd_wfm_ring_resident(r) == 1                    # still there 2 ms after its last block ...
        time.sleep(0.1)
        assert L.csdr_amd_wfm_ring_resident(r) == 0                    # ... gone within idle_us
        assert L.csdr_amd_wfm_ring_submit(r) == 1 and L.csdr_amd_wfm_ring_wait(r, 1, 5.0) > 0
        assert L.csdr_amd_wfm_ring_launches(r) == 2
        assert L.csdr_amd_wfm_ring_stop(r) == 0 and L.csdr_amd_wfm_ring_resident(r) == 0
    finally:
        L.csdr_amd_wfm_ring_destroy(r)


CHILD = r'''
import sys, time, ctypes as C
sys.path.insert(0, %r)
import numpy as np
import csdr_amd
g = csdr_amd.Context(0)
t = np.ascontiguousarray(g.firdes_lowpass_f(79, 0.05), np.float32)
r = g.L.csdr_amd_wfm_ring_create(g.h, 64, -0.085, 10, t.ctypes.data_as(C.c_void_p), t.size, 5, 50e-6, 48000, 16384, 6)
assert r
assert g.L.csdr_amd_wfm_ring_set_timeouts(r, 3.0e6, 20000.0) == 0      # a grid that would wait 3 s for the next block
assert g.L.csdr_amd_wfm_ring_submit(r) == 0 and g.L.csdr_amd_wfm_ring_wait(r, 0, 5.0) > 0
assert g.L.csdr_amd_wfm_ring_resident(r) == 1
print("resident", flush=True)
time.sleep(60)
'''


def test_ring_killed_host_releases_the_gpu(gpu):
    """SIGKILL to a process whose grid is resident (and would wait 3 s for a next block): another process's kernels run within a bounded time afterwards."""
    p = subprocess.Popen([sys.executable, "-c", CHILD % ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    try:
        line = p.stdout.readline()
        assert line.strip() == b"resident", (line, p.stderr.read()[-800:] if p.poll() is not None else b"")
        os.kill(p.pid, signal.SIGKILL)
        p.wait(timeout=30)
        t0 = time.time()
        x = np.arange(256, dtype=np.uint8)
        y = gpu.convert_u8_f(np.tile(x, 4096))                          # a kernel of THIS process on the GPU the dead one occupied
        dt = time.time() - t0
        assert y.size == 256 * 4096 and dt < 10.0, dt
    finally:
        if p.poll() is None:
            p.kill()
