"""GPU parity of the fused receiver front end convert_u8_f | shift_addition_cc | fir_decimate_cc (csdr_amd_ddc_*, BASELINE config 5's head)
against the oracle's stage-by-stage stream model.  Gate: relative RMS <= 1e-5 (BASELINE.json north_star, float paths)."""
import numpy as np
import pytest
from oracle import relrms
from tests_helpers import nfm_signal_u8

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


def oracle_front(port, u8, rate, D, taps):
    xf = port.convert_u8_f(u8).view(c64)
    sh, _ = port.shift_addition_cc(xf, rate)
    return port.fir_decimate_cc(sh, D, taps)


def check(y, want, streams, base_count):
    for s in streams:
        w = want[s % base_count]
        assert y.shape[1] == w.size, (y.shape, w.size)
        assert relrms(y[s], w) <= TOL, "stream %d: %g" % (s, relrms(y[s], w))


@pytest.mark.parametrize("D,L,rate", [(50, 801, 0.11), (50, 801, -0.4321), (10, 79, -0.085), (20, 321, 0.3), (50, 801, 0.05), (50, 801, 0.25)])
def test_ddc_single_call(gpu, port, D, L, rate):
    """(rates 0.05 and 0.25: the reference's float phasor recurrence drifts systematically; the front end replays it per chunk and corrects.)
    One call over the whole stream: the matrix-core kernel does the bulk, the plain kernel the head/tail outputs; 19 streams = one
    full and one ragged 16-stream block."""
    n = 1024 * 200
    taps = port.firdes_lowpass_f(L, 0.5 / D)
    base = [nfm_signal_u8(400 + s, n, offset=-rate) for s in range(3)]
    u8 = np.stack([base[s % 3] for s in range(19)])
    y = gpu.ddc_u8(u8, rate, D, taps)
    assert gpu.last_ddc_kernel == "k_ddc_mfma"
    want = [oracle_front(port, b, rate, D, taps) for b in base]
    check(y, want, (0, 1, 2, 15, 16, 18), 3)


@pytest.mark.parametrize("block", [65536, 1024 * 37, 1024 * 3])
def test_ddc_streaming_blocks(gpu, port, block):
    """Consecutive blocks: history, output index and shift phase carried in the object; blocks that are no multiple of the tile length;
    tiny blocks run entirely on the plain kernel; the last block is not a multiple of 1024 samples."""
    D, L, rate = 50, 801, 0.11
    n = 1024 * 150 + 700
    taps = port.firdes_lowpass_f(L, 0.5 / D)
    base = [nfm_signal_u8(500 + s, n, offset=-rate) for s in range(2)]
    u8 = np.stack([base[s % 2] for s in range(17)])
    y = gpu.ddc_u8(u8, rate, D, taps, block=block)
    if block >= 65536:
        assert "k_ddc_mfma" in gpu.ddc_kernels
    want = [oracle_front(port, b, rate, D, taps) for b in base]
    check(y, want, (0, 1, 16), 2)


def test_ddc_plain_kernel_paths(gpu, port):
    """Unaligned pitch (no line-aligned fetch possible) and an odd decimation (outside the matrix-core kernel's shapes): the plain kernel."""
    n = 1024 * 24
    rate = -0.2
    base = [nfm_signal_u8(600, n, offset=-rate)]
    u8 = np.stack([base[0]] * 3)
    for D, L, pad in ((50, 801, 16), (25, 401, 0), (7, 101, 0)):
        taps = port.firdes_lowpass_f(L, 0.5 / D)
        y = gpu.ddc_u8(u8, rate, D, taps, pitch_pad=pad)
        assert gpu.ddc_kernels == {"k_ddc_direct"}
        check(y, [oracle_front(port, base[0], rate, D, taps)], (0, 2), 1)


def test_ddc_matches_unfused_device_ops(gpu, port):
    """The fused front end against the three device-batch operators it replaces, on the GPU (larger input than the oracle comparison)."""
    D, L, rate = 50, 801, 0.05
    n = 1024 * 1024
    taps = port.firdes_lowpass_f(L, 0.5 / D)
    u8 = nfm_signal_u8(700, n, offset=-rate)[None, :]
    y = gpu.ddc_u8(u8, rate, D, taps)
    xf = gpu.convert_u8_f(u8[0]).view(c64)
    sh, _ = gpu.shift_addition_cc(xf, rate)
    ref = gpu.fir_decimate_cc(sh, D, taps)
    m = min(ref.size, y.shape[1])
    assert m >= (n - L) // D and relrms(y[0, :m], ref[:m]) <= TOL


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_ddc_and_nfm_random_call_sizes(gpu, port, seed):
    """The receiver front end and the NFM chain over RANDOM call sizes (multiples of 1024 samples, the stream's tail in a last ragged call): the matrix-core kernel
    takes whole calls -- first / last tile partial, windows starting in the history buffer, seed tables rebuilt on the way -- and only the ragged last call goes to the
    plain kernel; the complex stream against the oracle's three stages (1e-5), the chain's audio against the single-call run of the same object (bit identical)."""
    rng = np.random.default_rng(seed)
    D, L, rate = 50, 801, [0.11, -0.2718, 0.05][seed % 3]
    sizes = [1024 * int(rng.integers(2, 70)) for _ in range(int(rng.integers(3, 8)))]
    n = sum(sizes) + 1024 * 3 + int(rng.integers(1, 512)) * 2
    S = int(rng.integers(1, 34))
    taps = port.firdes_lowpass_f(L, 0.5 / D)
    base = [nfm_signal_u8(700 + 10 * seed + k, n, offset=-rate) for k in range(min(S, 2))]
    u8 = np.stack([base[s % len(base)] for s in range(S)])
    y = gpu.ddc_u8(u8, rate, D, taps, block=sizes + [1024 * 3])
    assert "k_ddc_mfma" in gpu.ddc_kernels
    want = [oracle_front(port, b, rate, D, taps) for b in base]
    check(y, want, sorted({0, S // 2, S - 1}), len(base))
    if seed == 21:
        pcm1, af1 = gpu.nfm_chain(u8[:5], -rate)
        pcm2, af2 = gpu.nfm_chain(u8[:5], -rate, block=sizes + [1024 * 3])
        m = min(pcm1.shape[1], pcm2.shape[1])
        assert m >= pcm1.shape[1] - 2048 and np.array_equal(pcm1[:, :m], pcm2[:, :m])
