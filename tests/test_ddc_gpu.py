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
