import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import csdr_amd, oracle
from tests_helpers import wfm_signal_u8
gpu = csdr_amd.Context(0); port = oracle.port()
taps = port.firdes_lowpass_f(79, 0.05)
n = 16384 * 15
base = [wfm_signal_u8(300 + s, n) for s in range(3)]
u8 = np.stack([base[s % 3] for s in range(19)])
want = [port.wfm_chain(b, -0.085, 10, taps) for b in base]
for pad, block in ((16, None), (16, 81920), (0, 81920), (16, 16384 * 6), (32, 65536 * 2)):
    s16, af = gpu.wfm_chain(u8, -0.085, 10, taps, block=block, pitch_pad=pad)
    print("pad", pad, "block", block, gpu.last_wfm_kernel)
    for s in (0, 1, 16, 18):
        pf = want[s % 3][1]; m = min(pf.size, af.shape[1])
        bad = np.nonzero(np.abs(af[s, :m] - pf[:m]) > 1e-4)[0]
        print("   stream", s, "m", m, "bad", bad.size, "first", bad[:6], "last", bad[-3:] if bad.size else "")
