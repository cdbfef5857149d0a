#!/usr/bin/env python3
"""tools/probes/cli_default_block_sweep.py -- every hot-path CLI command on a RAGGED input at the default block (4 Mi elements per read: the path users get, not the small
blocks the parity tests stream in) against the same command at 65536: exit codes, output lengths, differences.  A probe: prints a table, exits 1 on a failed command."""
import os
import subprocess
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CLI = os.path.join(ROOT, "csdr_amd", "csdr")
rng = np.random.default_rng(5)
n = 5 * 1048576 + 12345 + 3
t = np.arange(n)
sig = 0.6 * np.exp(1j * (2 * np.pi * 0.085 * t + 3 * np.sin(2 * np.pi * 1e-3 * t))) + 0.02 * (rng.normal(size=n) + 1j * rng.normal(size=n))
cf = sig.astype(np.complex64)
iq = np.empty(2 * n, np.float32); iq[0::2] = sig.real; iq[1::2] = sig.imag
u8 = np.clip(np.round(127.5 * (iq + 1)), 0, 255).astype(np.uint8)
fl = (0.5 * np.sin(2 * np.pi * 1e-3 * t) + 0.1 * rng.uniform(-1, 1, n)).astype(np.float32)
CMDS = [("convert_u8_f", u8, np.float32), ("wfm_chain_u8_s16 -0.085", u8, np.int16), ("nfm_chain_u8_s16 0.11", u8, np.int16), ("ddc_u8_cc 0.11 50 0.005 HAMMING", u8, np.complex64),
        ("shift_addition_cc 0.1", cf, np.complex64), ("shift_math_cc 0.1", cf, np.complex64), ("fir_decimate_cc 10 0.05 HAMMING", cf, np.complex64), ("fir_decimate_cc 50 0.005 HAMMING", cf, np.complex64),
        ("fmdemod_quadri_cf", cf, np.float32), ("bandpass_fir_fft_cc -0.1 0.1 0.05", cf, np.complex64), ("amdemod_cf", cf, np.float32), ("realpart_cf", cf, np.float32),
        ("fractional_decimator_ff 5", fl, np.float32), ("deemphasis_wfm_ff 48000 50e-6", fl, np.float32), ("deemphasis_nfm_ff 48000", fl, np.float32), ("convert_f_s16", fl, np.int16),
        ("fastagc_ff", fl, np.float32), ("limit_ff", fl, np.float32), ("fastdcblock_ff", fl, np.float32), ("dcblock_ff", fl, np.float32), ("gain_ff 0.5", fl, np.float32),
        ("chain convert_u8_f | shift_addition_cc -0.085 | fir_decimate_cc 10 0.05 HAMMING | fmdemod_quadri_cf | fractional_decimator_ff 5 | deemphasis_wfm_ff 48000 50e-6 | convert_f_s16", u8, np.int16)]
bad = 0
for cmd, data, ot in CMDS:
    outs = []
    argv = [CLI] + (["chain", cmd[6:]] if cmd.startswith("chain ") else cmd.split())
    for blk in (None, "65536"):
        env = dict(os.environ); env.pop("CSDR_AMD_BLOCK", None)
        if blk: env["CSDR_AMD_BLOCK"] = blk
        try:
            p = subprocess.run(argv, input=data.tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=60)
            outs.append((p.returncode, np.frombuffer(p.stdout[: len(p.stdout) // np.dtype(ot).itemsize * np.dtype(ot).itemsize], ot), p.stderr.decode()[-200:]))
        except subprocess.TimeoutExpired:
            outs.append((-9, np.zeros(0, ot), "timeout"))
    (ra, a, ea), (rb, b, eb) = outs
    m = min(a.size, b.size)
    if m:
        x = a[:m].astype(np.complex128 if ot == np.complex64 else np.float64); y = b[:m].astype(x.dtype)
        den = np.sqrt((np.abs(y) ** 2).sum()) or 1.0
        rel = float(np.sqrt((np.abs(x - y) ** 2).sum()) / den); mx = float(np.abs(x - y).max())
    else:
        rel = mx = float("nan")
    flag = "" if (ra == 0 and rb == 0) else "   <-- rc %d / %d: %s | %s" % (ra, rb, ea.strip().splitlines()[-1:] if ea.strip() else "", eb.strip().splitlines()[-1:] if eb.strip() else "")
    if ra or rb: bad += 1
    print("%-48s default %9d  64Ki %9d  rel %.2e  max %.3g%s" % (cmd[:48], a.size, b.size, rel, mx, flag))
sys.exit(1 if bad else 0)
