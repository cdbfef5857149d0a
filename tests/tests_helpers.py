"""Shared signal generators for the GPU tests."""
import numpy as np


def wfm_signal_u8(seed, n, offset=0.085):
    """u8 IQ of an FM broadcast-like signal `offset` x fs above centre (the chain under test shifts it back)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    msg = np.sin(2 * np.pi * 1e3 / 2.4e6 * t) + 0.3 * rng.uniform(-1, 1, n)
    sig = 0.7 * np.exp(1j * (2 * np.pi * np.cumsum(0.03125 * msg) + 2 * np.pi * offset * t)) + 0.01 * (rng.normal(size=n) + 1j * rng.normal(size=n))
    iq = np.empty(2 * n, np.float32); iq[0::2] = sig.real; iq[1::2] = sig.imag
    return np.clip(np.round(127.5 * (iq + 1)), 0, 255).astype(np.uint8)


def nfm_signal_u8(seed, n, offset=0.11, deviation=5e3 / 2.4e6):
    """u8 IQ of a narrow-band FM signal (BASELINE config 5: deviation 5 kHz at 2.4 MS/s) `offset` x fs above centre: it passes the
    25 kHz channel filter of the NFM chain, so relative errors are measured against a full-scale output."""
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    msg = np.sin(2 * np.pi * 1e3 / 2.4e6 * t) + 0.3 * np.convolve(rng.uniform(-1, 1, n + 199), np.ones(200) / 200, "valid")
    sig = 0.7 * np.exp(1j * (2 * np.pi * np.cumsum(deviation * msg) + 2 * np.pi * offset * t)) + 0.01 * (rng.normal(size=n) + 1j * rng.normal(size=n))
    iq = np.empty(2 * n, np.float32); iq[0::2] = sig.real; iq[1::2] = sig.imag
    return np.clip(np.round(127.5 * (iq + 1)), 0, 255).astype(np.uint8)
