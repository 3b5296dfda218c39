"""Shared helpers for the parity tests: seeded synthetic IQ and error metrics."""
import math

import numpy as np


def rel_err(a, b):
    """max |a - b| relative to the reference's peak magnitude (the tolerance unit used throughout: 1e-5)."""
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return 0.0
    s = float(np.max(np.abs(b)))
    return float(np.max(np.abs(a - b))) / max(s, 1e-30)


def demod_frequencies(center, fs, n):
    """evenly spaced, never on a channel centre: f0 + (k + 0.37) Fs / N - Fs / 2"""
    return [int(center + (k + 0.37) * fs / n - fs / 2) for k in range(n)]


def synth_iq(n, fs, center, demods, seed=0xC0B1C5D2, t0=0, noise=0.05, dc=(0.01, 0.01)):
    """complex64[n]: white noise sigma `noise` per component + one modulated carrier per demod + a DC offset.
    demods: list of (kind, frequency) with kind in NBFM/FM/AM/USB/LSB; amplitude 0.5 / sqrt(N)."""
    rng = np.random.default_rng(seed)
    t = (np.arange(n, dtype=np.float64) + t0) / fs
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * noise
    amp = 0.5 / math.sqrt(max(len(demods), 1))
    for kind, f in demods:
        df = f - center
        if kind in ("NBFM", "FM"):
            dev = 2500.0 if kind == "NBFM" else 50000.0
            ph = 2 * np.pi * df * t + (dev / 1000.0) * np.sin(2 * np.pi * 1000.0 * t)
            x += amp * np.exp(1j * ph)
        elif kind == "AM":
            x += amp * (1 + 0.8 * np.sin(2 * np.pi * 1000.0 * t)) * np.exp(2j * np.pi * df * t)
        elif kind == "USB":
            x += amp * np.exp(2j * np.pi * (df + 1000.0) * t)
        elif kind == "LSB":
            x += amp * np.exp(2j * np.pi * (df - 1000.0) * t)
        elif kind == "DSB":
            x += amp * np.sin(2 * np.pi * 700.0 * t) * np.exp(1j * (0.4 + 2 * np.pi * (df + 35.0) * t))   # suppressed carrier, 35 Hz off tune
        elif kind == "CW":
            x += amp * np.exp(2j * np.pi * (df + 60.0) * t) * (np.floor(t * 40.0) % 2 == 0)    # keyed carrier, 20 Hz dots
        elif kind == "I/Q":
            x += amp * np.exp(2j * np.pi * (df + 3000.0) * t)
    x += dc[0] + 1j * dc[1]
    return x.astype(np.complex64)
