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
        elif kind == "FMS":
            # FM broadcast multiplex: (L + R) + 19 kHz pilot + (L - R) on the suppressed 38 kHz subcarrier, 75 kHz deviation
            Ls = 0.8 * np.sin(2 * np.pi * 1000.0 * t) + 0.3 * np.sin(2 * np.pi * 3300.0 * t)
            Rs = 0.9 * np.sin(2 * np.pi * 700.0 * t + 1.0)
            mpx = 0.45 * (Ls + Rs) + 0.1 * np.sin(2 * np.pi * 19000.0 * t) + 0.45 * (Ls - Rs) * np.sin(2 * np.pi * 38000.0 * t)
            x += amp * np.exp(1j * (2 * np.pi * df * t + 2 * np.pi * 75000.0 * np.cumsum(mpx) / fs))
    x += dc[0] + 1j * dc[1]
    return x.astype(np.complex64)


def synth_iq_fast(n, fs, center, demods, seed=0xC0B1C5D2, noise=0.05, dc=(0.01, 0.01)):
    """the same signal model as synth_iq for the NBFM / AM / USB carriers of the BASELINE configurations, formed on the GPU when
    one is present (hundreds of carriers over millions of samples), numpy otherwise; returns a host complex64 array"""
    try:
        import torch
        use = torch.cuda.is_available()
    except Exception:
        use = False
    if not use:
        return synth_iq(n, fs, center, demods, seed=seed, noise=noise, dc=dc)
    import torch
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    x = torch.randn(n, 2, generator=g, device=dev, dtype=torch.float32) * noise
    amp = 0.5 / math.sqrt(max(len(demods), 1))
    SL = 1 << 21
    for s0 in range(0, n, SL):
        s1 = min(n, s0 + SL)
        t = torch.arange(s0, s1, device=dev, dtype=torch.float64) / fs
        tone = torch.sin(2 * math.pi * 1000.0 * t)
        ar = torch.zeros(s1 - s0, device=dev, dtype=torch.float64)
        ai = torch.zeros(s1 - s0, device=dev, dtype=torch.float64)
        for kind, f in demods:
            df = float(f - center)
            if kind in ("NBFM", "FM"):
                ph = (2 * math.pi * df) * t + ((2500.0 if kind == "NBFM" else 50000.0) / 1000.0) * tone
                ar += amp * torch.cos(ph); ai += amp * torch.sin(ph)
            elif kind == "AM":
                ph = (2 * math.pi * df) * t
                env = amp * (1 + 0.8 * tone)
                ar += env * torch.cos(ph); ai += env * torch.sin(ph)
            elif kind == "USB":
                ph = (2 * math.pi * (df + 1000.0)) * t
                ar += amp * torch.cos(ph); ai += amp * torch.sin(ph)
            else:
                raise ValueError(kind)
        x[s0:s1, 0] += (ar + dc[0]).float()
        x[s0:s1, 1] += (ai + dc[1]).float()
    return x.cpu().numpy().view(np.complex64).reshape(-1).copy()
