#!/usr/bin/env python3
"""How well is "1e-5 of the frame's peak" defined for the 2^21-point spectrum frame (C5)?  (measurement helper, run on the GPU box)

The display value of a bin is log10(maa + 0.25 - (floor - 0.75)) / log10(ceil + 1 - floor) (SpectrumVisualProcessor.cpp:562): `floor` is the
MINIMUM of the averaged magnitudes over all 2^21 bins, i.e. it is set by the one bin whose magnitude is ~1e-3 of the typical noise bin, where a
float32 transform's ABSOLUTE rounding error (which scales with the frame's total energy) is a per-cent-level relative error.  This script
forms two consecutive frames of the C5 test signal and compares, against a float64 transform: the reference's liquid FFT, the HIP FFT; then
the floors / display values each of them leads to."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cubicsdr_amd.engine import Context, SpectrumProcessor  # noqa: E402
from oracle.cubicsdr_chain import RefSpectrum  # noqa: E402
from tests.util import synth_iq_fast  # noqa: E402
import oracle.liquid_api as A  # noqa: E402

F = 1 << 20
N = 2 * F
fs = 100000000
be = "ref" if A.available("ref") else "port"
x = synth_iq_fast(2 * N, fs, 0, [("NBFM", 0.21 * fs), ("AM", -0.33 * fs), ("USB", 0.05 * fs), ("NBFM", -0.07 * fs)], seed=4242)
ctx = Context(0)
sp = SpectrumProcessor(ctx, F, max_frames=2)
rs = RefSpectrum(be, F)
for fr in range(2):
    fx = x[fr * N:(fr + 1) * N]
    exact = np.fft.fft(fx.astype(np.complex128))
    ref = rs.fft(fx).astype(np.complex128)
    gpu = sp.fft_only(fx).astype(np.complex128)
    mag = np.abs(exact)
    pk = mag.max()
    i0 = int(np.argmin(mag))
    print("frame %d: peak |X| %.6g, median %.4g, min %.4g at bin %d" % (fr, pk, np.median(mag), mag[i0], i0))
    for name, y in (("reference liquid fft", ref), ("HIP fft", gpu)):
        err = np.abs(y - exact)
        m = np.abs(y)
        print("   %-22s max |err| %.3g (%.3g of peak), rms %.3g | its min magnitude %.6g (exact %.6g): floor off by %.3g"
              % (name, err.max(), err.max() / pk, np.sqrt(np.mean(err ** 2)), m.min(), mag.min(), m.min() - mag.min()))
    print("   HIP vs reference: max |diff| %.3g of peak; min-magnitude difference %.3g" % (np.abs(gpu - ref).max() / pk, abs(np.abs(gpu).min() - np.abs(ref).min())))
    # display value sensitivity: d y / d floor for a bin of magnitude v is 1 / (ln 10 (v + 1 - floor) log10(ceil + 1 - floor))
    ce = pk
    for v in (mag.min(), np.median(mag)):
        print("   d(display)/d(floor) at |X| = %.3g: %.3g per unit" % (v, 1.0 / (np.log(10.0) * (v + 1.0 - mag.min()) * np.log10(ce + 1.0 - mag.min()))))
sp.close(); ctx.close()
