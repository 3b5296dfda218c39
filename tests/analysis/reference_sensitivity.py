"""TEST INFRASTRUCTURE (analysis script, CPU only, uses the oracle): how far does the REFERENCE's own output move when its input moves by less
than the parity tolerance?

The C3 parity run (tests/test_gpu_parity.py::test_c3_shape_mixed_m122) measures audio 9.1e-6 of the 1e-5 bound while the resampled IQ is at
1.2e-6.  This script feeds the reference chain (oracle/cubicsdr_chain.py on the reference's liquid binary) the SAME C3 signal twice -- once
as is, once with every sample multiplied by (1 + eps * noise), eps from half a float32 ulp upwards -- and reports, per eps, the worst
demodulator's change in resampled IQ and in audio with the tests' error metric (max |a - b| / peak |a|).  The ratio audio / IQ is the
conditioning of the modems on this signal (256 carriers sharing the amplitude: each FM carrier sits near the noise, the discriminator
amplifies a phase difference on the samples where the instantaneous amplitude dips): it is a property of the reference arithmetic, not of
the HIP path.

usage: python tests/analysis/reference_sensitivity.py [n_blocks] > profiles/r03_reference_sensitivity.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tests.test_gpu_parity as G                                              # noqa: E402
from tests.util import demod_frequencies, rel_err, synth_iq_fast               # noqa: E402


def main():
    n_blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    fs, M, block, center = 61440000, 122, 1024068, 100000000
    kinds = ["NBFM", "AM", "USB"] * 85 + ["NBFM"]
    freqs = demod_frequencies(center, fs, len(kinds))
    demods = list(zip(kinds, freqs))
    bws = [G._DEFAULT_BW[k] for k in kinds]
    x = synth_iq_fast(n_blocks * block, fs, center, demods, seed=41)           # the signal of test_c3_shape_mixed_m122
    base = G._ref_demods(x, fs, M, block, demods, bws, n_blocks)
    chans = list(G._ref_demods.last_channels)
    rng = np.random.default_rng(7)
    print("reference chain (%s), C3: %d demodulators, %d blocks of %d samples" % (G._backend(), len(kinds), n_blocks, block))
    print("%10s %12s %12s %8s   worst audio by modem kind" % ("eps", "iq moves", "audio moves", "ratio"))
    for eps in (6e-8, 2e-7, 6e-7):
        noise = (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size)).astype(np.complex64)
        y = (x * (1.0 + eps * noise)).astype(np.complex64)
        moved = G._ref_demods(y, fs, M, block, demods, bws, n_blocks)
        e_iq = e_au = 0.0
        by_kind = {}
        for i in range(len(kinds)):
            if chans[i] == 0:
                continue                                                       # (channel 0 sits behind the DC blocker: reported apart in the tests)
            ai = np.concatenate([w["iq"] for w in base[i]]); bi = np.concatenate([w["iq"] for w in moved[i]])
            aa = np.concatenate([w["audio"] for w in base[i]]); ba = np.concatenate([w["audio"] for w in moved[i]])
            if ai.size != bi.size or aa.size != ba.size:
                continue
            ei, ea = rel_err(bi, ai), rel_err(ba, aa)
            e_iq, e_au = max(e_iq, ei), max(e_au, ea)
            by_kind[kinds[i]] = max(by_kind.get(kinds[i], 0.0), ea)
        print("%10.1e %12.3e %12.3e %8.1f   %s" % (eps, e_iq, e_au, e_au / max(e_iq, 1e-30), {k: "%.2e" % v for k, v in sorted(by_kind.items())}))


if __name__ == "__main__":
    main()
