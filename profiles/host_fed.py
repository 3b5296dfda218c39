"""Host-fed (PCIe-inclusive) rate of the C3 pipeline: the same entry points handed HOST blocks (pageable, then page-locked).  Run on the GPU box from the repo root."""
import os, sys, time, ctypes as C, numpy as np
sys.path.insert(0, '.')
import cubicsdr_amd.hip as H
from cubicsdr_amd.engine import Context, DemodBank, SDRPost, SpectrumProcessor
from tests.util import demod_frequencies
fs, M, block, NB, nd = 61440000, 122, 1024068, 16, 256
center = 100000000
ctx = Context(0)
post = SDRPost(ctx, fs, M, block, max_blocks=NB)
bank = DemodBank(ctx, nd, max_blocks=NB)
kinds = ["NBFM", "AM", "USB"]; bw = {"NBFM": 12500, "AM": 6000, "USB": 5400}
for i, f in enumerate(demod_frequencies(center, fs, nd)):
    bank.configure(i, post, kinds[i % 3], bw[kinds[i % 3]], f)
spec = SpectrumProcessor(ctx, 65536, max_frames=(NB * block) // 131072 + 2)
rng = np.random.default_rng(0)
x = (rng.standard_normal(NB * block * 2).astype(np.float32) * 0.05).view(np.complex64)
L = H.lib()
for pinned in (False, True):
    if pinned:
        H.check(L.csdr_host_register(ctx.h, x.ctypes.data_as(C.c_void_p), x.nbytes))
    for _ in range(2):
        post.execute(x, NB, block, center); bank.execute(post); spec.process(x, NB, block, contiguous=True)
    ctx.synchronize()
    t0 = time.perf_counter(); reps = 8
    for _ in range(reps):
        post.execute(x, NB, block, center); bank.execute(post); spec.process(x, NB, block, contiguous=True)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print("host-fed (%s): %.0f MS/s (%.1f GB/s of IQ over PCIe, counted once)" % ("page-locked" if pinned else "pageable", reps * NB * block / dt / 1e6, reps * NB * block * 8 / dt / 1e9))
