#!/usr/bin/env python3
"""One-block calls under rocprofv3 --kernel-trace --memory-copy-trace (measurement helper): 300 calls of the C3 pipeline on the library's three streams.
  cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -- python profiles/small_trace.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from cubicsdr_amd.engine import Context, DemodBank, SDRPost, SpectrumProcessor
cfg = dict(bench.CONFIGS["C3"])
FS, M, BLOCK, ND, F, kinds = cfg["fs"], cfg["M"], cfg["block"], cfg["n_demods"], cfg["fft"], cfg["kinds"]
dev = torch.device("cuda", 0)
ring = torch.randn(4 * BLOCK, 2, device=dev) * 0.05
c = Context(0); p = SDRPost(c, FS, M, BLOCK, max_blocks=1); b = DemodBank(c, ND, max_blocks=1)
for i, f in enumerate(bench.demod_frequencies(bench.CENTER, FS, ND)):
    k = kinds[i % len(kinds)]; b.configure(i, p, k, bench.MODEM_BW[k], f, bench.AUDIO_RATE)
s = SpectrumProcessor(c, F, max_frames=BLOCK // (2 * F) + 2)
x = ring[:BLOCK]
n = int(os.environ.get("CALLS", "300"))
for _ in range(20):
    p.execute(x, 1, BLOCK, bench.CENTER); b.execute(p); s.process(x, 1, BLOCK, contiguous=True)
c.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    p.execute(x, 1, BLOCK, bench.CENTER); b.execute(p); s.process(x, 1, BLOCK, contiguous=True)
c.synchronize()
print("calls/s", n / (time.perf_counter() - t0))
