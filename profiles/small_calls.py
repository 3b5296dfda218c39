#!/usr/bin/env python3
"""Where do the ~100 us of a ONE-block call go? (measurement helper)  Host time spent inside each of the three stage calls, calls per second at 1 / 3
streams, and the GPU-side sum of the kernels of one call."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from cubicsdr_amd.engine import Context, DemodBank, SDRPost, SpectrumProcessor
cfg = dict(bench.CONFIGS["C3"])
FS, M, BLOCK, ND, F, kinds = cfg["fs"], cfg["M"], cfg["block"], cfg["n_demods"], cfg["fft"], cfg["kinds"]
dev = torch.device("cuda", 0)
ring = torch.randn(4 * BLOCK, 2, device=dev) * 0.05
for streams in (1, 3):
    os.environ["CSDR_STREAMS"] = str(streams)
    c = Context(0); p = SDRPost(c, FS, M, BLOCK, max_blocks=1); b = DemodBank(c, ND, max_blocks=1)
    for i, f in enumerate(bench.demod_frequencies(bench.CENTER, FS, ND)):
        k = kinds[i % len(kinds)]; b.configure(i, p, k, bench.MODEM_BW[k], f, bench.AUDIO_RATE)
    s = SpectrumProcessor(c, F, max_frames=BLOCK // (2 * F) + 2)
    x = ring[:BLOCK]
    for _ in range(20):
        p.execute(x, 1, BLOCK, bench.CENTER); b.execute(p); s.process(x, 1, BLOCK, contiguous=True)
    c.synchronize()
    n = 400
    t = [0.0, 0.0, 0.0]
    t0 = time.perf_counter()
    for _ in range(n):
        a = time.perf_counter(); p.execute(x, 1, BLOCK, bench.CENTER)
        b1 = time.perf_counter(); b.execute(p)
        c1 = time.perf_counter(); s.process(x, 1, BLOCK, contiguous=True)
        d = time.perf_counter()
        t[0] += b1 - a; t[1] += c1 - b1; t[2] += d - c1
    host = time.perf_counter() - t0
    c.synchronize()
    wall = time.perf_counter() - t0
    c.profile_enable(1)
    for _ in range(50):
        p.execute(x, 1, BLOCK, bench.CENTER); b.execute(p); s.process(x, 1, BLOCK, contiguous=True)
    c.synchronize()
    prof = c.profile()
    gpu_us = sum(v[0] / 50 for v in prof.values()) * 1e3
    launches = sum(v[2] for v in prof.values()) / 50
    print(json.dumps({"streams": streams, "calls_per_s": round(n / wall), "us_per_call_wall": round(1e6 * wall / n, 1), "host_enqueue_us_per_call": round(1e6 * host / n, 1),
                      "host_us": {"post_execute": round(1e6 * t[0] / n, 1), "bank_execute": round(1e6 * t[1] / n, 1), "spec_process": round(1e6 * t[2] / n, 1)},
                      "gpu_kernel_us_per_call": round(gpu_us, 1), "launches_per_call": launches,
                      "kernels_us": {k: round(1e3 * v[0] / 50, 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}}), flush=True)
    s.close(); b.close(); p.close(); c.close()
