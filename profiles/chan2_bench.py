#!/usr/bin/env python3
"""firpfbch2 (2x oversampled channelizer) timing, FFT kernel against the two-factor kernel (measurement build: CSDR_CHAN_FFT=0): python profiles/chan2_bench.py [M ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from cubicsdr_amd.engine import Context, SDRPost
dev = torch.device("cuda", 0)
for M in [int(a) for a in sys.argv[1:]] or [20, 200, 256]:
    fs = 500000 * M
    block = -(-fs // 60 // M) * M
    nb = max(2, (1 << 26) // block)
    ring = torch.randn(nb * block, 2, device=dev) * 0.05
    for label, env in (("two-factor", {"CSDR_CHAN_FFT": "0"}), ("fft", {})):
        for k, v in env.items(): os.environ[k] = v
        os.environ["CSDR_STREAMS"] = "1"
        ctx = Context(0); post = SDRPost(ctx, fs, M, block, max_blocks=nb, oversampled=True)
        for _ in range(30): post.execute(ring, nb, block, bench.CENTER)
        ctx.synchronize(); t = time.perf_counter()
        for _ in range(100): post.execute(ring, nb, block, bench.CENTER)
        ctx.synchronize(); dt = (time.perf_counter() - t) / 100
        n = nb * block
        print(json.dumps({"M": M, "variant": label, "kernel": post.kernel_name, "ms_per_batch_incl_dc": round(dt * 1e3, 4), "GSps": round(n / dt / 1e9, 1), "frac_of_8TBps (8 B read + 16 B written per sample)": round(24 * n / dt / 8e12, 3)}), flush=True)
        post.close(); ctx.close()
        for k in env: os.environ.pop(k, None)
