#!/usr/bin/env python3
"""C3 channelizer alone, steady state (measurement helper): python profiles/chan_quick.py  [ENV=VAL ...]  -> kernel ms mean/min/max over 200 launches after a 100-launch warm-up"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for kv in sys.argv[1:]:
    k, v = kv.split("=", 1); os.environ[k] = v
import torch
import bench
from cubicsdr_amd.engine import Context, SDRPost
FS, M, BLOCK, NB = 61_440_000, 122, 1_024_068, 128
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
ring = torch.randn(NB * BLOCK, 2, generator=g, device=dev, dtype=torch.float32) * 0.05
os.environ.setdefault("CSDR_STREAMS", "1")
ctx = Context(0); post = SDRPost(ctx, FS, M, BLOCK, max_blocks=NB)
WARM, N = int(os.environ.get('CQ_WARM', 100)), int(os.environ.get('CQ_N', 200))
for _ in range(WARM): post.execute(ring, NB, BLOCK, bench.CENTER)
ctx.synchronize(); ctx.profile_enable(1)
for _ in range(N): post.execute(ring, NB, BLOCK, bench.CENTER)
ctx.synchronize()
ms, n, _ = ctx.profile()["chan_analyze"]; lo, hi = ctx.profile_range()["chan_analyze"]
import hashlib
h = hashlib.sha256()
for ch in (0, 1, 30, 60, 61, 62, 121): h.update(post.read_channel(ch).tobytes())       # the same input in every run: equal digests = bit-identical rows
print(json.dumps({"args": sys.argv[1:], "rows_sha256": h.hexdigest()[:16], "kernel_ms_mean": round(ms / n, 4), "min": round(lo, 4), "max": round(hi, 4), "frac_of_8TBps": round(16 * NB * BLOCK / (ms / n * 1e-3) / 8e12, 3)}))
