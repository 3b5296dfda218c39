#!/usr/bin/env python3
"""Channelizer-only A/B timing on the GPU box (measurement helper, not the product; bench.py is the contract).

  python profiles/chan_bench.py [CASE ...]      CASE = C2 | C4 | C5 | C3 | M<channels>   (default: C2 C5 C4)

For every case the csdr_post object is configured twice -- CSDR_CHAN_FFT=0 (the two-factor direct-DFT kernel, round 3) and the
default (kernels_chanfft.hpp where the channel count allows) -- and `post.execute` over an HBM-resident noise ring is timed
(wall clock around ITERS launches, synchronised at both ends).  Prints one JSON line per (case, variant):
algorithmic bytes = 16 per input sample (8 read + 8 written), peak 8 TB/s."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from cubicsdr_amd.engine import Context, SDRPost  # noqa: E402

CASES = {"C2": (10_000_000, 20, 166_680, 256), "C3": (61_440_000, 122, 1_024_068, 128), "C5": (100_000_000, 200, 1_666_800, 32),
         "C4": (100_000_000, 1024, 1_667_072, 16), "C4L": (100_000_000, 1024, 1_667_072, 32), "C4X": (100_000_000, 1024, 1_667_072, 35)}
ITERS = int(os.environ.get("CHAN_BENCH_ITERS", "100"))


def run(name, fs, M, block, nb, variants):
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    ring = torch.randn(nb * block, 2, generator=g, device=dev, dtype=torch.float32) * 0.05
    ctx = Context(0)
    for label, env in variants:
        for k, v in env.items():
            os.environ[k] = v
        post = SDRPost(ctx, fs, M, block, max_blocks=nb)
        for _ in range(3):
            post.execute(ring, nb, block, bench.CENTER)
        ctx.synchronize()
        t = time.perf_counter()
        for _ in range(ITERS):
            post.execute(ring, nb, block, bench.CENTER)
        ctx.synchronize()
        dt = (time.perf_counter() - t) / ITERS
        n = nb * block
        print(json.dumps({"case": name, "M": M, "variant": label, "kernel": post.kernel_name, "samples_per_launch": n, "ms_per_batch_incl_dc": round(dt * 1e3, 4),
                          "GSps": round(n / dt / 1e9, 2), "frac_of_8TBps": round(16 * n / dt / 8e12, 3)}), flush=True)
        post.close()
        for k in env:
            os.environ.pop(k, None)
    ctx.close()


def main():
    names = sys.argv[1:] or ["C2", "C5", "C4"]
    for nm in names:
        if nm in CASES:
            fs, M, block, nb = CASES[nm]
        else:
            M = int(nm[1:])
            fs = 500_000 * M
            block = -(-fs // 60 // M) * M
            nb = max(1, (1 << 26) // block)
        variants = [("direct-dft", {"CSDR_CHAN_FFT": "0"}), ("default", {})] if os.environ.get("CHAN_BENCH_BASE", "1") == "1" else [("default", {})]
        extra = os.environ.get("CHAN_BENCH_VARIANTS", "")     # e.g. "tf32:CSDR_CHANFFT_TF=32;tf64:CSDR_CHANFFT_TF=64"
        for item in filter(None, extra.split(";")):
            label, kv = item.split(":", 1)
            variants.append((label, dict(p.split("=", 1) for p in kv.split(","))))
        run(nm, fs, M, block, nb, variants)


if __name__ == "__main__":
    main()
