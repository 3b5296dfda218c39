#!/usr/bin/env python3
"""Per-launch time series of the C3 channelizer (measurement helper): every launch synchronised and timed on the host, back to back and after
idle gaps; a pure-HBM copy as the control.  Prints JSON lines."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from cubicsdr_amd.engine import Context, SDRPost  # noqa: E402

FS, M, BLOCK, NB = 61_440_000, 122, 1_024_068, 128


def series(fn, sync, n):
    out = []
    for _ in range(n):
        t = time.perf_counter()
        fn()
        sync()
        out.append(round((time.perf_counter() - t) * 1e3, 4))
    return out


def stats(v):
    s = sorted(v)
    return {"n": len(v), "min": s[0], "p10": s[len(s) // 10], "median": s[len(s) // 2], "p90": s[(9 * len(s)) // 10], "max": s[-1], "mean": round(sum(v) / len(v), 4)}


def main():
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(1)
    ring = torch.randn(NB * BLOCK, 2, generator=g, device=dev, dtype=torch.float32) * 0.05
    dst = torch.empty_like(ring)
    os.environ["CSDR_STREAMS"] = "1"
    ctx = Context(0)
    post = SDRPost(ctx, FS, M, BLOCK, max_blocks=NB)
    chan = lambda: post.execute(ring, NB, BLOCK, bench.CENTER)
    copy = lambda: dst.copy_(ring)
    for name, fn, sync in (("chan", chan, ctx.synchronize), ("copy", copy, torch.cuda.synchronize)):
        series(fn, sync, 20)
        for gap in (0.0, 0.0, 0.05, 0.5, 2.0, 0.0):
            if gap:
                time.sleep(gap)
            v = series(fn, sync, 300)
            print(json.dumps({"kernel": name, "idle_gap_s": gap, "stats": stats(v), "first20": v[:20], "every10th": v[::10]}), flush=True)
    # unsynchronised back-to-back launches: HIP-event durations of the kernel alone (no host in the loop)
    ctx.profile_enable(1)
    for _ in range(300):
        chan()
    ctx.synchronize()
    print(json.dumps({"kernel": "chan, 300 back-to-back launches, HIP events", "mean": ctx.profile()["chan_analyze"][0] / 300, "range": ctx.profile_range()["chan_analyze"]}), flush=True)
    post.close(); ctx.close()


if __name__ == "__main__":
    main()
