#!/usr/bin/env python3
"""Where does the 5-11 % bimodality of the dominant kernel (chan_analyze_p2, C3) come from?  (measurement helper, not the product)

Needs the measurement build (CSDR_BUILD_LAB=1 python -m cubicsdr_amd.build): CSDR_OUT_OFFSET_KB, CSDR_CHAN_XCD, CSDR_CHAN_PCT, CSDR_LAB_TRACE.

  A  ten fresh csdr_post objects in turn (allocation churn in between), each timed over ITERS launches with per-launch HIP events:
     mean / min / max of the channelizer kernel, the buffer addresses the library printed, sclk / mclk / power right after
  B  ONE allocation, the output moved inside it in 64 KB steps (0 .. 4 MB)
  C  the input ring moved in 64 KB steps
  D  tile order (workgroups of one XCD take consecutive tiles) and resident share

Prints one JSON line per measurement."""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from cubicsdr_amd.engine import Context, SDRPost  # noqa: E402

FS, M, BLOCK, NB = 61_440_000, 122, 1_024_068, 128
ITERS = int(os.environ.get("VAR_ITERS", "60"))


def clocks():
    """sclk / mclk / power / temperature as rocm-smi reports them (best effort: {} when the tool is missing)"""
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20)
        d = json.loads(r.stdout)
        card = d[sorted(d)[0]]
        out = {}
        for k, v in card.items():
            kl = k.lower()
            if "sclk" in kl or "mclk" in kl or "fclk" in kl or "power" in kl or ("temperature" in kl and ("junction" in kl or "hotspot" in kl or "edge" in kl)):
                out[k] = v
        return out
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def timed(ring, env, label, iters=ITERS, with_clocks=False):
    for k, v in env.items():
        os.environ[k] = str(v)
    os.environ["CSDR_LAB_TRACE"] = "1"
    ctx = Context(0)
    post = SDRPost(ctx, FS, M, BLOCK, max_blocks=NB)
    for _ in range(3):
        post.execute(ring, NB, BLOCK, bench.CENTER)
    ctx.synchronize()
    os.environ.pop("CSDR_LAB_TRACE", None)
    ctx.profile_enable(1)
    t = time.perf_counter()
    for _ in range(iters):
        post.execute(ring, NB, BLOCK, bench.CENTER)
    ctx.synchronize()
    wall = (time.perf_counter() - t) / iters
    prof = ctx.profile()
    rng = ctx.profile_range()
    ms, n, _ = prof["chan_analyze"]
    rec = {"label": label, "env": env, "kernel_ms_mean": round(ms / n, 4), "kernel_ms_min": round(rng["chan_analyze"][0], 4), "kernel_ms_max": round(rng["chan_analyze"][1], 4),
           "wall_ms_incl_dc": round(wall * 1e3, 4), "ring_ptr": hex(ring.data_ptr())}
    if with_clocks:
        rec["clocks"] = clocks()
    print(json.dumps(rec), flush=True)
    ctx.profile_enable(False)
    post.close(); ctx.close()
    for k in env:
        os.environ.pop(k, None)
    return rec


def main():
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(1)
    slack = 1 << 20                                   # samples of slack behind the ring (part C moves the input)
    base = torch.randn(NB * BLOCK + slack, 2, generator=g, device=dev, dtype=torch.float32) * 0.05
    ring = base[: NB * BLOCK]
    print(json.dumps({"idle_clocks": clocks()}), flush=True)
    parts = os.environ.get("VAR_PARTS", "ABCD")
    keep = []
    if "A" in parts:
        for i in range(10):
            timed(ring, {}, "A%d fresh object" % i, with_clocks=True)
            keep.append(torch.empty((3 + 7 * i) << 20, dtype=torch.uint8, device=dev))      # churn: the next object's buffers land elsewhere
            if i % 3 == 2:
                keep.pop(0)
    if "B" in parts:
        for off in list(range(0, 1025, 64)) + [1536, 2048, 3072, 4096]:
            timed(ring, {"CSDR_OUT_OFFSET_KB": off}, "B out+%dKB" % off, iters=30)
    if "C" in parts:
        for k in range(0, 17):
            r2 = base[k * 8192: k * 8192 + NB * BLOCK]
            timed(r2, {"CSDR_OUT_OFFSET_KB": 0}, "C in+%dKB" % (64 * k), iters=30)
    if "D" in parts:
        for rep in range(3):
            timed(ring, {"CSDR_CHAN_XCD": 0}, "D xcd=0 #%d" % rep, iters=40)
            timed(ring, {"CSDR_CHAN_XCD": 1}, "D xcd=1 #%d" % rep, iters=40)
        for pct in (50, 75):
            timed(ring, {"CSDR_CHAN_PCT": pct}, "D resident %d%%" % pct, iters=40)
            timed(ring, {"CSDR_CHAN_PCT": pct, "CSDR_CHAN_XCD": 1}, "D resident %d%% xcd=1" % pct, iters=40)
    print(json.dumps({"end_clocks": clocks()}), flush=True)


if __name__ == "__main__":
    main()
