#!/bin/bash
# firpfbch2, M / 2 odd: chan_analyze_p2 (matrix-pipe form) against the two-factor kernel over the channel counts
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_ab16.txt 2>&1
cp cubicsdr_amd/libcsdr_hip.so /tmp/orig.so
for so in os2_none os2_all; do
cp _ab/$so.so cubicsdr_amd/libcsdr_hip.so
echo "== $so"
python - <<'PY'
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench
from cubicsdr_amd.engine import Context, SDRPost
for M in (6, 10, 14, 18, 22, 26, 30, 34, 38, 42, 46, 50, 54, 58, 62, 66, 74, 86, 98, 110, 122, 126):
    fs = 500000 * M; block = -(-fs // 60 // M) * M; nb = max(1, (1 << 26) // block)
    dev = torch.device("cuda", 0); g = torch.Generator(device=dev); g.manual_seed(1)
    ring = torch.randn(nb * block, 2, generator=g, device=dev, dtype=torch.float32) * 0.05
    ctx = Context(0); post = SDRPost(ctx, fs, M, block, max_blocks=nb, oversampled=True)
    for _ in range(3): post.execute(ring, nb, block, bench.CENTER)
    ctx.synchronize(); t = time.perf_counter()
    for _ in range(40): post.execute(ring, nb, block, bench.CENTER)
    ctx.synchronize(); dt = (time.perf_counter() - t) / 40; n = nb * block
    print(json.dumps({"M": M, "kernel": post.kernel_name, "ms": round(dt * 1e3, 4), "GSps": round(n / dt / 1e9, 2), "frac24": round(24 * n / dt / 8e12, 3)}), flush=True)
    post.close(); ctx.close()
PY
done
cp /tmp/orig.so cubicsdr_amd/libcsdr_hip.so
