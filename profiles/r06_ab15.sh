#!/bin/bash
# firpfbch2 with M / 2 odd on chan_analyze_p2 (matrix-pipe form): parity, then the rate against the two-factor kernel (lab switch CSDR_CHAN_P2_OS2=0 needs a lab build: the committed number of the old path is in profiles/r05_chan2_fft.txt)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_ab15.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "channelizer2 or m_twice_odd or batched_equals" 2>&1 | tail -3
python - <<'PY'
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench
from cubicsdr_amd.engine import Context, SDRPost
for M in (122, 66, 126, 34, 6, 120):
    fs = 500000 * M; block = -(-fs // 60 // M) * M; nb = max(1, (1 << 26) // block)
    dev = torch.device("cuda", 0); g = torch.Generator(device=dev); g.manual_seed(1)
    ring = torch.randn(nb * block, 2, generator=g, device=dev, dtype=torch.float32) * 0.05
    ctx = Context(0); post = SDRPost(ctx, fs, M, block, max_blocks=nb, oversampled=True)
    for _ in range(3): post.execute(ring, nb, block, bench.CENTER)
    ctx.synchronize(); t = time.perf_counter()
    for _ in range(60): post.execute(ring, nb, block, bench.CENTER)
    ctx.synchronize(); dt = (time.perf_counter() - t) / 60; n = nb * block
    print(json.dumps({"M": M, "oversampled": True, "kernel": post.kernel_name, "ms_per_batch_incl_dc": round(dt * 1e3, 4), "GSps": round(n / dt / 1e9, 2), "frac_of_8TBps_at_24B": round(24 * n / dt / 8e12, 3)}), flush=True)
    post.close(); ctx.close()
PY
for so in mx3 os2; do cp _ab/$so.so cubicsdr_amd/libcsdr_hip.so; echo -n "$so "; python profiles/chan_quick.py 2>/dev/null; done
cp _ab/os2.so cubicsdr_amd/libcsdr_hip.so
