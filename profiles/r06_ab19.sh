#!/bin/bash
# a single factor 17 / 19 / 23: in-register butterflies of the wide-odd instance (wide0) against the matrix-pipe prime pass (wide1)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_ab19.txt 2>&1
cp cubicsdr_amd/libcsdr_hip.so /tmp/orig.so
for so in wide0 wide1; do
cp _ab/$so.so cubicsdr_amd/libcsdr_hip.so
echo "== $so"
CHAN_BENCH_ITERS=60 CHAN_BENCH_BASE=0 python profiles/chan_bench.py M34 M68 M76 M92 M136 M152 M184 M204 M228 M272 M276 M340 M380 2>/dev/null
done
cp _ab/wide1.so cubicsdr_amd/libcsdr_hip.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "channelizer_fft_sizes" 2>&1 | tail -3
cp /tmp/orig.so cubicsdr_amd/libcsdr_hip.so
