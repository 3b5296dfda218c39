#!/bin/bash
# critically sampled M = 2 x prime below 66: vector form against the matrix-pipe form (one row tile)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_ab17.txt 2>&1
cp cubicsdr_amd/libcsdr_hip.so /tmp/orig.so
for so in mxmin33 mxmin3; do
cp _ab/$so.so cubicsdr_amd/libcsdr_hip.so
echo "== $so"
CHAN_BENCH_ITERS=60 CHAN_BENCH_BASE=0 python profiles/chan_bench.py M6 M10 M14 M22 M26 M34 M38 M46 M58 M62 2>/dev/null
done
cp /tmp/orig.so cubicsdr_amd/libcsdr_hip.so
