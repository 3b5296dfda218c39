#!/bin/bash
# primes 157 .. 199: chirp-z pass (pmx: CSDR_CF_BLUE_MIN = 157) against the matrix-pipe direct pass (pmx211)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_ab13.txt 2>&1
cp cubicsdr_amd/libcsdr_hip.so /tmp/orig.so
for so in pmx pmx211; do
cp _ab/$so.so cubicsdr_amd/libcsdr_hip.so
echo "== $so"
CHAN_BENCH_ITERS=60 CHAN_BENCH_BASE=0 python profiles/chan_bench.py M314 M326 M334 M346 M358 M362 M382 M386 M394 M398 2>/dev/null
done
cp _ab/pmx211.so cubicsdr_amd/libcsdr_hip.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "channelizer_fft_sizes" 2>&1 | tail -3
cp /tmp/orig.so cubicsdr_amd/libcsdr_hip.so
