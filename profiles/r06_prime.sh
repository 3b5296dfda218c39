#!/bin/bash
# channel counts with a prime factor 29 .. 89: the direct prime pass of chan_analyze_fft against the two-factor direct-DFT kernel (lab build of the post unit)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_prime.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "channelizer_fft_sizes" 2>&1 | tail -4
cp _ab/post_lab.so cubicsdr_amd/libcsdr_hip.so
CHAN_BENCH_ITERS=100 CHAN_BENCH_VARIANTS="" python profiles/chan_bench.py M116 M124 M134 M142 M146 M148 M158 M164 M166 M172 M174 M178 M186 M188 M212 M222 M232 M236 M244 M246 M248 M258 M268 M282 M284 M290 M292 M296 M310 M316 M318 M328 M332 M344 M348 M354 M356 M366 M370 M372 M376 2>/dev/null
cp _ab/new.so cubicsdr_amd/libcsdr_hip.so
