#!/bin/bash
# direct prime pass of chan_analyze_fft: vector form (pvalu) against the matrix-pipe form (pmx); parity tests of the channel counts first
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_ab12.txt 2>&1
cp cubicsdr_amd/libcsdr_hip.so /tmp/orig.so
cp _ab/pmx.so cubicsdr_amd/libcsdr_hip.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "channelizer_fft_sizes or every_even" 2>&1 | tail -3
for so in pvalu pmx; do
cp _ab/$so.so cubicsdr_amd/libcsdr_hip.so
echo "== $so"
CHAN_BENCH_ITERS=60 CHAN_BENCH_BASE=0 python profiles/chan_bench.py M116 M124 M134 M146 M158 M166 M178 M188 M194 M202 M212 M232 M244 M254 M268 M282 M290 M298 M302 2>/dev/null
done
cp /tmp/orig.so cubicsdr_amd/libcsdr_hip.so
