#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_ab5.txt 2>&1
for v in new all_direct; do
cp _ab/$v.so cubicsdr_amd/libcsdr_hip.so
echo "== $v"; CHAN_BENCH_BASE=0 CHAN_BENCH_ITERS=100 python profiles/chan_bench.py M194 M202 M206 M214 M218 M226 M254 M262 M274 M298 M326 M358 M388 M398 2>/dev/null
done
cp _ab/new.so cubicsdr_amd/libcsdr_hip.so
