#!/bin/bash
# chan_analyze_fft variants (whole-library builds) on profiles/chan_bench.py: bash profiles/r06_ab4.sh "CASES" variant ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_ab4.txt 2>&1
cases=$1; shift
for rep in 1 2; do
for v in "$@"; do
cp _ab/$v.so cubicsdr_amd/libcsdr_hip.so
echo "== $v"; CHAN_BENCH_BASE=0 CHAN_BENCH_ITERS=200 python profiles/chan_bench.py $cases 2>/dev/null
done
done
cp _ab/new.so cubicsdr_amd/libcsdr_hip.so
