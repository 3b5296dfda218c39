#!/bin/bash
# one-block calls from the C++ host of profiles/experiments/r05_small_calls.cpp against whole-library variants (no Python in the loop)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_small_cpp.txt 2>&1
for rep in 1 2 3; do
for v in "$@"; do
cp _ab/$v.so cubicsdr_amd/libcsdr_hip.so
hipcc -O2 -std=c++17 profiles/experiments/r05_small_calls.cpp -Iinclude -Lcubicsdr_amd -lcsdr_hip -Wl,-rpath,$PWD/cubicsdr_amd -lpthread -o /tmp/small_calls 2>/dev/null
echo "== $v"; /tmp/small_calls 1500 2>&1 | tail -4
done
done
