#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_small_cpp2.txt 2>&1
cp _ab/fetchpost.so cubicsdr_amd/libcsdr_hip.so
hipcc -O2 -std=c++17 profiles/experiments/r05_small_calls.cpp -Iinclude -Lcubicsdr_amd -lcsdr_hip -Wl,-rpath,$PWD/cubicsdr_amd -lpthread -o /tmp/small_calls 2>/dev/null
for rep in 1 2; do
for s in 3 5 2 1; do
echo "== streams $s"; CSDR_STREAMS=$s /tmp/small_calls 1500 2>&1 | tail -2
done
done
