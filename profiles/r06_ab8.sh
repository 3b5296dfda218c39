#!/bin/bash
# chan_analyze_p2 (matrix-pipe form): variants, kernel alone
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_ab8.txt 2>&1
cp cubicsdr_amd/libcsdr_hip.so /tmp/orig.so
for rep in 1 2; do
for so in $CQ_VARIANTS; do
cp _ab/$so.so cubicsdr_amd/libcsdr_hip.so
echo -n "$so "; python profiles/chan_quick.py 2>/dev/null
done
done
cp /tmp/orig.so cubicsdr_amd/libcsdr_hip.so
