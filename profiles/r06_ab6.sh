#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_ab6.txt 2>&1
cp _ab/post_lab.so cubicsdr_amd/libcsdr_hip.so
for rep in 1 2; do
for e in "X=0" "CSDR_CHAN_XCD=0" "CSDR_CHAN_XCD=1" "CSDR_CHAN_XCD=2" "CSDR_CHAN_PCT=75" "CSDR_CHAN_PCT=50"; do
python profiles/chan_quick.py $e 2>/dev/null
done
done
cp _ab/new.so cubicsdr_amd/libcsdr_hip.so
