#!/bin/bash
# matrix-pipe transform phase of chan_analyze_p2 against the vector form: kernel alone (digest of seven rows), then the C3 bench, then the channelizer parity tests
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_ab7.txt 2>&1
for rep in 1 2; do
for so in valu mx; do
cp _ab/$so.so cubicsdr_amd/libcsdr_hip.so
echo $so; python profiles/chan_quick.py 2>/dev/null
done
done
bash profiles/ab_so.sh C3 _ab/valu.so _ab/mx.so
cp _ab/mx.so cubicsdr_amd/libcsdr_hip.so
python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "channelizer or dc or c3 or C3" 2>&1 | tail -5
