#!/bin/bash
# front-end (tail-wave instances): three barrier intervals per chunk (sched3: the next chunk's mix in the last block-wide stage's interval) against four (sched4)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_ab21.txt 2>&1
cp _ab/sched3.so cubicsdr_amd/libcsdr_hip.so
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c3 or C3 or demod or nbfm or modem or sharded or batched" 2>&1 | tail -3
bash profiles/ab_so.sh C3 _ab/sched4.so _ab/sched3.so
bash profiles/ab_so.sh C3N _ab/sched4.so _ab/sched3.so
bash profiles/ab_so.sh C5 _ab/sched4.so _ab/sched3.so
