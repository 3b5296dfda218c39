#!/bin/bash
# front-end stage 0: four outputs per thread (quads) against two (pairs)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_ab18.txt 2>&1
bash profiles/ab_so.sh C3 _ab/pairs.so _ab/quads.so
bash profiles/ab_so.sh C3N _ab/pairs.so _ab/quads.so
cp _ab/quads.so cubicsdr_amd/libcsdr_hip.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c3 or C3 or demod or nbfm or modem or sharded" 2>&1 | tail -3
