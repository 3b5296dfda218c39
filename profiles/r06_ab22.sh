#!/bin/bash
# front-end, three-interval schedule: the next chunk's mix whole beside stage 2 (split0) or dealt to the stage-1 and stage-2 intervals (split1)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_ab22.txt 2>&1
cp _ab/split1.so cubicsdr_amd/libcsdr_hip.so
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c3 or C3 or demod or nbfm or modem or sharded or batched" 2>&1 | tail -3
bash profiles/ab_so.sh C3 _ab/split0.so _ab/split1.so
bash profiles/ab_so.sh C3N _ab/split0.so _ab/split1.so
