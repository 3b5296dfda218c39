#!/bin/bash
# A/B of whole-library variants on the bench's per-kernel figures: bash profiles/r06_ab3.sh "CONFIGS" variant ...   (variants = _ab/NAME.so)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_ab3.txt 2>&1
cfgs=$1; shift
for cfg in $cfgs; do
for rep in 1 2; do
for v in "$@"; do
cp _ab/$v.so cubicsdr_amd/libcsdr_hip.so
python bench.py --config $cfg --steps 6 --warmup 2 --cpu-seconds 0 --no-latency --no-strong > gpurun_out/bq.json 2> gpurun_out/bq.err; python -c "
import json; d=json.load(open('gpurun_out/bq.json')); k=d['roofline']['kernels_ms_per_batch']; print('$cfg $v', round(d['value']), {n: round(v,4) for n,v in k.items() if v > 0.03 or 'demod' in n})"
done
done
done
cp _ab/new.so cubicsdr_amd/libcsdr_hip.so
