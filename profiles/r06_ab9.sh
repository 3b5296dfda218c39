#!/bin/bash
# C3 bench on 1 / 3 / 5 streams with the matrix-pipe channelizer
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_ab9.txt 2>&1
for rep in 1 2; do
for st in 1 3 5 2; do
python bench.py --config C3 --steps 6 --warmup 2 --cpu-seconds 0 --no-latency --no-strong --no-profile --streams $st > gpurun_out/bq.json 2> gpurun_out/bq.err; python -c "
import json; d=json.load(open('gpurun_out/bq.json')); print('streams $st', round(d['value']), d['ms_per_step'])"
done
done
