#!/bin/bash
# co-residency of the channelizer (memory pipe) and the front-end (LDS pipe) on three streams: shares of the resident workgroup slots each launch takes (lab build)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_ab11.txt 2>&1
cp cubicsdr_amd/libcsdr_hip.so /tmp/orig.so
cp _ab/lab.so cubicsdr_amd/libcsdr_hip.so
for rep in 1 2; do
for e in "100 100" "50 100" "50 50" "100 50" "75 75"; do
set -- $e
CSDR_CHAN_PCT=$1 CSDR_FE_PCT=$2 python bench.py --config C3 --steps 6 --warmup 2 --cpu-seconds 0 --no-latency --no-strong --no-profile --streams 3 > gpurun_out/bq.json 2> gpurun_out/bq.err; python -c "
import json; d=json.load(open('gpurun_out/bq.json')); print('chan_pct $1 fe_pct $2 streams 3:', round(d['value']), d['ms_per_step'])"
done
done
cp /tmp/orig.so cubicsdr_amd/libcsdr_hip.so
