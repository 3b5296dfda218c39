#!/bin/bash
# one-block calls (the real-time shape) of whole-library variants: bash profiles/r06_small.sh variant ...   -> bench.py's small_batches figures
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_small.txt 2>&1
for rep in 1 2; do
for v in "$@"; do
cp _ab/$v.so cubicsdr_amd/libcsdr_hip.so
python bench.py --config C3 --steps 3 --warmup 1 --cpu-seconds 0 --no-strong > gpurun_out/bq.json 2> gpurun_out/bq.err; python -c "
import json; d=json.load(open('gpurun_out/bq.json')); s=d['config']['small_batches']; print('$v', round(d['value']), {k: round(v['blocks_per_s']) for k,v in s.items()}, 'two threads', round(s['1']['two_host_threads']['blocks_per_s']), 'lib default streams', round(d['config'].get('library_default_streams',{}).get('MS_per_s',0)))"
done
done
