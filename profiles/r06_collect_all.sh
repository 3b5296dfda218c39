#!/bin/bash
# end-of-round evidence: rocprofv3 summaries of the headline configuration (profiles/collect.sh) + bench lines of every configuration
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash profiles/collect.sh r06_c3 128 C3 > gpurun_out/collect_r06_c3.log 2>&1
python bench.py > gpurun_out/r06_c3_bench.json 2> gpurun_out/r06_c3_bench.err
for cfg in C2 C3N C5; do python bench.py --config $cfg --cpu-seconds 0 --no-latency > gpurun_out/r06_$(echo $cfg | tr A-Z a-z)_bench.json 2>> gpurun_out/r06_bench.err; done
python bench.py --config C4 --shard broadcast --cpu-seconds 0 --no-latency > gpurun_out/r06_c4_bench.json 2>> gpurun_out/r06_bench.err
python bench.py --config C4 --shard slab --cpu-seconds 0 --no-latency > gpurun_out/r06_c4slab_bench.json 2>> gpurun_out/r06_bench.err
python bench.py --config C4 --shard slab --no-overlap --cpu-seconds 0 --no-latency > gpurun_out/r06_c4slab_sequential_bench.json 2>> gpurun_out/r06_bench.err
COLLECT_STATS_ONLY=1 bash profiles/collect.sh r06_c4 32 C4 > gpurun_out/collect_r06_c4.log 2>&1
COLLECT_STATS_ONLY=1 bash profiles/collect.sh r06_c5 32 C5 > gpurun_out/collect_r06_c5.log 2>&1
tail -3 gpurun_out/r06_bench.err
python -c "
import json
for c in ('c3','c2','c3n','c5','c4','c4slab','c4slab_sequential'):
    try:
        d=json.load(open('gpurun_out/r06_%s_bench.json'%c)); print(c, round(d['value']), d['roofline'].get('kernel'), round(d['roofline'].get('frac',0),3))
    except Exception as e: print(c, 'failed', e)
"
