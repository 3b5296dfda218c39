#!/bin/bash
# round 6, first A/B session on one box: mirrored window order of chan_analyze_p2, rotated oscillator table of the front-end, chunked launches of the fused spectrum chain
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_ab1.txt 2>&1
set -x
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "channelizer or c3 or mixed" 2>&1 | tail -5
for so in _ab/p2_r5order.so _ab/new.so; do cp $so cubicsdr_amd/libcsdr_hip.so; python profiles/chan_quick.py; done
bash profiles/ab_so.sh C3 _ab/p2_r5order.so _ab/fe_norot.so _ab/new.so
bash profiles/ab_so.sh C3N _ab/fe_norot.so _ab/new.so
for lib in _ab/spec_lab.so _ab/spec_lab_plain.so; do
cp $lib cubicsdr_amd/libcsdr_hip.so
for c in 0 32 64 128 256; do
CSDR_SPEC_CHUNK=$c python bench.py --config C3 --steps 6 --warmup 2 --cpu-seconds 0 --no-latency --no-strong > gpurun_out/bq.json 2> gpurun_out/bq.err; python -c "
import json; d=json.load(open('gpurun_out/bq.json')); k=d['roofline']['kernels_ms_per_batch']; print('$lib chunk $c', round(d['value']), {n: round(v,4) for n,v in k.items() if 'spec' in n})"
done
done
cp _ab/new.so cubicsdr_amd/libcsdr_hip.so
