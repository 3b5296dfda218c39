# quick per-kernel timing of the C3 bench (measurement helper): bash profiles/bq.sh [ENV=VAL ...]
export CSDR_BUILD_LAB=1
for kv in "$@"; do export "$kv"; done
python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-latency --no-strong > gpurun_out/bq.json 2> gpurun_out/bq.err; tail -2 gpurun_out/bq.err | grep -v amdgpu.ids; python -c "
import json,sys; d=json.load(open('gpurun_out/bq.json')); r=d['roofline']; print('$*', round(d['value']), {k:round(v,4) for k,v in r['kernels_ms_per_batch'].items() if k.startswith('spec')}, {k:round(v['ms_per_batch'],4) for k,v in r['stages'].items()})"
