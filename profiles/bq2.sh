# front-end timing of a bench configuration (measurement helper): bash profiles/bq2.sh CONFIG [ENV=VAL ...]
export CSDR_BUILD_LAB=1
cfg=$1; shift
for kv in "$@"; do export "$kv"; done
python bench.py --config $cfg --steps 3 --warmup 2 --cpu-seconds 0 --no-latency --no-strong > gpurun_out/bq.json 2> gpurun_out/bq.err; python -c "
import json; d=json.load(open('gpurun_out/bq.json')); r=d['roofline']; print('$cfg $*', round(d['value']), {k:round(v,4) for k,v in r['kernels_ms_per_batch'].items() if k.startswith('demod_front')})"
