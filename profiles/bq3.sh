# whole-bench value of a configuration at 1 and 3 streams (measurement helper): bash profiles/bq3.sh CONFIG [ENV=VAL ...]
export CSDR_BUILD_LAB=1
cfg=$1; shift
for kv in "$@"; do export "$kv"; done
for st in 1 3; do
python bench.py --config $cfg --steps 6 --warmup 2 --cpu-seconds 0 --no-latency --no-strong --streams $st > gpurun_out/bq.json 2> gpurun_out/bq.err; python -c "
import json; d=json.load(open('gpurun_out/bq.json')); print('$cfg $* streams $st', round(d['value']))"
done
