# A/B of whole library builds over several configurations (measurement helper): bash profiles/ab_so2.sh "C2 C5 C3" lib1.so lib2.so ...
cfgs=$1; shift
cp cubicsdr_amd/libcsdr_hip.so /tmp/libcsdr_hip_orig.so
for cfg in $cfgs; do
for rep in 1 2; do
for so in "$@"; do
cp "$so" cubicsdr_amd/libcsdr_hip.so || exit 1
python bench.py --config $cfg --steps 6 --warmup 2 --cpu-seconds 0 --no-latency --no-strong > gpurun_out/bq.json 2> gpurun_out/bq.err; python -c "
import json; d=json.load(open('gpurun_out/bq.json')); k=d['roofline']['kernels_ms_per_batch']; print('$cfg $so', round(d['value']), {n: round(v,4) for n,v in k.items() if 'frontend' in n})"
done
done
done
cp /tmp/libcsdr_hip_orig.so cubicsdr_amd/libcsdr_hip.so
