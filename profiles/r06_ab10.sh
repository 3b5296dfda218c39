#!/bin/bash
# chan_analyze_p2 (matrix-pipe form): mirrored windows of the odd waves; timing, then the counters of the kept form
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_ab10.txt 2>&1
cp cubicsdr_amd/libcsdr_hip.so /tmp/orig.so
for rep in 1 2; do
for so in mx2 mx3 mx3_e12; do
cp _ab/$so.so cubicsdr_amd/libcsdr_hip.so
echo -n "$so "; python profiles/chan_quick.py 2>/dev/null
done
done
for so in mx2 mx3; do
cp _ab/$so.so cubicsdr_amd/libcsdr_hip.so
bash profiles/chan_counters.sh $so C3 > /dev/null 2>&1
python - <<PY
import json; d=json.load(open('gpurun_out/chanctr_${so}_C3.json'))
for k,v in d.items():
    print('$so', k, 'avg_us', v.get('_avg_ns',0)/1e3, 'fetch B/sample', v.get('FETCH_SIZE',0)*2*1024/ (128*1024068), 'write', v.get('WRITE_SIZE',0)*1024/(128*1024068), {c: round(v[c],3) for c in v if c.startswith('_')}, 'VALU', v.get('SQ_INSTS_VALU'), 'SALU', v.get('SQ_INSTS_SALU'), 'LDS', v.get('SQ_INSTS_LDS'))
PY
done
cp /tmp/orig.so cubicsdr_amd/libcsdr_hip.so
