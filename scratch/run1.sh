#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 120 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 > gpurun_out/bench_g.json 2> gpurun_out/bench_g.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_g.json"))
print("prof", d["value"], d["ms_per_step"])
k=d["roofline"]["kernels_ms_per_step"]
print({a: round(v*1000,1) for a,v in k.items()}, "sum", round(sum(k.values())*1000,1))
PY
CSDR_STREAMS=1 timeout 120 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 | python -c "
import sys,json; d=json.load(sys.stdin); k=d['roofline']['kernels_ms_per_step']; print('solo', {a: round(v*1000,1) for a,v in k.items()}, 'sum', round(sum(k.values())*1000,1))"
timeout 120 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-profile | python -c "import sys,json; d=json.load(sys.stdin); print('noprof', d['value'], d['ms_per_step'])"
timeout 120 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-profile --blocks 64 | python -c "import sys,json; d=json.load(sys.stdin); print('noprof64', d['value'], d['ms_per_step'])"
