#!/bin/bash
cd /root/repo
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for mode in "--one-stream" ""; do
python bench.py --steps 20 --warmup 3 --cpu-seconds 0 $mode > gpurun_out/bench_e.json 2> gpurun_out/bench_e.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_e.json"))
print("$mode", d["value"], d["ms_per_step"], d["config"]["event_ms_per_step"])
k=d["roofline"]["kernels_ms_per_step"]
print({a: round(v*1000,1) for a,v in k.items()}, "sum", round(sum(k.values())*1000,1))
PY
done
python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-profile | python -c "import sys,json; d=json.load(sys.stdin); print('noprof', d['value'], d['ms_per_step'])"
