#!/bin/bash
# usage: lib_sweep.sh OUTFILE lib-label ...   (label 0 = the in-tree library, else scratch/abl/lib<label>.so)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/$1; shift
: > $OUT
for n in "$@"; do
  if [ $n = 0 ]; then LIB=$(pwd)/cubicsdr_amd/libcsdr_hip.so; else LIB=$(pwd)/scratch/abl/lib$n.so; fi
  python - "$LIB" $n >> $OUT 2>gpurun_out/lib_sweep_$n.err <<'PY'
import sys, json, io, runpy, contextlib
import cubicsdr_amd.hip as H
H.LIB_PATH = sys.argv[1]
import cubicsdr_amd.build as B
B.build = lambda *a, **k: None
label = sys.argv[2]
sys.argv = ["bench.py", "--steps", "4", "--warmup", "1", "--cpu-seconds", "0", "--no-latency"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    try:
        runpy.run_path("bench.py", run_name="__main__")
    except SystemExit:
        pass
d = json.loads(buf.getvalue().strip().splitlines()[-1])
k = d["roofline"]["kernels_ms_per_batch"]
print("lib", label, round(d["value"]), " ".join("%s=%.4f" % (a, b) for a, b in sorted(k.items(), key=lambda t: -t[1])[:7]))
PY
done
cat $OUT
