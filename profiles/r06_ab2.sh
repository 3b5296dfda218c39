#!/bin/bash
# round 6, chan_analyze_p2 variants on one box: steady-state time + rows digest (profiles/chan_quick.py) and HBM-side fetch per launch (rocprofv3 --pmc FETCH_SIZE, own pass)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
exec > gpurun_out/r06_ab2.txt 2>&1
export TMPDIR=/tmp
REPO=$(pwd)
for rep in 1 2; do
for v in "$@"; do
cp _ab/$v.so cubicsdr_amd/libcsdr_hip.so
echo "== $v"; python profiles/chan_quick.py 2>/dev/null
done
done
for v in "$@"; do
cp _ab/$v.so cubicsdr_amd/libcsdr_hip.so
rm -rf /tmp/pf; ( cd /tmp && CQ_WARM=4 CQ_N=12 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o b -- python $REPO/profiles/chan_quick.py ) > /tmp/pf.log 2>&1
python - "$v" <<'PY'
import csv, glob, sys
vals = []
for f in glob.glob("/tmp/pf/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "chan_analyze" in r.get("Kernel_Name", ""): vals.append(float(r["Counter_Value"]))
n = 128 * 1024068
print("== %s FETCH_SIZE KiB/launch avg %.0f over %d launches -> %.2f B/sample (x2 per the guide)" % (sys.argv[1], sum(vals) / max(1, len(vals)), len(vals), 2 * 1024 * sum(vals) / max(1, len(vals)) / n))
PY
done
cp _ab/new.so cubicsdr_amd/libcsdr_hip.so
