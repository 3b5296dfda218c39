#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run from the repo root through gpurun):
# usage: profiles/collect.sh TAG [BLOCKS_PER_BATCH] [CONFIG]
#   1. --kernel-trace --stats            -> per-kernel call counts / average durations (default two-stream run, and once more
#                                           with CSDR_STREAMS=1: every kernel alone on the device)
#   2. --pmc FETCH_SIZE  (own pass)      -> HBM read traffic per launch  (KiB; gfx950: doubled per MI355X_MICROARCH.md "HBM")
#   3. --pmc WRITE_SIZE  (own pass)      -> HBM write traffic per launch (KiB)
# Outputs land in gpurun_out/prof_<tag>/ ; the summaries are copied to profiles/ by hand and committed.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-r02}
CFG=${3:-C3}
OUT=/tmp/prof_$TAG
DST=$(pwd)/gpurun_out/prof_$TAG
mkdir -p $OUT $DST
REPO=$(pwd)
NB=${2:-64}
BENCH="python $REPO/bench.py --config $CFG --steps 3 --warmup 1 --batches 8 --cpu-seconds 0 --no-profile --no-latency --blocks $NB"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- $BENCH ) > $OUT/stats.log 2>&1
# the same command with the stages on ONE stream: overlap-free kernel durations (compare with roofline.solo of bench.py)
( cd /tmp && CSDR_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/solo -o b -- $BENCH ) > $OUT/solo.log 2>&1
( cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o b -- $BENCH ) > $OUT/fetch.log 2>&1
( cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o b -- $BENCH ) > $OUT/write.log 2>&1
python - "$OUT" "$DST" "$NB" "$CFG" <<'PY'
import csv, glob, json, sys, collections
out, dst = sys.argv[1], sys.argv[2]
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    rows = [r for r in csv.reader(open(f))]
    keep = [rows[0]] + [r for r in rows[1:] if "csdr::" in r[0]]
    csv.writer(open(dst + "/kernel_stats.csv", "w")).writerows(keep)
for f in glob.glob(out + "/solo/**/*kernel_stats.csv", recursive=True):
    rows = [r for r in csv.reader(open(f))]
    keep = [rows[0]] + [r for r in rows[1:] if "csdr::" in r[0]]
    csv.writer(open(dst + "/kernel_stats_one_stream.csv", "w")).writerows(keep)
traffic = collections.defaultdict(dict)
for name in ("fetch", "write"):
    acc = collections.defaultdict(list)
    for f in glob.glob(out + "/" + name + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if "csdr::" in k:
                acc[k.split("(")[0].replace("void ", "").replace("csdr::", "")].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        traffic[k][name.upper() + "_SIZE_KiB_avg_per_launch"] = sum(v) / len(v)
        traffic[k]["launches_" + name] = len(v)
traffic["_meta"] = {"blocks_per_launch": int(sys.argv[3]), "config": sys.argv[4], "command": "bench.py --config %s --steps 3 --warmup 1 --batches 8 --cpu-seconds 0 --no-profile --no-latency --blocks %s" % (sys.argv[4], sys.argv[3])}
json.dump(traffic, open(dst + "/pmc_traffic.json", "w"), indent=1, sort_keys=True)
print(open(dst + "/kernel_stats.csv").read())
print(json.dumps(traffic, indent=1, sort_keys=True))
PY
tail -2 $OUT/stats.log
