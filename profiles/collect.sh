#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run from the repo root through gpurun):
# usage: profiles/collect.sh TAG [BLOCKS_PER_BATCH] [CONFIG]
#   1. --kernel-trace --stats            -> per-kernel call counts / average durations of the default command (stages on ONE stream:
#                                           every kernel alone on the device), and once more with --streams 3 (the library's default
#                                           folding: kernels of three stages share the GPU, their durations stretch accordingly)
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
NB=${2:-128}
BENCH="python $REPO/bench.py --config $CFG --steps 2 --warmup 1 --batches 6 --cpu-seconds 0 --no-profile --no-latency --blocks $NB"
# counter passes: the same command on a noise-only ring (the kernels' work does not depend on the sample values)
PMCBENCH="$BENCH --ring noise"
# duration passes: a few hundred batches, so that the ~40 ms core-clock ramp after the idle start (profiles/r05_variance.txt: + 45 % on the issue-bound
# kernels for the first ~20 batches) is a few per cent of the average and the summary agrees with bench.py's steady-state HIP-event figures
# (rounds 1-4 timed 18 batches here: their averages carry the ramp)
LONGBENCH="python $REPO/bench.py --config $CFG --steps 8 --warmup 4 --batches 24 --cpu-seconds 0 --no-profile --no-latency --no-strong --blocks $NB --ring noise"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- $LONGBENCH ) > $OUT/stats.log 2>&1
# the same command on the library's default three streams
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/solo -o b -- $LONGBENCH --streams 3 ) > $OUT/solo.log 2>&1
( cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o b -- $PMCBENCH ) > $OUT/fetch.log 2>&1
( cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o b -- $PMCBENCH ) > $OUT/write.log 2>&1
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
    csv.writer(open(dst + "/kernel_stats_three_streams.csv", "w")).writerows(keep)
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
traffic["_meta"] = {"blocks_per_launch": int(sys.argv[3]), "config": sys.argv[4], "command": "bench.py --config %s --steps 2 --warmup 1 --batches 6 --cpu-seconds 0 --no-profile --no-latency --blocks %s --ring noise" % (sys.argv[4], sys.argv[3]),
                    "duration_passes": "bench.py --config %s --steps 8 --warmup 4 --batches 24 --cpu-seconds 0 --no-profile --no-latency --no-strong --blocks %s --ring noise (kernel_stats*.csv)" % (sys.argv[4], sys.argv[3])}
json.dump(traffic, open(dst + "/pmc_traffic.json", "w"), indent=1, sort_keys=True)
print(open(dst + "/kernel_stats.csv").read())
print(json.dumps(traffic, indent=1, sort_keys=True))
PY
for f in stats solo fetch write; do tail -3 $OUT/$f.log > $DST/$f.log.tail 2>/dev/null; done
if [ "${COLLECT_STATS_ONLY:-0}" = "1" ]; then tail -2 $OUT/stats.log; exit 0; fi
# SQ counters of every kernel (own passes, one stream: every kernel alone on the device)
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" ; do
  i=$((i+1))
  ( cd /tmp && CSDR_STREAMS=1 timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/sq$i -o b -- $PMCBENCH ) > $OUT/sq$i.log 2>&1
done
python - "$OUT" "$DST" <<'PY'
import csv, glob, json, sys, collections
out, dst = sys.argv[1], sys.argv[2]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/sq*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "csdr::" in k:
            res[k.split("(")[0].replace("void ", "").replace("csdr::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in res.items()}
for k, d in summ.items():
    wc = d.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        d["_wait_frac"] = d.get("SQ_WAIT_ANY", 0.0) / wc; d["_issue_stall_frac"] = d.get("SQ_WAIT_INST_ANY", 0.0) / wc; d["_active_frac"] = d.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
    if d.get("SQ_LDS_IDX_ACTIVE"):
        d["_lds_conflict_frac"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
json.dump(summ, open(dst + "/sq_summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps({k: {c: v for c, v in d.items() if c.startswith("_")} for k, d in summ.items()}, indent=1))
PY
tail -2 $OUT/stats.log
