#!/bin/bash
# SQ / traffic counters of the channelizer kernel alone (measurement helper): usage profiles/chan_counters.sh TAG CASE [env assignments...]
# Runs profiles/chan_bench.py CASE under rocprofv3 (separate passes: kernel trace, FETCH_SIZE, WRITE_SIZE, three SQ sets) and writes
# gpurun_out/chanctr_<TAG>_<CASE>.json
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=$1; CASE=$2; shift 2
for kv in "$@"; do export "$kv"; done
REPO=$(pwd)
OUT=/tmp/chanctr_${TAG}_$CASE
mkdir -p $OUT gpurun_out
export CHAN_BENCH_BASE=0 CHAN_BENCH_ITERS=10
CMD="python $REPO/profiles/chan_bench.py $CASE"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- $CMD ) > $OUT/stats.log 2>&1
( cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o b -- $CMD ) > $OUT/fetch.log 2>&1
( cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o b -- $CMD ) > $OUT/write.log 2>&1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE GRBM_COUNT" ; do
  i=$((i+1))
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/sq$i -o b -- $CMD ) > $OUT/sq$i.log 2>&1
done
python - "$OUT" "$REPO/gpurun_out/chanctr_${TAG}_$CASE.json" <<'PY'
import csv, glob, json, sys, collections
out, dst = sys.argv[1], sys.argv[2]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "chan_analyze" in k:
            res[k.split("(")[0].replace("void ", "").replace("csdr::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in res.items()}
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "chan_analyze" in r["Name"]:
            summ.setdefault(r["Name"].split("(")[0].replace("void ", "").replace("csdr::", ""), {})["_avg_ns"] = float(r["AverageNs"])
for k, d in summ.items():
    wc = d.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        d["_wait_frac"] = d.get("SQ_WAIT_ANY", 0.0) / wc; d["_issue_stall_frac"] = d.get("SQ_WAIT_INST_ANY", 0.0) / wc; d["_active_frac"] = d.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
        d["_valu_frac"] = d.get("SQ_ACTIVE_INST_VALU", 0.0) / wc; d["_lds_frac"] = d.get("SQ_ACTIVE_INST_LDS", 0.0) / wc
    if d.get("SQ_LDS_IDX_ACTIVE"):
        d["_lds_conflict_frac"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
json.dump(summ, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps(summ, indent=1, sort_keys=True))
PY
tail -2 $OUT/stats.log
