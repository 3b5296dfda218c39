#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
OUT=/tmp/pmc
mkdir -p $OUT gpurun_out
B="python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --no-profile"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_BRANCH" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/s$i -o p -- $B > $OUT/s$i.log 2>&1
  echo "set $i rc=$?"
done
python - <<'PY'
import csv, glob, collections, json
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "csdr::" not in k: continue
        k = k.split("(")[0].replace("csdr::", "")
        res[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in res.items()}
json.dump(out, open("gpurun_out/pmc_summary.json", "w"), indent=1)
for k, d in out.items():
    print(k, {c: round(v) for c, v in d.items()})
PY
ls /tmp/pmc/s1 | head; tail -3 /tmp/pmc/s1.log
