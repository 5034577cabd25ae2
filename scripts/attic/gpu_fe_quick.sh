#!/bin/bash
# front-end kernel alone: rocprofv3 kernel-trace average + the LDS PMC pass.  usage: gpurun -- 'bash scripts/gpu_fe_quick.sh tag'
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for cfg in "640 320" "480 160"; do
  set -- $cfg
  WIN=$1 HOP=$2 STEPS=200 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$1 -o t -- python $R/scripts/fe_only.py > $OUT/t$1.log 2>&1
  f=$(find $OUT/t$1 -name '*kernel_stats.csv' | head -1); grep frontend $f | cut -d, -f1-4 | cut -c1-120
  WIN=$1 HOP=$2 STEPS=20 timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/p$1 -o p -- python $R/scripts/fe_only.py > $OUT/p$1.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/p$1/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "frontend" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$1/$2", {k: round(sum(v) / len(v) / 1e6, 2) for k, v in acc.items()}, "(M per launch)")
PY
done
