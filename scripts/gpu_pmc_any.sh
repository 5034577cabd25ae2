#!/bin/bash
# kernel trace + PMC passes (issue / wait / LDS / MFMA counters) of a python target; per-kernel means -> gpurun_out/<tag>/summary.txt
# usage: gpurun -- 'bash scripts/gpu_pmc_any.sh tag scripts/target.py [kernel-substring]'
TAG=$1; TARGET=$2; FILTER=${3:-tcr::}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/$TARGET > $OUT/trace.log 2>&1; echo "trace rc=$?"
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -o p -- python $R/$TARGET > $OUT/pmc$i.log 2>&1; echo "pass $i rc=$?"
done
cd $R
python - "$OUT" "$FILTER" <<'PY'
import csv, glob, os, sys, collections
out, flt = sys.argv[1], sys.argv[2]
agg = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if flt not in r["Kernel_Name"]: continue
        k = (r["Kernel_Name"].split("(")[0].replace("void tcr::", "")[:44], r["Counter_Name"])
        a = agg.setdefault(k, [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
with open(os.path.join(out, "summary.txt"), "w") as fh:
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            line = f"{r['Name'].split('(')[0].replace('void tcr::','')[:60]:62s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}"
            fh.write(line + "\n")
            if float(r["Percentage"]) > 1.5: print(line)
    for (k, c), (s, n) in agg.items():
        line = f"{k:46s} {c:30s} {s / n:18.1f} x{n}"
        fh.write(line + "\n")
        print(line)
PY
