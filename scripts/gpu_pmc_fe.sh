#!/bin/bash
# rocprofv3 PMC passes of the front-end alone (scripts/fe_only.py; KNOB23=1: the two-waves kernel).  usage: gpurun -- 'bash scripts/gpu_pmc_fe.sh tag'
TAG=${1:-pmc_fe}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -o p -- python $R/scripts/fe_only.py > $OUT/pmc$i.log 2>&1; echo "pass $i rc=$?"
done
cd $R
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
agg = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(sys.argv[1], "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if "frontend" not in r["Kernel_Name"]: continue
        k = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
        a = agg.setdefault(k, [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
with open(os.path.join(sys.argv[1], "summary.txt"), "w") as fh:
    for (k, c), (s, n) in agg.items():
        line = f"{k:42s} {c:32s} {s / n:16.1f} x{n}"
        print(line); fh.write(line + "\n")
PY
