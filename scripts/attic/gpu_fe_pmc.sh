#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-fepmc}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for VAR in ${2:-2 4}; do
 i=0
 for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  FE_VAR=$VAR timeout 200 rocprofv3 --pmc $PMC --output-format csv -d $OUT/v${VAR}_p$i -o p -- python $R/scripts/fe_only.py > $OUT/v${VAR}_p$i.log 2>&1; echo "var $VAR pass $i rc=$?"
 done
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/v*_p*/p_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "frontend_" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f.split("/")[-2], {k: round(sum(v)/len(v)) for k, v in agg.items()})
PY
