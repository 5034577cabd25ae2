cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for W in 0 1 4 7; do
  export TCR_FE_WHATIF=$W
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/fepmc$W -o p -- python $R/scripts/fe_only.py > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
agg=collections.OrderedDict()
for f in glob.glob("$R/gpurun_out/fepmc$W/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "frontend_pk" in r["Kernel_Name"]:
            a=agg.setdefault(r["Counter_Name"],[0,0]); a[0]+=float(r["Counter_Value"]); a[1]+=1
print("whatif $W:", {k: round(v[0]/v[1]/1e6,2) for k,v in agg.items()})
PY
done
