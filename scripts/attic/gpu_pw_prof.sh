R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pw; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for K in 0 3; do
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$K -o t -- python $R/scripts/dscnn_eval_only.py $K > $OUT/t$K.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$OUT/t$K/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]: print("$K", r["Name"][:70], r["Calls"], r["AverageNs"])
PY
done
timeout 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/p0 -o p -- python $R/scripts/dscnn_eval_only.py 0 > $OUT/p0.log 2>&1
timeout 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/p1 -o p -- python $R/scripts/dscnn_eval_only.py 0 > $OUT/p1.log 2>&1
python - <<PY
import csv,glob,collections
for d in ("p0","p1"):
    agg=collections.defaultdict(lambda:[0.0,0])
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%d,recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv1x1" in r["Kernel_Name"]:
                a=agg[(r["Kernel_Name"][:40],r["Counter_Name"])]; a[0]+=float(r["Counter_Value"]); a[1]+=1
    for k,v in sorted(agg.items()): print(k, v[0]/v[1])
PY
