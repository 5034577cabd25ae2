#!/bin/bash
# rocprofv3 PMC passes with a caller-chosen counter list per pass.  usage: gpurun -- 'bash scripts/gpu_pmc2.sh tag script.py "C1 C2" "C3 C4" ...'
TAG=$1; shift
SCRIPT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for PMC in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -o p -- python $R/$SCRIPT > $OUT/pmc$i.log 2>&1; echo "pass $i ($PMC) rc=$?"
done
python - <<PY
import csv, glob, collections
agg = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:70], r["Grid_Size"], r["Counter_Name"])
        a = agg.setdefault(k, [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, g, c), (s, n) in agg.items():
    if "tcr" in k: print(f"{k:70s} grid {g:>8s} {c:32s} {s/n:16.1f} x{n}")
PY
