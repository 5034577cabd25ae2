#!/bin/bash
# kernel trace of a python target; per-kernel means -> gpurun_out/<tag>/summary.txt
# usage: gpurun -- 'TUNE=... bash scripts/gpu_trace_any.sh tag scripts/target.py'
TAG=$1; TARGET=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/$TARGET > $OUT/trace.log 2>&1; echo "trace rc=$?"
cd $R
python - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
with open(os.path.join(out, "summary.txt"), "w") as fh:
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            line = f"{r['Name'].split('(')[0].replace('void tcr::','')[:60]:62s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}"
            fh.write(line + "\n")
            if float(r["Percentage"]) > 0.8: print(line)
PY
