#!/bin/bash
# rocprofv3 kernel trace of a python script; prints the per-kernel stats and the launch sequence of the last step.
# usage: gpurun -- 'bash scripts/gpu_trace_script.sh tag script.py [env...]'
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/"$@" > $OUT/trace.log 2>&1; echo rc=$?
python - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last occurrence of head_fwd marks a step end; print the sequence of the last step (from the previous head_fwd)
idx = [i for i, n in enumerate(names) if "head_fwd_kernel" in n]
a, b = (idx[-2] + 1, idx[-1] + 1) if len(idx) >= 2 else (0, len(rows))
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    n = r["Kernel_Name"].replace("void tcr::", "").split("(")[0][:60]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us  {n}  grid {r.get('Grid_Size_X', '?')} wg {r.get('Workgroup_Size_X', '?')} lds {r.get('LDS_Block_Size', r.get('LDS_Block_Size_v', '?'))}")
PY
