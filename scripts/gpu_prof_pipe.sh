#!/bin/bash
# Kernel trace of the two-stream inference pipeline (scripts/pipe_only.py): per-kernel stats + the launches of a stretch of steady state.
TAG=${1:-r03pipe}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/scripts/pipe_only.py > $OUT/trace.log 2>&1; echo "trace rc=$?"
python - <<PY
import csv, glob
f = glob.glob("$OUT/trace/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "frontend_pk" in r["Kernel_Name"] or "fused" in r["Kernel_Name"]]
mid = len(rows) // 2
t0 = int(rows[mid]["Start_Timestamp"])
with open("$OUT/summary_timeline.txt", "w") as o:
    o.write("# two-stream inference pipeline, three batches deep, batch 4096: launches of 8 consecutive steps in steady state (us; queue id)\n")
    for r in rows[mid:mid + 16]:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        o.write(f"{s:9.1f} {e:9.1f} {e - s:7.1f} q={r['Queue_Id']} {r['Kernel_Name'][:60]}\n")
    fe = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[20:] if "frontend_pk" in r["Kernel_Name"]]
    nt = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[20:] if "fused" in r["Kernel_Name"]]
    starts = [int(r["Start_Timestamp"]) for r in rows[20:] if "frontend_pk" in r["Kernel_Name"]]
    o.write(f"# front-end kernel avg {sum(fe) / len(fe) / 1e3:.1f} us, network kernel avg {sum(nt) / len(nt) / 1e3:.1f} us, step (front-end start to start) avg {(starts[-1] - starts[0]) / (len(starts) - 1) / 1e3:.1f} us\n")
print(open("$OUT/summary_timeline.txt").read())
PY
