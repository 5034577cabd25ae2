#!/bin/bash
# Kernel trace of the multi-stream inference pipeline (scripts/pipe_only.py): per-kernel stats + the launches of a stretch of steady state;
# then ONE counter pass of the same run (counter collection serialises the dispatches: its kernel durations say so).
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
    o.write("# inference pipeline (whole batches alternating over three streams), batch 4096: launches of 8 consecutive steps in steady state (us; queue id)\n")
    for r in rows[mid:mid + 16]:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        o.write(f"{s:9.1f} {e:9.1f} {e - s:7.1f} q={r['Queue_Id']} {r['Kernel_Name'][:60]}\n")
    fe = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[20:] if "frontend_pk" in r["Kernel_Name"]]
    nt = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[20:] if "fused" in r["Kernel_Name"]]
    starts = [int(r["Start_Timestamp"]) for r in rows[20:] if "frontend_pk" in r["Kernel_Name"]]
    o.write(f"# front-end kernel avg {sum(fe) / len(fe) / 1e3:.1f} us, network kernel avg {sum(nt) / len(nt) / 1e3:.1f} us, step (front-end start to start) avg {(starts[-1] - starts[0]) / (len(starts) - 1) / 1e3:.1f} us\n")
print(open("$OUT/summary_timeline.txt").read())
PY

timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc -o p -- python $R/scripts/pipe_only.py > $OUT/pmc.log 2>&1; echo "pmc rc=$?"
python - <<PY
import csv, glob, collections
ct = glob.glob("$OUT/pmc/*counter_collection.csv")
kt = glob.glob("$OUT/pmc/*kernel_trace.csv")
with open("$OUT/summary_pmc.txt", "w") as o:
    if kt:
        rows = list(csv.DictReader(open(kt[0])))
        for key in ("frontend_pk", "fused"):
            d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if key in r["Kernel_Name"]][20:]
            if d: o.write(f"# counter pass: {key} kernel avg {sum(d) / len(d) / 1e3:.1f} us over {len(d)} launches\n")
    if ct:
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(ct[0])):
            if "tcr::" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"][:50], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            o.write(f"{k:52s} {c:28s} {sum(v) / len(v):16.1f} x{len(v)}\n")
print(open("$OUT/summary_pmc.txt").read())
PY
