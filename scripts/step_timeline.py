#!/usr/bin/env python
"""Timeline of the median step out of a rocprofv3 kernel trace (csv): start / end / duration (us) and queue of every launch.

usage: python scripts/step_timeline.py gpurun_out/<tag>/trace/t_kernel_trace.csv <marker-kernel> [from_us]
"""
import csv, re, sys


def short(n):
    m = re.match(r"(?:void )?(?:tcr::)?([A-Za-z0-9_]+(?:<[^>(]*>)?)", n)
    return m.group(1) if m else n[:40]


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
spans = sorted((int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"]), a, b) for a, b in zip(idx[:-1], idx[1:]))
_, a, b = spans[len(spans) // 2]
t0 = int(rows[a]["End_Timestamp"])
lo = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
for r in rows[a + 1:b + 1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    if s >= lo:
        print(f"{s:9.1f} {e:9.1f} {e - s:8.1f} q={r['Queue_Id']} {short(r['Kernel_Name'])}")
