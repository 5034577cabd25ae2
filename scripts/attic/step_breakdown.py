#!/usr/bin/env python
"""Per-kernel time of ONE step out of a rocprofv3 kernel trace (sqlite .db written by scripts/gpu_check.sh).

usage: python scripts/step_breakdown.py gpurun_out/<tag>/prof/trace_results.db <marker-kernel-prefix> [which]
The step is the span between two consecutive launches of the marker kernel (e.g. adam_kernel for the DS-CNN
training step, sgd_momentum_kernel for TC-ResNet training); `which` picks the marker occurrence (default: last).
"""
import collections
import re
import sqlite3
import sys


def short(n: str) -> str:
    m = re.match(r"_ZN3tcr(\d+)", n)
    if not m:
        return n[:40]
    ln, st = int(m.group(1)), m.end()
    base, rest = n[st:st + ln], n[st + ln:]
    t = re.match(r"I((?:L[ib]\d+E)+)E", rest)
    if t:
        base += "<" + ",".join(re.findall(r"L[ib](\d+)E", t.group(1))) + ">"
    return base


def main(db_path, marker, which=-1):
    cur = sqlite3.connect(db_path).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
    names = [short(r[0]) for r in rows]
    idx = [i for i, n in enumerate(names) if n.startswith(marker)]
    a, b = idx[which - 1], idx[which]
    agg, tot = collections.OrderedDict(), 0
    for i in range(a + 1, b + 1):
        d = rows[i][2] - rows[i][1]
        tot += d
        x = agg.setdefault(names[i], [0, 0])
        x[0] += d
        x[1] += 1
    print(f"step wall {(rows[b][2] - rows[a][2]) / 1e3:.1f} us, kernel time {tot / 1e3:.1f} us, {b - a} launches")
    for k, (d, n) in sorted(agg.items(), key=lambda x: -x[1][0]):
        print(f"{k:50s} {n:3d} {d / 1e3:9.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else -1)
