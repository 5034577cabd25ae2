#!/usr/bin/env python
"""Condense a gpurun_out/<tag> rocprofv3 run (scripts/gpu_prof.sh) into small CSVs under profiles/.

usage: python scripts/summarize_prof.py gpurun_out/r01b profiles/r01
  -> profiles/r01_kernel_stats.csv   (rocprofv3 --kernel-trace --stats, names shortened)
  -> profiles/r01_pmc.csv            (per-kernel mean counter value per launch, one --pmc pass per group)
"""
import collections
import csv
import glob
import os
import re
import sys


def short(name: str) -> str:
    name = name.strip('"')
    m = re.match(r"(?:void )?(?:tcr::)?([A-Za-z0-9_]+(?:<[^>(]*>)?)", name)
    n = m.group(1) if m else name[:60]
    return n if "tcr::" in name else "other:" + n[:48]


def main(src, dst):
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    for sub, suffix in (("trace", "_kernel_stats.csv"), ("trace_fwd", "_fwd_kernel_stats.csv")):
        stats = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
        if not stats:
            continue
        rows = list(csv.DictReader(open(stats[0])))
        with open(dst + suffix, "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "percent"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"]])
        print("wrote", dst + suffix)
    agg = collections.OrderedDict()
    for f in sorted(glob.glob(os.path.join(src, "pmc*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            if "tcr::" not in r["Kernel_Name"]:
                continue
            k = (short(r["Kernel_Name"]), r["Counter_Name"], os.path.basename(os.path.dirname(f)))
            a = agg.setdefault(k, [0.0, 0])
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    with open(dst + "_pmc.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "counter", "pass", "mean_per_launch", "launches"])
        for (k, c, p), (s, n) in agg.items():
            w.writerow([k, c, p, f"{s / n:.1f}", n])
    print("wrote", dst + "_kernel_stats.csv", dst + "_pmc.csv")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
