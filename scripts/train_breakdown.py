#!/usr/bin/env python
"""Condense scripts/gpu_prof_train.sh output: one training step's launches (between two `sgd_momentum_kernel`s) by kernel, and
the HBM bytes per step from the FETCH_SIZE / WRITE_SIZE passes (gfx950: FETCH_SIZE counts 64 B units in KB/2 -> x2; see
MI355X_MICROARCH.md).

usage: python scripts/train_breakdown.py gpurun_out/<tag> profiles/r02_train
  -> <dst>_step_breakdown.txt, <dst>_pmc.csv
"""
import collections, csv, glob, os, re, sys


def short(name: str) -> str:
    name = name.strip('"')
    m = re.match(r"(?:void )?(?:tcr::)?([A-Za-z0-9_]+(?:<[^>(]*>)?)", name)
    return m.group(1) if m else name[:60]


def main(src, dst, marker="sgd_momentum_kernel"):
    lines = []
    f = glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True)
    if f:
        rows = list(csv.DictReader(open(f[0])))
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        names = [short(r["Kernel_Name"]) for r in rows]
        idx = [i for i, n in enumerate(names) if n.startswith(marker)]
        # median step of the run (by wall time between markers)
        spans = sorted(((int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"]), a, b) for a, b in zip(idx[:-1], idx[1:])))
        wall, a, b = spans[len(spans) // 2]
        agg, tot = collections.OrderedDict(), 0
        for i in range(a + 1, b + 1):
            d = int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])
            tot += d
            x = agg.setdefault(names[i], [0, 0])
            x[0] += d; x[1] += 1
        lines.append(f"# one training step (batch 4096, features precomputed; step = launches up to and including {marker}) under rocprofv3 --kernel-trace: the median step of the run")
        lines.append(f"step wall {wall / 1e3:.1f} us, kernel time {tot / 1e3:.1f} us, {b - a} launches")
        for k, (d, n) in sorted(agg.items(), key=lambda x: -x[1][0]):
            lines.append(f"{k:56s} {n:3d} {d / 1e3:9.1f}")
        steps = len(idx)
    else:
        steps = int(os.environ.get("STEPS", "8"))
    agg = collections.OrderedDict()
    for fn in sorted(glob.glob(os.path.join(src, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(fn)):
            if "tcr::" not in r["Kernel_Name"]:
                continue
            k = (short(r["Kernel_Name"]), r["Counter_Name"])
            x = agg.setdefault(k, [0.0, 0])
            x[0] += float(r["Counter_Value"]); x[1] += 1
    if agg:
        with open(dst + "_pmc.csv", "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["kernel", "counter", "sum_over_run", "launches", "per_step"])
            for (k, c), (s, n) in agg.items():
                w.writerow([k, c, f"{s:.1f}", n, f"{s / steps:.1f}"])
        fetch = sum(s for (k, c), (s, n) in agg.items() if c == "FETCH_SIZE") * 1024 * 2
        write = sum(s for (k, c), (s, n) in agg.items() if c == "WRITE_SIZE") * 1024
        lines.append(f"# HBM traffic per step over {steps} steps: FETCH_SIZE x 2 (gfx950) {fetch / steps / 1e6:.1f} MB + WRITE_SIZE {write / steps / 1e6:.1f} MB"
                     f" = {(fetch + write) / steps / 1e6:.1f} MB")
        per = collections.OrderedDict()
        for (k, c), (s, n) in agg.items():
            if c in ("FETCH_SIZE", "WRITE_SIZE"):
                per[k] = per.get(k, 0.0) + s * 1024 * (2 if c == "FETCH_SIZE" else 1)
        for k, v in sorted(per.items(), key=lambda x: -x[1])[:14]:
            lines.append(f"#   {k:54s} {v / steps / 1e6:9.1f} MB/step")
    open(dst + "_step_breakdown.txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "sgd_momentum_kernel")
