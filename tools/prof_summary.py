#!/usr/bin/env python
"""Condense a rocprofv3 output directory into small text summaries (kept under profiles/).

  python tools/prof_summary.py <rocprof_out_dir> <summary.txt>

* ``*kernel_stats.csv`` (from --kernel-trace --stats) is copied through (top 40 kernels);
* ``*kernel_trace.csv`` -> per-kernel calls / total / average duration (us);
* ``*counter_collection.csv`` (from --pmc) -> per-kernel sum and per-launch mean of every counter.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name, n=90):
    name = name.replace("(anonymous namespace)::", "")
    return name if len(name) <= n else name[: n - 3] + "..."


def main(src, dst):
    out = []
    for path in sorted(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)):
        out.append(f"== {os.path.relpath(path, src)} (rocprofv3 --kernel-trace --stats) ==")
        with open(path) as f:
            rows = list(csv.reader(f))
        for r in rows[:41]:
            out.append(" | ".join(short(c, 100) for c in r))
        out.append("")
    for path in sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)):
        agg = defaultdict(lambda: [0, 0.0])
        with open(path) as f:
            for r in csv.DictReader(f):
                try:
                    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
                except Exception:
                    continue
                a = agg[r.get("Kernel_Name", "?")]
                a[0] += 1
                a[1] += d
        tot = sum(v[1] for v in agg.values()) or 1.0
        out.append(f"== {os.path.relpath(path, src)}: per-kernel durations ==")
        out.append(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'share':>7}  kernel")
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
            out.append(f"{v[0]:7d} {v[1]:12.1f} {v[1] / v[0]:10.2f} {100 * v[1] / tot:6.2f}%  {short(k)}")
        out.append("")
    for path in sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)):
        agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        with open(path) as f:
            for r in csv.DictReader(f):
                try:
                    v = float(r["Counter_Value"])
                except Exception:
                    continue
                a = agg[r.get("Kernel_Name", "?")][r.get("Counter_Name", "?")]
                a[0] += 1
                a[1] += v
        out.append(f"== {os.path.relpath(path, src)}: counters per kernel (sum over launches, mean per launch) ==")
        for k, cs in sorted(agg.items(), key=lambda kv: -sum(c[1] for c in kv[1].values()))[:40]:
            for cn, (n, s) in sorted(cs.items()):
                out.append(f"{cn:>16} launches={n:6d} sum={s:16.1f} mean={s / max(n, 1):14.2f}  {short(k)}")
        out.append("")
    os.makedirs(os.path.dirname(os.path.abspath(dst)), exist_ok=True)
    with open(dst, "w") as f:
        f.write("\n".join(out) + "\n")
    print(f"wrote {dst} ({len(out)} lines)")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
