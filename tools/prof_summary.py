#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid size) launches, mean/min/max us and
share of the total GPU time.  usage: prof_summary.py <dir or *_kernel_trace.csv> [skip_first_n_steps_fraction]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(<[^(]*>)?)", name)
    name = m.group(1) if m else name
    return name[:90]


def main():
    path = sys.argv[1]
    if os.path.isdir(path):
        hits = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
        path = hits[0]
    rows = list(csv.DictReader(open(path)))
    agg = defaultdict(list)
    for r in rows:
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
        wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
        agg[(short(r["Kernel_Name"]), grid // max(wg, 1), r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""))].append(dur)
    total = sum(sum(v) for v in agg.values())
    print(f"{'kernel':90s} {'blocks':>7s} {'vgpr':>5s} {'n':>5s} {'mean_us':>9s} {'min':>8s} {'max':>8s} {'share':>6s}")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k[0]:90s} {k[1]:7d} {k[2]:>3s}+{k[3]:<2s} {len(v):5d} {sum(v)/len(v):9.1f} {min(v):8.1f} {max(v):8.1f} {100*sum(v)/total:5.1f}%")
    print(f"total GPU time {total/1e3:.2f} ms over {len(rows)} launches")


if __name__ == "__main__":
    main()
