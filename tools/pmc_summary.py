#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV: per kernel (short name, grid) the mean of each counter."""
import csv, glob, os, re, sys
from collections import defaultdict
path = sys.argv[1]
hits = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
agg = defaultdict(lambda: defaultdict(list))
for h in hits:
    for r in csv.DictReader(open(h)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); name = re.sub(r"^void ", "", name)
        m = re.match(r"([\w:]+(<[^(]*>)?)", name); name = (m.group(1) if m else name)[:70]
        key = (name, r.get("Grid_Size", ""))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k, c in agg.items():
    if flt and flt not in k[0]: continue
    print(k[0], "grid", k[1], "n", len(next(iter(c.values()))))
    for cn, v in sorted(c.items()):
        print(f"    {cn:32s} {sum(v)/len(v):16.1f}")
