#!/usr/bin/env python
"""Timeline of the last bench step from a rocprofv3 --kernel-trace CSV (kernels >= min_us)."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'tfc_edge_fwd_kernel' in r['Kernel_Name']]
seq = rows[idx[-2]:]
t0 = int(seq[0]['Start_Timestamp'])
tot = 0
for r in seq:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']); name = re.sub(r'^void ', '', name)
    m = re.match(r'([\w:]+(<[^(]*>)?)', name); name = (m.group(1) if m else name)[:70]
    if d >= min_us:
        print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {d:8.1f} {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):6d} {name}")
print('kernel time in step', round(tot, 1), 'span', (int(seq[-1]['End_Timestamp']) - t0) / 1e3, 'launches', len(seq))
