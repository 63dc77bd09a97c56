#!/usr/bin/env python
"""Timeline of one bench step from a rocprofv3 --kernel-trace CSV with the hardware queue of every kernel,
gaps on each queue, and how long two queues were busy at the same time.  usage: step_timeline2.py <dir> [step_from_end]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'tfc_edge_fwd_kernel' in r['Kernel_Name']]
# a step starts at every second edge_fwd launch (two per step)
starts = idx[0::2]
a, b = starts[-back], starts[-back + 1]
seq = rows[a:b]
t0 = int(seq[0]['Start_Timestamp'])
busy = {}
for r in seq:
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    q = r['Queue_Id']
    name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']); name = re.sub(r'^void ', '', name)
    m = re.match(r'([\w:]+(<[^(]*>)?)', name); name = (m.group(1) if m else name)[:60]
    busy.setdefault(q, []).append((s, e))
    print(f"q{q} {s:9.1f} {e:9.1f} {e-s:8.1f} {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):6d} {name}")
span = max(e for v in busy.values() for _, e in v)
print("step span", round(span, 1), "us")
for q, v in busy.items():
    print(f"queue {q}: {len(v)} kernels, busy {sum(e - s for s, e in v):.1f} us")
