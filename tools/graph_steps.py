#!/usr/bin/env python
"""Find the graph-replayed steps in a trace of tools/try_graph.py and print the timeline of one of them."""
import csv, glob, subprocess, sys
d = sys.argv[1]
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'tfc_edge_fwd_kernel' in r['Kernel_Name']]
starts = idx[0::2]
spans = [(int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3 for a, b in zip(starts[:-1], starts[1:])]
best = min(range(len(spans)), key=lambda i: spans[i])
print("steps", len(spans), "shortest span", spans[best], "at", best)
subprocess.run([sys.executable, __file__.replace("graph_steps.py", "step_timeline2.py"), d, str(len(starts) - best)])
