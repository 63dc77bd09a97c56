"""tools/kern_table.py <bench json>...: per-tag launches per step, average and total ms of the eager breakdown pass."""
import json
import sys

for f in sys.argv[1:]:
    d = json.load(open(f))
    k = d["kernels"]
    rows = sorted(((v["launches_per_step"] * v["avg_ms"], t, v) for t, v in k.items()), reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"{f}: {d['ms_per_step']:.3f} ms/step ({d['config']['launch']}), sum of tagged kernels {tot:.3f} ms")
    for ms, t, v in rows:
        print(f"  {t:28s} {v['launches_per_step']:6.1f} x {v['avg_ms'] * 1e3:9.1f} us = {ms:7.3f} ms  {100 * ms / tot:5.1f} %")
