"""Per-tag kernel times of a bench line (the in-library event pass): python tools/kern_tags.py <bench.json> [substring]"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernels"]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
tot = sum(v["launches_per_step"] * v["avg_ms"] for v in k.values())
print(f"step {d['ms_per_step']:.3f} ms, sum of kernels {tot:.3f} ms")
for t, v in sorted(k.items(), key=lambda kv: -kv[1]["launches_per_step"] * kv[1]["avg_ms"]):
    if sub in t:
        print(f"  {t:34s} {v['launches_per_step']:5.1f} x {v['avg_ms'] * 1e3:9.1f} us = {v['launches_per_step'] * v['avg_ms']:.3f} ms")
