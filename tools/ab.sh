#!/bin/bash
# tools/ab.sh <tag> <workload> [ENV=VAL ...]: one bench.py run (no CPU baseline) under the given environment,
# JSON line to gpurun_out/r03/bench_<workload>_<tag>.json; prints ms_per_step and the dense / conv fractions.
set -e
tag=$1; wl=$2; shift 2
mkdir -p gpurun_out/r03
steps=20; warm=5
if [ "$wl" = c4 ]; then steps=5; warm=2; fi
env "$@" timeout -k 10 400 python bench.py --no-cpu-baseline --workload $wl --steps $steps --warmup $warm \
    > gpurun_out/r03/bench_${wl}_${tag}.json 2> gpurun_out/r03/bench_${wl}_${tag}.err
python - <<PY
import json
d = json.load(open("gpurun_out/r03/bench_${wl}_${tag}.json"))
r = d["roofline"]; m = d["mfma_roofline"]
print("${tag} ${wl}: %.3f ms  bwd %.3f fwd %.3f  mlp %.3f lin %.3f" % (d["ms_per_step"], r["frac"], r["fwd_kernel"]["frac"],
      m["radial_mlp"]["frac"], m["o3_linear"]["frac"]), {k: v for k, v in m.items() if k not in ("radial_mlp", "o3_linear") and v} and "")
PY
