#!/usr/bin/env python
"""Summarise tools/pmc_conv.sh (rocprofv3 --pmc passes over tools/bench_conv.py): per conv kernel and
grid size, where the waves' cycles go and how the caches behave.  usage: pmc_conv_summary.py [dir]"""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_conv"
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/*/vm/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        kn = "fwd" if "tfc_conv_fwd" in k else ("bwd" if "tfc_conv_bwd" in k else None)
        if kn:
            agg[(kn, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in sorted(agg):
    d = {c: sum(v) / len(v) for c, v in agg[key].items()}
    wc, nw = d["SQ_WAVE_CYCLES"], d["SQ_WAVES"]
    print(f"tfc_conv_{key[0]}_kernel grid {key[1]} ({nw:.0f} waves)")
    print("   share of wave cycles: waiting on a counter %.2f | issuing %.2f (valu %.3f, scalar %.3f)" % (
        d["SQ_WAIT_ANY"] / wc, d["SQ_ACTIVE_INST_ANY"] / wc, d["SQ_ACTIVE_INST_VALU"] / wc, d["SQ_ACTIVE_INST_SCA"] / wc))
    print("   instructions per wave: valu %.0f  salu %.0f  smem %.0f  vmem_rd %.0f ;  cycles per wave %.0f ; resident waves per busy SQ cycle %.1f" % (
        d["SQ_INSTS_VALU"] / nw, d["SQ_INSTS_SALU"] / nw, d["SQ_INSTS_SMEM"] / nw, d["SQ_INSTS_VMEM_RD"] / nw, wc / nw, wc / d["SQ_BUSY_CYCLES"]))
    print("   L2: hit rate %.3f of %.2f M requests, %.2f M reads to the fabric ; L1 -> L2 read requests %.2f M for %.2f M wave loads" % (
        d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"]), d["TCC_REQ_sum"] / 1e6, d["TCC_EA0_RDREQ_sum"] / 1e6,
        d["TCP_TCC_READ_REQ_sum"] / 1e6, d["SQ_INSTS_VMEM_RD"] / 1e6))
