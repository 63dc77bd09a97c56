#!/usr/bin/env python
"""Turn the rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; bench run + calibration run) into
per-launch HBM traffic of the conv kernels.  usage: pmc_traffic.py <dir with fetch/ write/ cal_fetch/ cal_write/> <out.json>"""
import csv, glob, json, os, re, sys
from collections import defaultdict

def load(path, counter):
    out = defaultdict(list)
    for h in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(h)):
            if r["Counter_Name"] != counter: continue
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); name = re.sub(r"^void ", "", name)
            m = re.match(r"([\w:]+)", name)
            out[(m.group(1), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return out

root, dst = sys.argv[1], sys.argv[2]
GiB = 1 << 30
cal_f, cal_w = load(os.path.join(root, "cal_fetch"), "FETCH_SIZE"), load(os.path.join(root, "cal_write"), "WRITE_SIZE")
def cal(tab, key):
    v = [x for (k, g), xs in tab.items() if key in k for x in xs]
    return sum(v) / len(v)
# counters are in KiB (rocprofv3 derived metric); factor = true bytes / reported bytes for the access shape
f_dword = GiB / (cal(cal_f, "tfc_gather_rows_kernel") * 1024)
w_dword = GiB / (cal(cal_w, "tfc_gather_rows_kernel") * 1024)
f_x4 = w_x4 = None  # (the 16-byte copy of the calibration run is informational only)
res = {"calibration": {"fetch_dword_per_lane": f_dword, "write_dword_per_lane": w_dword, "fetch_16B_per_lane": f_x4, "write_16B_per_lane": w_x4,
                       "note": "true bytes / (counter x 1024) on a 1 GiB once-through stream"}, "kernels": {}}
fe, wr = load(os.path.join(root, "fetch"), "FETCH_SIZE"), load(os.path.join(root, "write"), "WRITE_SIZE")
for kern in ("tfc_conv_fwd_kernel", "tfc_conv_bwd_kernel"):
    per_grid = {}
    for (k, g), xs in fe.items():
        if k != kern: continue
        ws = wr.get((k, g), [0.0])
        fbytes = sum(xs) / len(xs) * 1024 * f_dword
        wbytes = sum(ws) / len(ws) * 1024 * w_dword
        per_grid[str(g)] = {"fetch_bytes": fbytes, "write_bytes": wbytes, "launches_sampled": len(xs)}
    tot = sum(v["fetch_bytes"] + v["write_bytes"] for v in per_grid.values())
    res["kernels"][kern] = {"per_grid": per_grid, "avg_bytes_per_launch": tot / max(len(per_grid), 1)}
# the build the counters were collected on (bench.py puts the same hash on its line)
try:
    import hashlib
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matten_amd", "libtfc.so")
    res["lib_sha256_16"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
except OSError:
    res["lib_sha256_16"] = None
json.dump(res, open(dst, "w"), indent=1)
print(json.dumps(res, indent=1))
