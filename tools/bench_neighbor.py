"""Times the GPU neighbour list (tfc_nl_count + tfc_nl_fill + tfc_nl_ordering) on the bench
configurations' cells, next to the host list (numpy, fp64) on the same structures and next to the
sort-based tfc_csr_build it replaces.  usage: python tools/bench_neighbor.py [c3|c4|c5] [--host]
Prints one JSON line per configuration.  The kernels are bound by fp64 vector arithmetic and
latency (one distance test per surviving (i, j, image) candidate), not by HBM: the positions and
image vectors of a structure stay in L1/L2.  `candidates` counts the (i, j, image) triples before the
per-node image pruning, i.e. what the host implementation tests."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from matten_amd import _lib, ops  # noqa: E402
from matten_amd.data.neighbor import image_table, neighbor_list  # noqa: E402
from matten_amd.data.neighbor_gpu import neighbor_list_batch  # noqa: E402

CFG = {"c3": (4, 2.5, 32), "c4": (6, 2.5, 32), "c5": (12, 2.1, 1)}


def structures(n, a, batch):
    out = []
    for b in range(batch):
        g = np.random.default_rng(1000 + b)
        grid = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float64) * a
        out.append(dict(pos=grid + g.normal(0.0, 0.05, size=grid.shape), cell=np.eye(3) * (n * a)))
    return out


def main():
    names = [a for a in sys.argv[1:] if a in CFG] or ["c3", "c4", "c5"]
    host = "--host" in sys.argv
    dev = torch.device("cuda:0")
    for name in names:
        n, a, batch = CFG[name]
        st = structures(n, a, batch)
        cand = sum(len(s["pos"]) ** 2 * len(image_table(s["pos"], s["cell"], (True,) * 3, 5.0)) for s in st)
        neighbor_list_batch(st, 5.0, dev)  # warm-up
        torch.cuda.synchronize()
        _lib.lib().tfc_profile_enable(1)
        _lib.lib().tfc_profile_reset()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            out = neighbor_list_batch(st, 5.0, dev)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
        rep = _lib.profile_report()
        _lib.lib().tfc_profile_enable(0)
        k = {kk: v[1] / v[0] for kk, v in rep.items() if kk.startswith("tfc_nl")}
        E = out["edge_index"].shape[1]
        N = int(out["ptr"][-1])
        # the sort-based ordering on the same list
        ei = out["edge_index"].clone()
        ops.EdgeOrdering(ei, N)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ops.EdgeOrdering(ei, N)
        torch.cuda.synchronize()
        t_sort = (time.perf_counter() - t0) / reps
        line = {
            "workload": name, "structures": batch, "nodes": N, "edges": E, "candidates": cand,
            "gpu_wall_ms_incl_host_tables_and_sync": wall * 1e3,
            "kernel_ms": k,
            "edges_per_s_kernels_only": E / (sum(k.values()) * 1e-3),
            "sort_based_ordering_wall_ms": t_sort * 1e3,
        }
        if host:
            t0 = time.perf_counter()
            for s in st[: max(1, min(4, batch))]:
                neighbor_list(s["pos"], s["cell"], True, 5.0)
            line["host_numpy_ms_per_structure"] = (time.perf_counter() - t0) / max(1, min(4, batch)) * 1e3
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
