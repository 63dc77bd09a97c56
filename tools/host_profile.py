"""cProfile of the eager step's host side (config 3): which Python functions the ~3 ms of enqueue time go to.
usage: python tools/host_profile.py [n_steps]"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from matten_amd import harness as H  # noqa: E402
from matten_amd import keys as K  # noqa: E402

wl = bench.WORKLOADS["c3"]
dev = torch.device("cuda:0")
batch = bench.build_batch(wl, 0, dev)
torch.manual_seed(35)
model = H.ConvStack(__import__("matten_amd.nn", fromlist=["x"]), wl["emb"], wl["lmax"], wl["irreps"], wl["layers"], 64, 2, bench.R_CUT, 28.9).to(dev)
params = list(model.parameters())


def step():
    for p in params:
        p.grad = None
    b = dict(batch)
    pos = batch[K.POSITIONS].detach().requires_grad_(True)
    b[K.POSITIONS] = pos
    _, energy, _ = model(b, wl["graphs"])
    energy.sum().backward()


for _ in range(5):
    step()
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
