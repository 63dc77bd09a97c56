import faulthandler; faulthandler.enable()
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import matten_amd.nn as H
from matten_amd import keys as K, _lib, ops
from matten_amd.harness import ConvStack
from matten_amd.graphs import GraphedStep

variant = sys.argv[1]
wl = bench.WORKLOADS["c3"]
dev = torch.device("cuda")
if variant == "toy":
    m = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.Tanh(), torch.nn.Linear(64, 1)).to(dev)
    x = torch.randn(128, 64, device=dev)
    def step():
        for p in m.parameters(): p.grad = None
        m(x).sum().backward()
    gs = GraphedStep(step); gs(); torch.cuda.synchronize(); print("toy ok"); sys.exit(0)
batch = bench.build_batch(wl, 0, dev)
torch.manual_seed(35)
model = ConvStack(H, wl["emb"], wl["lmax"], wl["irreps"], wl["layers"], 64, 2, 5.0, 28.0).to(dev)
pos0 = batch[K.POSITIONS]
if variant == "fwd_nograd":
    def step():
        with torch.no_grad():
            return model(dict(batch), wl["graphs"])[1]
elif variant == "fwd_grad":
    def step():
        b = dict(batch); pos = pos0.detach().requires_grad_(True); b[K.POSITIONS] = pos
        return model(b, wl["graphs"])[1]
elif variant == "encoders":
    def step():
        with torch.no_grad():
            b = dict(batch)
            return model.radial_basis(model.spharm_edges(b))[K.EDGE_EMBEDDING]
elif variant == "linear":
    lin = model.convs[1].message_fn.linear_1
    x = torch.randn(2048, lin.spec.d_in, device=dev)
    def step():
        with torch.no_grad():
            return lin(x)
elif variant == "full":
    def step():
        for p in model.parameters(): p.grad = None
        b = dict(batch); pos = pos0.detach().requires_grad_(True); b[K.POSITIONS] = pos
        model(b, wl["graphs"])[1].sum().backward()
        return pos.grad
if variant == "full_bucket":
    from matten_amd.dp import FlatGradBucket
    bucket = FlatGradBucket(list(model.parameters()))
    def step():
        bucket.zero()
        b = dict(batch); pos = pos0.detach().requires_grad_(True); b[K.POSITIONS] = pos
        model(b, wl["graphs"])[1].sum().backward()
        return pos.grad
if variant == "full_eagerfirst_s":
    def step():
        for p in model.parameters(): p.grad = None
        b = dict(batch); pos = pos0.detach().requires_grad_(True); b[K.POSITIONS] = pos
        model(b, wl["graphs"])[1].sum().backward()
        return pos.grad
    s0 = torch.cuda.Stream()
    with torch.cuda.stream(s0):
        for _ in range(3): step()
    torch.cuda.synchronize()
if variant == "full_eagerfirst_nocache":
    def step():
        for p in model.parameters(): p.grad = None
        b = dict(batch); pos = pos0.detach().requires_grad_(True); b[K.POSITIONS] = pos
        model(b, wl["graphs"])[1].sum().backward()
        return pos.grad
    for _ in range(3): step()
    torch.cuda.synchronize()
    ops._TRANSPOSED.clear(); ops._ATTR_FACTORS.clear(); ops._ATTR_CLASSES.clear(); ops._ORDER_CACHE.clear(); ops._GEMM_WS.clear()
    from matten_amd.nn import point_conv
    point_conv.reset_prefetch_state()
if variant == "full_eagerfirst":
    def step():
        for p in model.parameters(): p.grad = None
        b = dict(batch); pos = pos0.detach().requires_grad_(True); b[K.POSITIONS] = pos
        model(b, wl["graphs"])[1].sum().backward()
        return pos.grad
    for _ in range(3): step()
    torch.cuda.synchronize()
gs = GraphedStep(step)
out = gs(); torch.cuda.synchronize()
print(variant, "ok")
