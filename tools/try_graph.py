import faulthandler; faulthandler.enable()
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import matten_amd.nn as H
from matten_amd import keys as K, _lib
from matten_amd.harness import ConvStack
from matten_amd.graphs import GraphedStep
from matten_amd.dp import FlatGradBucket

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
dev = torch.device("cuda")
batch = bench.build_batch(wl, 0, dev)
torch.manual_seed(35)
model = ConvStack(H, wl["emb"], wl["lmax"], wl["irreps"], wl["layers"], 64, 2, 5.0, 28.0).to(dev)
if os.environ.get("NO_WGRAD"):
    for n_, p_ in model.named_parameters():
        if any(k in n_ for k in ("linear_1", "linear_2", "self_connection")):
            p_.requires_grad_(False)
params = [p for p in model.parameters() if p.requires_grad]
bucket = FlatGradBucket(params)
pos0 = batch[K.POSITIONS]

def step():
    bucket.zero()
    b = dict(batch); pos = pos0.detach().requires_grad_(True); b[K.POSITIONS] = pos
    _, e, _ = model(b, wl["graphs"])
    e.sum().backward()
    return pos.grad, e.detach(), torch.cat([p.grad.reshape(-1) for p in params])

for _ in range(3): g_ref, e_ref, flat_ref = step()
torch.cuda.synchronize()
flat_ref = flat_ref.clone(); g_ref = g_ref.clone(); e_ref = e_ref.clone()
L = _lib.lib()
L.tfc_profile_reset(); L.tfc_profile_enable(int(os.environ.get("PROF","0")))
gs = GraphedStep(step)
L.tfc_profile_enable(0)
print("captured")
for _ in range(5): out = gs()
torch.cuda.synchronize()
print("equal grads:", torch.equal(out[0], g_ref), torch.equal(out[2], flat_ref), torch.equal(out[1], e_ref))
t0 = time.perf_counter()
for _ in range(30): gs()
torch.cuda.synchronize()
print("graph replay ms/step", (time.perf_counter() - t0) / 30 * 1e3)
print(_lib.profile_report())
t0 = time.perf_counter()
for _ in range(30): step()
torch.cuda.synchronize()
print("eager ms/step", (time.perf_counter() - t0) / 30 * 1e3)
