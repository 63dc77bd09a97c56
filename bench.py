#!/usr/bin/env python
"""bench.py - conv edges/sec, forward + backward, of the tensor-field convolution hot path.

    python bench.py --gpus N --steps K --warmup W [--workload c3|c4|c5]
    (N > 1 either under the launcher, python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...
    bench.py --gpus N ..., or directly: the parent then starts that launcher itself as a child process,
    before anything touches a GPU, and exits with its status)

One step = one pass of the hot path over one synthetic batch, inputs already resident in HBM:
edge encoders (spherical harmonics + Bessel) -> 3 x PointConvMessagePassing -> energy, then the
backward that yields every parameter gradient AND the forces (-dE/dpos), then (N > 1) one flat
gradient all-reduce over RCCL.  Work is data parallel over whole graphs, per-GPU batch fixed
(weak scaling), no edge sharding.  value = sum over ranks of (edges x conv layers) / max-rank time, i.e.
edges pushed through a convolution layer (forward + backward) per second: "edge-layers/s"; the rate of
batch edges through the whole 3-layer model is value / 3 (`model_edges_per_s`).

Beside `ms_per_step` (one replay of the captured HIP graph on ONE static batch) the line carries what a training loop
pays: `eager_ms_per_step` (the same batch, launch by launch from Python) and `fresh_batch_ms_per_step` (a NEW batch of the
same padded shape every step, copied into the graph's static buffers, with the edge orderings, M matrices and class table
rebuilt inside the replayed graph - what every Lightning step of the reference does, eigenn/model/model.py:279-327), plus
`per_rank` (edges, step and all-reduce times of every rank) so that a scaling run explains itself.

Also on the JSON line: `roofline` for the dominant kernel (tfc_conv_bwd_kernel; HBM-bound;
algorithmic bytes per SURVEY.md §8d, duration from HIP events recorded by the library on the
launch stream inside the timed region) and `cpu_baseline` (the CPU oracle = the reference's
unfused op order, timed on this box's host cores on a bounded sample; rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.md §3 / SURVEY.md §8d
    "c3": dict(n=4, a=2.5, graphs=32, emb=64, lmax=2, irreps="64x0o+64x0e+64x1o+64x1e+64x2o+64x2e", layers=3,
               desc="config 3: jittered simple-cubic 4^3=64 atoms, a=2.5, r_cut=5, batch 32, lmax=2, 64 channels"),
    "c4": dict(n=6, a=2.5, graphs=32, emb=128, lmax=3,
               irreps="128x0o+128x0e+128x1o+128x1e+128x2o+128x2e+128x3o+128x3e", layers=3,
               desc="config 4: 6^3=216 atoms, a=2.5, r_cut=5, batch 32 per GPU, lmax=3, 128 channels"),
    "c5": dict(n=12, a=2.1, graphs=1, emb=64, lmax=2, irreps="64x0o+64x0e+64x1o+64x1e+64x2o+64x2e", layers=3,
               desc="config 5: 12^3=1728 atoms, a=2.1, r_cut=5, 1 structure per GPU, lmax=2, 64 channels"),
}
R_CUT = 5.0
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32 matrix peak


def conv_layer_bytes(layer, n_nodes, n_edges):
    """Algorithmic bytes of one forward / backward conv launch:
    B_fwd = 4(D_in + S + W) + 16 + 4 D_mid/deg ;  B_bwd = 4(D_mid + D_in + S + W) + 16 + 4(W + S) + 4 D_in/deg."""
    core = layer.message_fn.tp.tp
    d_in, s, d_mid, w = core.irreps_in1.dim, core.irreps_in2.dim, core.irreps_out.dim, core.weight_numel
    deg = n_edges / max(n_nodes, 1)
    b_fwd = 4 * (d_in + s + w) + 16 + 4 * d_mid / deg
    b_bwd = 4 * (d_mid + d_in + s + w) + 16 + 4 * (w + s) + 4 * d_in / deg
    return b_fwd * n_edges, b_bwd * n_edges, dict(d_in=d_in, s=s, d_mid=d_mid, w=w)


def dense_flops(model, n_nodes, n_edges):
    """Algorithmic flops per step of the genuinely dense pieces, forward + backward
    (backward = input gradient + weight gradient = 2 x forward):
    radial MLP 2 (B H + H H (n_hidden - 1) + H W) per edge; o3.Linear 2 sum mul_in mul_out (2l+1) per node."""
    mlp = lin = 0.0
    for layer in model.convs:
        ws = [p for p in layer.message_fn.tp.weight_nn.parameters()]
        mlp += 3 * 2.0 * n_edges * sum(w.shape[0] * w.shape[1] for w in ws)
        for linear in (layer.message_fn.linear_1, layer.update_fn.linear_2):
            lin += 3 * 2.0 * n_nodes * sum(mi * mo * d for _, mi, _, mo, d, _, _ in linear.spec.instructions)
    return mlp, lin


def mfma_roofline(kern, flops, tags):
    """flops / (time of the launches booked under `tags` in the per-kernel breakdown pass)."""
    ms = sum(kern[t]["launches_per_step"] * kern[t]["avg_ms"] for t in tags if t in kern)
    if ms <= 0:
        return None
    ach = flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": ach, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK_TFLOPS,
            "flops_per_step": flops, "ms_per_step": ms, "kernels": [t for t in tags if t in kern]}


def pmc_traffic(workload, kernel="tfc_conv_bwd_kernel"):
    """(HBM bytes per launch of `kernel`, file) from the committed rocprofv3 PMC passes of this same
    command (FETCH_SIZE and WRITE_SIZE in separate runs, corrected with the calibration launches of
    tools/pmc_calibrate.py as MI355X_MICROARCH.md prescribes; tools/pmc_traffic.py).  Hardware counters
    cannot be read from inside the process, so this is the one figure of the line that is not measured
    live: (None, None) when no such profile is committed for the workload."""
    import glob

    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"pmc_traffic_{workload}.json")))
    if not hits:
        return None, None
    try:
        with open(hits[-1]) as f:
            ks = json.load(f)["kernels"]
        name = next(k for k in ks if k.startswith(kernel))
        return float(ks[name]["avg_bytes_per_launch"]), os.path.relpath(hits[-1], ROOT)
    except Exception:
        return None, None


def lib_hash():
    """First 16 hex digits of the SHA-256 of the libtfc.so being timed (ties profiles and PMC files to a build)."""
    import hashlib

    from matten_amd import _lib

    try:
        with open(_lib.LIB_PATH, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        return None


def fctp_flops(model, n_nodes, r=None):
    """Flops per step of the self connections.  r=None: the algorithmic (dense) count of SURVEY.md §8d, 2 sum mul_u
    mul_v mul_w (2l+1) per node forward; x, weight and attribute gradients are three more products of that size.
    With r (distinct attribute rows, padded): what the low-rank route executes - forward, dx and dW' contract over
    (u, s) with s < r instead of (u, v), and W' = R W, dW = R^T dW', dR = dW' . W are three products without the node
    index."""
    f = 0.0
    for layer in model.convs:
        sc = getattr(layer.update_fn, "self_connection", None)
        if sc is None:
            continue
        for _, mu, _, mv, _, mw, d, _, _ in sc.spec.instructions:
            if r is None:
                f += 4 * 2.0 * n_nodes * mu * mv * mw * d
            else:
                f += 3 * 2.0 * n_nodes * mu * r * mw * d + 3 * 2.0 * r * mv * mu * mw
    return f


def pmc_traffic_hash(workload):
    """The library hash recorded with the committed PMC file (None for files made before it was recorded)."""
    import glob

    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"pmc_traffic_{workload}.json")))
    if not hits:
        return None
    try:
        with open(hits[-1]) as f:
            return json.load(f).get("lib_sha256_16")
    except Exception:
        return None


def fresh_batch_timing(args, wl, rank, device, batch0, compute, bucket, sync, dist):
    """K steps, each on a batch the graph has not seen: three synthetic batches (other seeds) padded by null edges to one
    edge count, copied in turn into static buffers; the captured step (rebuild_topology=True) redoes both edge groupings,
    the M matrices and the class table from them.  Also the eager step on a fresh batch."""
    from matten_amd import keys as K
    from matten_amd.data.crystal import pad_null_edges
    from matten_amd.graphs import GraphedStep

    nb = 3
    raws = [batch0] + [build_batch(wl, rank, device, seed_offset=7919 * (i + 1)) for i in range(nb - 1)]
    edges = [int(b[K.EDGE_INDEX].shape[1]) for b in raws]
    e_pad = (max(edges) + 63) // 64 * 64
    if dist is not None:  # one shape on every rank is not needed (no edge exchange): each rank pads to its own maximum
        pass
    padded = [pad_null_edges(b, e_pad) for b in raws]
    if len({int(b[K.POSITIONS].shape[0]) for b in padded}) != 1:
        raise RuntimeError("fresh batches differ in their atom counts")
    from matten_amd.graphs import PackedBatch

    packed_static = PackedBatch(padded[0])  # every tensor of the batch is a view into one buffer: copy-in is ONE copy
    packed = [PackedBatch(b, like=packed_static) for b in padded]
    static = packed_static.tensors

    def copy_in(i):
        packed_static.load(packed[i % nb])

    trace = bool(os.environ.get("TFC_TRACE_LAUNCHES"))  # (debugging aid: phase markers between the library's launch names)

    def mark(msg):
        if trace:
            torch.cuda.synchronize()
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    mark("fresh: capture begins")
    replay = GraphedStep(lambda: compute(static), rebuild_topology=True)
    mark("fresh: captured")

    def gstep(i):
        if not os.environ.get("TFC_BENCH_FRESH_NOCOPY"):  # (diagnosis: the rebuilt-topology graph alone)
            copy_in(i)
        g, grads = replay()
        bucket.all_reduce_mean(grads=grads)

    def estep(i):
        copy_in(i)  # bumps the versions: every cache keyed on these tensors misses, as for a new batch
        compute(static)
        bucket.all_reduce_mean()

    out = {"batches": nb, "edges": edges, "padded_edges": e_pad}
    for name, fn in (("ms_per_step", gstep), ("eager_ms_per_step", estep)):
        for i in range(2):
            mark(f"fresh: {name} warm-up {i}")
            fn(i)
        sync()
        mark(f"fresh: {name} timed steps")
        t0 = time.perf_counter()
        for i in range(args.steps):
            fn(i)
        sync()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        if dist is not None:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        out[name] = float(el.item()) / args.steps * 1e3
        if name == "ms_per_step":  # what the host spends per step on the copy-in and the graph launch (queue empty at the start)
            host = []
            for i in range(4):  # (each from an empty queue: a launch behind a full queue waits for the GPU, not for the host)
                sync()
                t0 = time.perf_counter()
                fn(i)
                host.append(time.perf_counter() - t0)
            out["host_ms_per_launch"] = min(host) * 1e3
            sync()
    # the replayed step on a batch it was not captured on against the eager step on the same batch: same kernels in the
    # same order on the same inputs, so the same bits (tests/test_gpu_graph.py does this on a small case; here at full size)
    copy_in(1)
    g, grads = replay()
    got = [g.clone()] + [t.clone() for t in grads]
    copy_in(1)
    g, grads = compute(static)
    ref = [g] + list(grads)
    sync()
    out["replay_equals_eager"] = bool(all(torch.equal(a, b) for a, b in zip(got, ref)))
    out["replay_vs_eager_max_rel"] = max(
        float((a - b).abs().max().item()) / max(float(b.abs().max().item()), 1e-30) for a, b in zip(got, ref))
    out["launch"] = "hip_graph_replay, topology rebuilt in the graph, batch copied in"
    return out


def build_batch(wl, rank, device, seed_offset=0):
    from matten_amd.data.crystal import synthetic_batch
    from matten_amd.harness import species_index
    from matten_amd import keys as K

    batch, _ = synthetic_batch(wl["n"], wl["a"], wl["graphs"], seed0=1000 + rank * wl["graphs"] + seed_offset, r_cut=R_CUT)
    batch["species_index"] = species_index(batch[K.ATOMIC_NUMBERS])
    batch = {k: v.to(device) for k, v in batch.items()}
    return batch


def cpu_baseline(wl, avg_neigh, state, n_graphs_sample, threads):
    """The oracle (reference op order, unfused, autograd backward) on the host cores."""
    from matten_amd import keys as K
    from matten_amd.data.crystal import synthetic_batch
    from matten_amd.harness import ConvStack, species_index
    from oracle import eigenn_ref as R

    torch.set_num_threads(threads)
    batch, _ = synthetic_batch(wl["n"], wl["a"], n_graphs_sample, seed0=1000, r_cut=R_CUT)
    batch["species_index"] = species_index(batch[K.ATOMIC_NUMBERS])
    model = ConvStack(R, wl["emb"], wl["lmax"], wl["irreps"], wl["layers"], 64, 2, R_CUT, avg_neigh)
    model.load_state_dict(state)
    params = list(model.parameters())
    n_edges = int(batch[K.EDGE_INDEX].shape[1])

    def step():
        b = dict(batch)
        b[K.POSITIONS] = batch[K.POSITIONS].clone().requires_grad_(True)
        _, energy, _ = model(b, n_graphs_sample)
        torch.autograd.grad(energy.sum(), [b[K.POSITIONS]] + params, allow_unused=True)

    step()  # warm-up
    times = []
    t_budget = time.time()
    while len(times) < 3 or (time.time() - t_budget < 10.0 and len(times) < 10):
        t0 = time.time()
        step()
        times.append(time.time() - t0)
    med = sorted(times)[len(times) // 2]
    return {
        "value": n_edges * wl["layers"] / med,
        "unit": "edge-layers/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{n_graphs_sample} of the {wl['graphs']} graphs of the batch ({n_edges} edges), median of {len(times)} fwd+bwd steps, fp32",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N ranks on one GPU)")
    ap.add_argument("--overlap", type=int, default=0, help="1: launch the gradient all-reduce in slices from grad hooks during backward (not yet timed over RCCL, see tools/rehearse_dp.py)")
    ap.add_argument("--check", action="store_true", help="also verify a small case against the oracle before timing")
    ap.add_argument("--graph", type=int, default=1,
                    help="1 (default): the step (forward + backward into the flat gradient buffer) is captured once into a HIP graph "
                         "and every timed step is one replay of it (the gradient all-reduce stays outside the graph); 0: every step is "
                         "issued launch by launch from Python")
    ap.add_argument("--fresh", type=int, default=1,
                    help="1 (default): also time the step on a NEW batch every step (three synthetic batches of one padded shape, "
                         "copied into the static buffers of a graph that rebuilds the edge orderings itself); 0: skip")
    ap.add_argument("--graph-allreduce", type=int, default=0,
                    help="1: try to capture the gradient all-reduce (RCCL) inside the HIP graph as well")
    ap.add_argument("--dist-single", action="store_true",
                    help="with --gpus 1: still initialise the process group (world size 1) and run the gradient all-reduce and the "
                         "MAX/SUM reductions through it, so that the RCCL path executes on a one-GPU box")
    args = ap.parse_args()

    if os.environ.get("TFC_BENCH_WATCHDOG"):  # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler

        faulthandler.dump_traceback_later(int(os.environ["TFC_BENCH_WATCHDOG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Called directly: start one rank per GPU under torch's launcher as a CHILD process.  Nothing in this
        # process has touched a GPU yet (device_count() does not initialise the runtime on this image), and it
        # never will: it only waits and hands the child's exit status on.
        import socket
        import subprocess

        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} needs {args.gpus} devices, this machine shows {have}")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    # stdout carries the ONE JSON line and nothing else: whatever native libraries print there (RCCL's version banner
    # at the first collective) is sent to stderr, the line itself goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    dist = None
    if world > 1 or args.dist_single:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket

            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank %= max(torch.cuda.device_count(), 1)  # rehearsal: ranks may share a card
            dist.init_process_group(backend=args.backend)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    import matten_amd.nn as H
    from matten_amd import _lib
    from matten_amd import keys as K
    from matten_amd.harness import ConvStack

    wl = WORKLOADS[args.workload]
    if args.check and rank == 0:
        from tests.parity import run_layer_parity

        r = run_layer_parity("smoke", device=str(device))
        assert r["ok"], f"parity check failed: {r['worst']}"

    batch = build_batch(wl, rank, device)
    n_nodes, n_edges = int(batch[K.POSITIONS].shape[0]), int(batch[K.EDGE_INDEX].shape[1])
    # average_num_neighbors: a dataset constant in the reference (mean over the atoms of the training set,
    # eigenn/dataset/matbench.py:148-162); here the mean over the atoms of all ranks' batches, same on every rank
    cnt = torch.tensor([float(n_edges), float(n_nodes)], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    avg_neigh = float(cnt[0].item() / cnt[1].item())
    torch.manual_seed(35)
    model = ConvStack(H, wl["emb"], wl["lmax"], wl["irreps"], wl["layers"], 64, 2, R_CUT, avg_neigh)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(device)
    params = list(model.parameters())

    # one flat gradient buffer; param.grad are views into it => collectives on slices of it, no copies,
    # launched from grad hooks so that they overlap with the rest of backward
    from matten_amd.dp import FlatGradBucket

    bucket = FlatGradBucket(params, overlap=bool(args.overlap), bucket_bytes=int(os.environ.get("TFC_BUCKET_BYTES", 8 << 20)),
                            force=args.dist_single)

    pos0 = batch[K.POSITIONS]
    _SYNC_BEFORE_AR = bool(os.environ.get("TFC_SYNC_BEFORE_AR"))  # rehearsal aid

    def compute(src=None, reduce_inside=False):
        src = batch if src is None else src
        bucket.zero()
        b = dict(src)
        pos = src[K.POSITIONS].detach().requires_grad_(True)
        b[K.POSITIONS] = pos
        _, energy, _ = model(b, wl["graphs"])
        energy.sum().backward()
        if reduce_inside:
            bucket.all_reduce_mean()
        return pos.grad, [p.grad for p in params]  # (-forces, parameter gradients)

    def eager_step():
        g, _ = compute()
        if _SYNC_BEFORE_AR:
            torch.cuda.synchronize()
        bucket.all_reduce_mean()
        return g

    step = eager_step
    mode = "eager"
    ar_in_graph = False
    if args.graph and not args.overlap:
        # The batch shape is fixed, so the ~170 launches of a step are captured once and replayed: same kernels,
        # same arithmetic, same bits (tests/test_gpu_graph.py), no per-launch host work.
        from matten_amd.graphs import GraphedStep

        topo = [batch[K.EDGE_INDEX], batch[K.BATCH], batch["species_index"]]
        replay = None
        if args.graph_allreduce and dist is not None:
            # the collective as a node of the graph: RCCL kernels captured like any other launch
            try:
                replay = GraphedStep(lambda: compute(reduce_inside=True), static_topology=topo)
                ar_in_graph = True
            except Exception as ex:
                print(f"bench.py: capture with the all-reduce inside failed ({type(ex).__name__}: {ex}); capturing without it", file=sys.stderr)
                ar_in_graph = f"failed: {type(ex).__name__}: {str(ex)[:200]}"
                replay = None
                torch.cuda.synchronize()
        if replay is None:
            try:
                replay = GraphedStep(compute, static_topology=topo)
            except Exception as ex:  # capture refused: keep the launch-by-launch step
                print(f"bench.py: HIP-graph capture failed ({ex}); timing the eager step", file=sys.stderr)
        if replay is not None:
            mode = "hip_graph_replay"
            if ar_in_graph is True:
                def step():
                    return replay()[0]
            else:
                def step():
                    g, grads = replay()  # the tensors the captured step writes its gradients into
                    bucket.all_reduce_mean(grads=grads)
                    return g

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    L = _lib.lib()
    L.tfc_profile_reset()
    # HIP events around the conv kernels only, on their launch stream (eager steps; see below for graph replays)
    L.tfc_profile_enable(0 if os.environ.get("TFC_BENCH_NOPROF") or mode != "eager" else 2)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    own_elapsed = elapsed
    L.tfc_profile_enable(0)
    prof = _lib.profile_report()
    # the same batch launch by launch (no event timers), as a training loop without graphs would issue it
    eager_elapsed = None
    if mode != "eager":
        for _ in range(2):
            eager_step()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eager_step()
        sync()
        eager_elapsed = time.perf_counter() - t0
    # the gradient all-reduce alone (pack + collective + divide), for the per-rank report
    ar_ms = None
    if dist is not None:
        grads_now = [p.grad.clone() if p.grad is not None else None for p in params]
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            bucket.all_reduce_mean(grads=grads_now)
        torch.cuda.synchronize()
        ar_ms = (time.perf_counter() - t0) / args.steps * 1e3
    if mode != "eager":
        # HIP events cannot be recorded between the nodes of a replayed graph, so the conv-kernel durations come from
        # K eager steps of the same work, right behind the timed replays (same process, same buffers); the rocprofv3
        # kernel trace of this command (profiles/) times the kernels inside the replays themselves
        L.tfc_profile_reset()
        L.tfc_profile_enable(2)
        for _ in range(args.steps):
            eager_step()
        sync()
        L.tfc_profile_enable(0)
        prof = _lib.profile_report()
    # per-kernel breakdown of a step (informational; outside the timed region).  On ONE stream: with the radial networks
    # on their side stream two kernels share the chip and the events around either measure both (config 4: the
    # self connection's launches read 6.7 ms that way and 3.3 ms alone), so the `mfma_roofline` figures below are those
    # of each kernel running alone; the timed region and the convolution figures above keep the two streams.
    from matten_amd import ops as _ops_mod
    from matten_amd.nn import point_conv as _pc

    two_streams = _ops_mod.MLP_SIDE_STREAM
    _ops_mod.MLP_SIDE_STREAM = False
    _pc.reset_prefetch_state()
    try:
        eager_step()  # (caches and workspaces of the one-stream route)
        sync()
        L.tfc_profile_reset()
        L.tfc_profile_enable(1)
        n_break = 3
        for _ in range(n_break):
            eager_step()
        sync()
        L.tfc_profile_enable(0)
        prof_all = _lib.profile_report()
        L.tfc_profile_reset()
    finally:
        _ops_mod.MLP_SIDE_STREAM = two_streams
        _pc.reset_prefetch_state()

    # a NEW batch every step
    fresh = None
    if args.fresh and args.graph and not args.overlap:
        try:
            fresh = fresh_batch_timing(args, wl, rank, device, batch, compute, bucket, sync, dist)
        except Exception as ex:  # a report, never a reason to lose the headline number
            import traceback

            traceback.print_exc(file=sys.stderr)
            fresh = {"error": f"{type(ex).__name__}: {str(ex)[:300]}"}
            torch.cuda.synchronize()

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    e = torch.tensor([float(n_edges)], dtype=torch.float64, device=device)
    per_rank = None
    if dist is not None:
        mine = torch.tensor([float(n_edges), own_elapsed / args.steps * 1e3, ar_ms or 0.0,
                             (eager_elapsed or 0.0) / args.steps * 1e3], dtype=torch.float64, device=device)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": i, "edges": int(v[0].item()), "step_ms": v[1].item(), "allreduce_ms": v[2].item(),
                     "eager_step_ms": v[3].item() or None} for i, v in enumerate(allr)]
    if eager_elapsed is not None:
        te = torch.tensor([eager_elapsed], dtype=torch.float64, device=device)
        if dist is not None:
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
        eager_elapsed = float(te.item())
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(e, op=dist.ReduceOp.SUM)
    elapsed, total_edges = float(t.item()), float(e.item())

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = total_edges * wl["layers"] / (elapsed / args.steps)
        # roofline of the dominant kernel
        fb = bb = 0.0
        shapes = []
        for layer in model.convs:
            f, b, sh = conv_layer_bytes(layer, n_nodes, n_edges)
            fb += f
            bb += b
            shapes.append(sh)
        kern = {k: {"launches_per_step": c / n_break, "avg_ms": tot / c} for k, (c, tot) in prof_all.items()}
        n_layers = len(model.convs)
        roof = None
        if "tfc_conv_bwd_kernel" in prof:
            c, tot = prof["tfc_conv_bwd_kernel"]
            avg_s = tot / c * 1e-3
            bytes_per_launch = bb / n_layers
            ach = bytes_per_launch / avg_s / 1e9
            roof = {
                "kernel": "tfc_conv_bwd_kernel",
                "bound": "hbm",
                "achieved": ach,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS,
                "bytes_per_launch": bytes_per_launch,
                "avg_launch_ms": tot / c,
            }
            # `frac` prices the ALGORITHMIC bytes (every operand as if read from HBM once); `traffic` is what the
            # counters saw reach HBM (gathered rows are served from L2 / Infinity Cache), `traffic_frac` its rate
            roof["timing_source"] = ("HIP events recorded by the library inside the timed eager steps" if mode == "eager" else
                                     "HIP events recorded by the library over the same number of eager steps right behind the timed replays")
            roof["traffic"], roof["traffic_source"] = pmc_traffic(args.workload)
            roof["traffic_lib_sha256_16"] = pmc_traffic_hash(args.workload)
            roof["traffic_frac"] = roof["traffic"] / avg_s / 1e9 / HBM_PEAK_GBS if roof["traffic"] else None
            if "tfc_conv_fwd_kernel" in prof:
                c2, tot2 = prof["tfc_conv_fwd_kernel"]
                ach2 = (fb / n_layers) / (tot2 / c2 * 1e-3) / 1e9
                tr2, _ = pmc_traffic(args.workload, "tfc_conv_fwd_kernel")
                roof["fwd_kernel"] = {"achieved": ach2, "frac": ach2 / HBM_PEAK_GBS, "avg_launch_ms": tot2 / c2,
                                      "bytes_per_launch": fb / n_layers, "traffic": tr2,
                                      "traffic_frac": tr2 / (tot2 / c2 * 1e-3) / 1e9 / HBM_PEAK_GBS if tr2 else None}
                # the north star's figure: the convolution forward + backward together
                both = (fb + bb) / n_layers / ((tot / c + tot2 / c2) * 1e-3) / 1e9
                roof["fwd_plus_bwd"] = {"achieved": both, "frac": both / HBM_PEAK_GBS}
        mlp_flops, lin_flops = dense_flops(model, n_nodes, n_edges)
        dense = {
            "radial_mlp": mfma_roofline(kern, mlp_flops, ["mlp_fused_fwd", "mlp_fused_bwd", "gemm_mlp_fwd", "gemm_mlp_dh", "gemm_mlp_dw"]),
            "o3_linear": mfma_roofline(kern, lin_flops, ["gemm_linear_fwd", "gemm_linear_dx", "gemm_linear_dw"]),
            # algorithmic (dense) flops of the self connection; the launches run its low-rank route over the distinct
            # attribute rows, whose executed flops are emb / r times fewer - the figure prices the work asked for
            "self_connection": mfma_roofline(kern, fctp_flops(model, n_nodes), ["gemm_fctp_fwd", "gemm_fctp_dx", "gemm_fctp_dw", "gemm_fctp_da",
                                                                                  "gemm_fctp_lowrank_w", "gemm_fctp_da_rows", "tfc_class_rows_kernel"]),
        }
        sc = dense["self_connection"]
        if sc:
            from matten_amd import ops as _ops

            if _ops.ROUTE_COUNTS.get("fctp_low_rank", 0) > 0 and _ops.ROUTE_COUNTS.get("fctp_dense", 0) == 0:
                # `achieved` / `frac` price the flops the launches EXECUTE (the low-rank route contracts over the r distinct
                # attribute rows instead of the attribute width); the dense count the reference's formulation asks for, and
                # what it would mean at the same time, are kept beside them
                n_species = int(batch["species_index"].max().item()) + 1
                r_rows = max(4, (n_species + 3) // 4 * 4)
                ex = fctp_flops(model, n_nodes, r_rows)
                sc["algorithmic_flops_per_step"] = sc["flops_per_step"]
                sc["algorithmic_equivalent_tflops"] = sc["achieved"]
                sc["flops_per_step"] = ex
                sc["achieved"] = ex / (sc["ms_per_step"] * 1e-3) / 1e12
                sc["frac"] = sc["achieved"] / sc["peak"]
                sc["note"] = f"low-rank route over r = {r_rows} distinct attribute rows: executed flops; the dense formulation is algorithmic_flops_per_step"
        if dense["radial_mlp"]:
            # the last layer of the forward runs on the bf16 instruction with three-piece operands (6 instructions of 32 cycles
            # per K = 16 against 8 of 64): its own ceiling is 16 / 6 of the fp32 matrix peak
            dense["radial_mlp"]["note"] = "peak = fp32 MFMA; the forward's last layer runs as bf16 x 3 pieces (ceiling 419 TFLOP/s fp32-equivalent)"
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            sample = {"c3": 4, "c4": 1, "c5": 1}[args.workload]
            try:
                cpu = cpu_baseline(wl, avg_neigh, state, sample, threads)
            except Exception as ex:  # the baseline is a report, never a reason to lose the GPU number
                cpu = {"value": None, "unit": "edge-layers/s", "cores": threads, "kind": "port", "sample": f"failed: {ex}"}
        out = {
            "metric": "conv edges/sec fwd+bwd",
            "value": value,
            "unit": "edge-layers/s",
            "model_edges_per_s": value / wl["layers"],
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "eager_ms_per_step": None if eager_elapsed is None else eager_elapsed / args.steps * 1e3,
            "fresh_batch": fresh,
            "fresh_batch_ms_per_step": (fresh or {}).get("ms_per_step"),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": wl["desc"],
                "nodes_per_gpu": n_nodes,
                "edges_per_gpu": n_edges,
                "conv_layers": wl["layers"],
                "avg_num_neighbors": avg_neigh,
                "launch": mode,
                "step": "edge encoders + 3 conv layers forward, backward to all parameters and positions (forces)"
                + (f", gradient all-reduce ({args.backend}) in {len(bucket.bounds)} slices" + (" overlapped with backward" if args.overlap else " after backward") if dist is not None else ""),
                "parallelism": f"dp{world}",
                "layer_shapes": shapes,
            },
            "roofline": roof,
            "per_rank": per_rank,
            "exposed_collective_ms": None if per_rank is None else max(r["allreduce_ms"] for r in per_rank),
            "allreduce_in_graph": ar_in_graph,
            "lib_sha256_16": lib_hash(),
            "mfma_roofline": dense,
            "cpu_baseline": cpu,
            "kernels": kern,
            "kernels_timing_source": "HIP events recorded by the library around every kernel over 3 eager steps on ONE stream "
                                     "(each kernel alone on the chip; the timed region overlaps the radial networks with the rest)",
        }
        rl = kern.get("mlp_fused_fwd", {}).get("launches_per_step")
        out["radial_mlp_launches_per_step"] = rl
        if rl is not None and abs(rl - n_layers) > 1e-9:  # one radial network per layer and step, prefetched or not
            print(f"bench.py: WARNING {rl} radial-network launches per step for {n_layers} layers", file=sys.stderr)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
