"""Autograd wrappers around the libtfc C ABI (include/tfc.h).

PyTorch is the host here: it owns device memory, streams and the autograd graph.  Every
`torch.autograd.Function` below allocates its outputs and scratch with torch, hands raw
device pointers to the HIP kernels on torch's current stream, and implements the backward
the reference obtains from autograd (the reference has no custom Function anywhere,
SURVEY.md §3.3).  There is no CPU / eager fallback: tensors must live on a ROCm device.
"""
from __future__ import annotations

import ctypes as C
import functools
import math
import weakref
from os import environ as _os_environ
from dataclasses import dataclass
from typing import List, NamedTuple, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import FctpIns, GateSeg, GemmDesc, IMap, IrBlock, LinearIns, Path

__all__ = [
    "EdgeOrdering",
    "edge_ordering",
    "ConvPlan",
    "conv",
    "scatter_add",
    "edge_geometry",
    "spherical_harmonics",
    "radial_embedding",
    "LinearSpec",
    "linear",
    "radial_mlp",
    "FctpSpec",
    "fctp",
    "GateSpec",
    "gate",
    "norm_activation",
]


def _stream():
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def _ptr(t: Optional[torch.Tensor], offset: int = 0):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr() + offset * t.element_size())


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"matten_amd: `{name}` must live on a ROCm device (no CPU fallback exists); got {t.device}")
    if t.dtype != torch.float32:
        raise TypeError(f"matten_amd: `{name}` must be float32, got {t.dtype}")
    return t.contiguous()


def _tracked(t):
    """Contiguous, made HERE - outside the autograd.Function - so that autograd sees the copy: the functions
    save their inputs for a differentiable backward (create_graph=True), and a copy made inside Function.forward would
    be cut off from the caller's graph (second-order terms through a non-contiguous input would vanish)."""
    if t is None or t.is_contiguous():
        return t
    return t.contiguous()  # (dtype and device are checked, not converted, by _f32)


# Everything the ops keep BETWEEN calls that holds device tensors lives in a dict made by _tensor_cache(): one registry,
# walked by clear_caches() - a cache added later cannot be forgotten there (a tensor freed inside a HIP-graph capture
# that an earlier eager step used on another stream makes the allocator record an event mid-capture, and
# hipStreamEndCapture then fails).  The caches are process-global on purpose: every entry is keyed by the identity of
# the tensor it was derived from (address / object id AND version counter, with the source held alive so that the
# address cannot be reused), a miss always falls back to recomputing, and workspaces are keyed by (device, stream), on
# which launches are ordered - so two host threads can at worst evict each other's entries, never read a wrong one.
# What the keys cannot see is a write through `.data`, which does not bump the version counter: do not use it on
# tensors handed to these ops.
_TENSOR_CACHES: "list[dict]" = []


def _tensor_cache() -> dict:
    d: dict = {}
    _TENSOR_CACHES.append(d)
    return d


# ======================================================================================
# edge ordering
# ======================================================================================


class EdgeOrdering:
    """The two groupings of the edge list the kernels reduce over (tfc_csr_build):
    by receiver edge_index[1] (forward: sum of messages) and by sender edge_index[0]
    (backward: gradient of the gathered sender features)."""

    def __init__(self, edge_index: torch.Tensor, n_nodes: int):
        if edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise ValueError(f"edge_index must have shape [2, E], got {tuple(edge_index.shape)}")
        if not edge_index.is_cuda:
            raise RuntimeError("matten_amd: edge_index must live on a ROCm device (no CPU fallback exists)")
        capturing = torch.cuda.is_current_stream_capturing()  # (graphs.GraphedStep(rebuild_topology=True))
        if not capturing:
            check_edge_orderings()  # the deferred verdict on earlier orderings (see below)
        ei = edge_index.to(torch.int64).contiguous()
        self.edge_index = ei
        self.n_nodes = int(n_nodes)
        self.n_edges = int(ei.shape[1])
        dev = ei.device
        E, N = self.n_edges, self.n_nodes
        L = _lib.lib()
        self.status = torch.zeros(2, dtype=torch.int32, device=dev)
        ws = torch.empty(max(int(L.tfc_csr_workspace_bytes(N, E)), 8), dtype=torch.uint8, device=dev)

        def build(key_row, other_row, slot):
            rp = torch.empty(N + 1, dtype=torch.int32, device=dev)
            eid = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
            nbr = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
            _lib.check(
                L.tfc_csr_build(
                    _ptr(ei, key_row * E), _ptr(ei, other_row * E), E, N, _ptr(rp), _ptr(eid), _ptr(nbr),
                    _ptr(self.status, slot), _ptr(ws), _stream(),
                )
            )
            return rp, eid, nbr

        self.by_dst = build(1, 0, 0)  # centre = receiver, nbr = sender
        self.by_src = build(0, 1, 1)  # centre = sender,   nbr = receiver
        self._edge_rows = None
        self._m_cache = {}
        self._host_status = None
        if capturing:
            pass  # no host work inside a capture: the caller of a replayed step validates its batches itself (self.status)
        elif VALIDATE_EDGES == "sync":
            self.validate()
        elif VALIDATE_EDGES == "deferred" and E:
            # the flag travels to pinned host memory behind the build; whoever builds the next ordering (or calls
            # check_edge_orderings()) reads it without stalling the queue - an index error surfaces one batch
            # late, like any asynchronous device-side error
            self._host_status = torch.empty(2, dtype=torch.int32, pin_memory=True)
            self._host_status.copy_(self.status, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            _PENDING_STATUS.append((ev, self._host_status, E, N))

    @classmethod
    def from_neighbor_list(cls, edge_index: torch.Tensor, n_nodes: int, row_ptr: torch.Tensor) -> "EdgeOrdering":
        """Ordering of a symmetric, canonically ordered neighbour list (tfc_nl_fill's output) without
        the sort: tfc_nl_ordering.  Falls back to the general builder if the list turns out not to be
        symmetric (one host synchronisation to read that flag).  Registered for `edge_index`, so the
        layers that later see this tensor reuse it."""
        self = cls.__new__(cls)
        ei = edge_index.contiguous()
        if ei.dtype != torch.int64 or not ei.is_cuda or ei.dim() != 2 or ei.shape[0] != 2:
            raise ValueError("from_neighbor_list: edge_index must be a [2, E] int64 tensor on a ROCm device")
        E, N, dev = int(ei.shape[1]), int(n_nodes), ei.device
        if row_ptr.dtype != torch.int32 or row_ptr.numel() != N + 1 or row_ptr.device != dev:
            raise ValueError("from_neighbor_list: row_ptr must be [N + 1] int32 on the device of edge_index")
        self.edge_index, self.n_nodes, self.n_edges = ei, N, E
        self.status = torch.zeros(2, dtype=torch.int32, device=dev)
        arrs = [torch.empty(max(E, 1), dtype=torch.int32, device=dev) for _ in range(4)]
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().tfc_nl_ordering(_ptr(ei), _ptr(row_ptr), E, N, *[_ptr(a) for a in arrs], _ptr(flag), _stream()))
        if int(flag.item()) != 0:
            return edge_ordering(edge_index, n_nodes)
        rp = row_ptr.contiguous()
        self.by_src = (rp, arrs[0], arrs[1])
        self.by_dst = (rp, arrs[2], arrs[3])
        self._edge_rows = None
        self._m_cache = {}
        _ORDER_CACHE[id(edge_index)] = (weakref.ref(edge_index), edge_index._version, N, self)
        return self

    def m_matrices(self, plan: "ConvPlan", y: torch.Tensor, rows) -> torch.Tensor:
        """The per-edge matrices M_e (CG tensor contracted with the edge attributes), rows in edge order, built
        once and shared - forward and backward - by every layer whose plan lays M out the same way (and that
        sees the same edge-attribute tensor): tfc_conv_m_build."""
        L = _lib.lib()
        h = plan.handle(y.device)
        key = (int(L.tfc_conv_m_signature(h)), y.data_ptr(), y._version)
        hit = self._m_cache.get(key)
        if hit is not None:
            return hit[0]
        E = y.shape[0]
        M = torch.empty(int(L.tfc_conv_m_floats(h, E)), dtype=torch.float32, device=y.device)
        _lib.check(L.tfc_conv_m_build(h, _ptr(y), E, _ptr(M), _stream()))
        if len(self._m_cache) >= 4:
            self._m_cache.pop(next(iter(self._m_cache)))
        self._m_cache[key] = (M, y)  # holding y keeps its address from being reused while cached
        return M

    def per_edge(self):
        """Ordering in which every edge is its own centre (used to materialise per-edge
        messages for callers that want the unfused EDGE_MESSAGE tensor)."""
        if self._edge_rows is None:
            dev = self.edge_index.device
            E = self.n_edges
            rp = torch.arange(E + 1, dtype=torch.int32, device=dev)
            eid = torch.arange(max(E, 1), dtype=torch.int32, device=dev)
            nbr = self.edge_index[0].to(torch.int32).contiguous() if E else eid
            self._edge_rows = (rp, eid, nbr)
        return self._edge_rows

    def validate(self):
        """Synchronises: raises if any index was outside [0, n_nodes)."""
        if int(self.status.max().item()) != 0:
            raise IndexError("edge_index contains node ids outside [0, n_nodes)")


_ORDER_CACHE: "dict[int, tuple]" = _tensor_cache()
# TFC_VALIDATE_EDGES: "deferred" (default; see EdgeOrdering.__init__), "sync" (raise at once: one host
# synchronisation per new edge_index tensor) or "off".  The reference raises IndexError from its gather when
# an edge names a node outside [0, N); here such edges are dropped from the groupings and flagged.
VALIDATE_EDGES = _os_environ.get("TFC_VALIDATE_EDGES", "deferred")
_PENDING_STATUS: "list[tuple]" = []


def clear_caches():
    """Drop everything the ops keep between calls (edge orderings and their M matrices, channel-contiguous
    gradient copies, attribute factorizations, split-K workspaces).  They are rebuilt on demand; needed before a
    HIP-graph capture (matten_amd.graphs), where freeing a cached tensor that an earlier eager step used on the
    default stream would record an allocator event on that stream in the middle of the capture."""
    torch.cuda.synchronize()
    check_edge_orderings(wait=True)
    for cache in _TENSOR_CACHES:
        cache.clear()


def snapshot_caches():
    """References to everything the caches hold right now.  A captured step (matten_amd.graphs) keeps one: the
    addresses of cached tensors it hit during the capture (workspaces of the side stream, orderings, class tables) are
    baked into its graph, and a later clear_caches() - the next GraphedStep calls it - must not hand their memory to
    someone else while that graph can still be replayed."""
    return [list(cache.values()) for cache in _TENSOR_CACHES]


def check_edge_orderings(wait: bool = False):
    """Raise IndexError if an edge ordering built earlier saw node ids outside [0, n_nodes).  Only flags that
    have already arrived on the host are looked at unless `wait`."""
    keep = []
    bad = None
    for ev, flags, e, n in _PENDING_STATUS:
        if wait:
            ev.synchronize()
        if ev.query():
            if int(flags.max()) != 0:
                bad = (e, n)
        else:
            keep.append((ev, flags, e, n))
    _PENDING_STATUS[:] = keep
    if bad is not None:
        raise IndexError(f"edge_index ({bad[0]} edges) contains node ids outside [0, {bad[1]}); those edges were dropped "
                         "from the convolution of an EARLIER batch, whose results are invalid")


def _flush_edge_checks_at_exit():
    """The verdict on the LAST batch has no later ordering to surface it: report it when the interpreter exits (training
    and prediction loops should call check_edge_orderings(wait=True) at the end of an epoch / a predict call)."""
    try:
        check_edge_orderings(wait=True)
    except IndexError as ex:
        import sys

        print(f"matten_amd: {ex}", file=sys.stderr)
    except Exception:
        pass


import atexit as _atexit

_atexit.register(_flush_edge_checks_at_exit)


def edge_ordering(edge_index: torch.Tensor, n_nodes: int) -> EdgeOrdering:
    """Ordering for this edge_index tensor, built once and shared by all layers (and by
    forward and backward) that see the same tensor object."""
    key = id(edge_index)
    hit = _ORDER_CACHE.get(key)
    if hit is not None:
        ref, version, n, ordering = hit
        if ref() is edge_index and version == edge_index._version and n == n_nodes:
            return ordering
    ordering = EdgeOrdering(edge_index, n_nodes)
    if len(_ORDER_CACHE) > 64:
        for k in [k for k, v in _ORDER_CACHE.items() if v[0]() is None]:
            _ORDER_CACHE.pop(k, None)
        while len(_ORDER_CACHE) > 64:
            _ORDER_CACHE.pop(next(iter(_ORDER_CACHE)))
    _ORDER_CACHE[key] = (weakref.ref(edge_index), edge_index._version, n_nodes, ordering)
    return ordering


# ======================================================================================
# fused gather + uvu tensor product + per-node sum
# ======================================================================================


class ConvPlan:
    """Host handle of a tfc_conv_plan (device tables live on the device that was current
    when the plan was first used)."""

    def __init__(self, paths: Sequence[dict], d_in: int, d_sh: int, d_out: int, w_numel: int, out_blocks=None):
        self.paths = [dict(p) for p in paths]
        self.d_in, self.d_sh, self.d_out, self.w_numel = int(d_in), int(d_sh), int(d_out), int(w_numel)
        self._handles = {}
        # (off, mul, dim) blocks of the output row: with them the kernels also write the channel-contiguous
        # copies the neighbouring o3.Linear layers read (tfc_conv_fwd_m_t / tfc_conv_bwd_m_t)
        self.out_blocks = tuple(sorted(set(out_blocks))) if out_blocks and CONV_TRANSPOSED else None
        self.in_blocks = tuple(sorted({(p["x_off"], p["mul"], 2 * p["l1"] + 1) for p in self.paths}))

    def handle(self, device: torch.device):
        idx = device.index if device.index is not None else torch.cuda.current_device()
        h = self._handles.get(idx)
        if h is None:
            arr = (Path * len(self.paths))()
            for a, p in zip(arr, self.paths):
                a.l1, a.l2, a.l3, a.mul = p["l1"], p["l2"], p["l3"], p["mul"]
                a.x_off, a.y_off, a.out_off, a.w_off = p["x_off"], p["y_off"], p["out_off"], p["w_off"]
                a.coeff = p["coeff"]
            out = C.c_void_p()
            with torch.cuda.device(idx):
                _lib.check(
                    _lib.lib().tfc_conv_plan_create(arr, len(self.paths), self.d_in, self.d_sh, self.d_out, self.w_numel, C.byref(out))
                )
                if self.out_blocks:
                    barr = (IrBlock * len(self.out_blocks))()
                    for b, (off, mul, dim) in zip(barr, self.out_blocks):
                        b.off, b.mul, b.dim = off, mul, dim
                    _lib.check(_lib.lib().tfc_conv_plan_set_out_blocks(out, barr, len(barr)))
            h = out
            self._handles[idx] = h
        return h

    def workspace(self, device, n_edges: int, need_dy: bool) -> torch.Tensor:
        n = int(_lib.lib().tfc_conv_workspace_floats(self.handle(device), n_edges, 1 if need_dy else 0))
        return torch.empty(max(n, 1), dtype=torch.float32, device=device)

    def __del__(self):
        try:
            for h in self._handles.values():
                _lib.lib().tfc_conv_plan_destroy(h)
        except Exception:
            pass


def _conv_fwd_raw(x, y, w, plan: ConvPlan, ordering, fwd_rows, n_out: int, scale: float):
    x, y, w = _f32(x, "x"), _f32(y, "edge_attrs"), _f32(w, "weights")
    E = y.shape[0]
    if x.shape[1] != plan.d_in or y.shape[1] != plan.d_sh or w.shape[1] != plan.w_numel or w.shape[0] != E:
        raise ValueError(
            f"conv: shapes x{tuple(x.shape)} y{tuple(y.shape)} w{tuple(w.shape)} do not match the plan "
            f"(d_in={plan.d_in}, d_sh={plan.d_sh}, w_numel={plan.w_numel})"
        )
    rp, eid, nbr = fwd_rows
    if rp.numel() != n_out + 1:
        raise ValueError("conv: ordering does not match the number of output rows")
    out = torch.empty(n_out, plan.d_out, dtype=torch.float32, device=x.device)
    # the channel-contiguous second copy serves the o3.Linear on NODE rows that follows; the per-edge route (n_out = E:
    # 1.4 GB at the last config-3 layer) has no such consumer
    per_edge = fwd_rows is ordering._edge_rows
    out_t = torch.empty(n_out * plan.d_out, dtype=torch.float32, device=x.device) if plan.out_blocks and n_out and not per_edge else None
    M = ordering.m_matrices(plan, y, fwd_rows)
    _lib.check(
        _lib.lib().tfc_conv_fwd_m_t(
            plan.handle(x.device), _ptr(x), _ptr(M), _ptr(w), _ptr(rp), _ptr(eid), _ptr(nbr), n_out, E,
            float(scale), _ptr(out), _ptr(out_t), _stream(),
        )
    )
    if out_t is not None:
        _transposed_put(out, plan.out_blocks, out_t)
    return out, x, y, w


def _conv_bwd_raw(gout, x, y, w, plan: ConvPlan, ordering, bwd_rows, scale: float, need_dy: bool):
    gout, x, y, w = _f32(gout, "grad_output"), _f32(x, "x"), _f32(y, "edge_attrs"), _f32(w, "weights")
    E = y.shape[0]
    rp, eid, nbr = bwd_rows
    dx = torch.empty_like(x)
    dw = torch.empty_like(w)
    dy = torch.empty_like(y) if need_dy else None
    L = _lib.lib()
    h = plan.handle(x.device)
    M = ordering.m_matrices(plan, y, bwd_rows)
    dyp = torch.empty(int(L.tfc_conv_dy_floats(h, E)), dtype=torch.float32, device=x.device) if need_dy else None
    per_edge = eid is nbr  # (conv(per_edge=True) passes the edge ids in both places)
    dx_t = torch.empty(x.numel(), dtype=torch.float32, device=x.device) if plan.out_blocks and x.numel() and not per_edge else None
    _lib.check(
        L.tfc_conv_bwd_m_t(
            h, _ptr(x), _ptr(M), _ptr(w), _ptr(gout), _ptr(rp), _ptr(eid), _ptr(nbr),
            x.shape[0], E, scale, _ptr(dx), _ptr(dx_t), _ptr(dw), _ptr(dy), _ptr(dyp), _stream(),
        )
    )
    if dx_t is not None:  # holds the in1 blocks only: serves a consumer that asks for (a subset of) those
        _transposed_put(dx, plan.in_blocks, dx_t)
    return dx, dy, dw


def _add(a, b):
    return b if a is None else (a if b is None else a + b)


class _ConvFn(torch.autograd.Function):
    """out = T(x, y, w): the fused gather + 'uvu' product + per-node sum, trilinear in its three inputs."""

    @staticmethod
    def forward(ctx, x, y, w, plan: ConvPlan, ordering, fwd_rows, bwd_rows, n_out: int, scale: float):
        out, x, y, w = _conv_fwd_raw(x, y, w, plan, ordering, fwd_rows, n_out, scale)
        ctx.save_for_backward(x, y, w)
        ctx.args = (plan, ordering, fwd_rows, bwd_rows, n_out, float(scale))
        return out

    @staticmethod
    def backward(ctx, gout):
        x, y, w = ctx.saved_tensors
        plan, ordering, fwd_rows, bwd_rows, n_out, scale = ctx.args
        need_dy = ctx.needs_input_grad[1]
        if torch.is_grad_enabled():  # create_graph=True: the backward itself must be differentiable
            dx, dy, dw = _ConvBwdFn.apply(gout, x, y, w, *ctx.args)
            return dx, (dy if need_dy else None), dw, None, None, None, None, None, None
        dx, dy, dw = _conv_bwd_raw(gout, x, y, w, plan, ordering, bwd_rows, scale, need_dy)
        return dx, dy, dw, None, None, None, None, None, None


class _ConvBwdFn(torch.autograd.Function):
    """(dx, dy, dw) = the three partial transposes of T applied to gout.  T is trilinear, so every derivative
    of these is T or one of its transposes again with one argument replaced by the incoming cotangent: the
    double backward (training on forces) runs on the same two kernels, to any order."""

    @staticmethod
    def forward(ctx, gout, x, y, w, plan, ordering, fwd_rows, bwd_rows, n_out, scale):
        dx, dy, dw = _conv_bwd_raw(gout, x, y, w, plan, ordering, bwd_rows, scale, True)
        ctx.save_for_backward(gout, x, y, w)
        ctx.args = (plan, ordering, fwd_rows, bwd_rows, n_out, scale)
        ctx.set_materialize_grads(False)
        return dx, dy, dw

    @staticmethod
    def backward(ctx, ggx, ggy, ggw):
        gout, x, y, w = ctx.saved_tensors
        a = ctx.args
        g_gout = g_x = g_y = g_w = None
        if ggx is not None:  # dx = T^x(gout; y, w) with x replaced
            g_gout = _add(g_gout, _ConvFn.apply(ggx, y, w, *a))
            _, dy1, dw1 = _ConvBwdFn.apply(gout, ggx, y, w, *a)
            g_y, g_w = _add(g_y, dy1), _add(g_w, dw1)
        if ggy is not None:
            g_gout = _add(g_gout, _ConvFn.apply(x, ggy, w, *a))
            dx2, _, dw2 = _ConvBwdFn.apply(gout, x, ggy, w, *a)
            g_x, g_w = _add(g_x, dx2), _add(g_w, dw2)
        if ggw is not None:
            g_gout = _add(g_gout, _ConvFn.apply(x, y, ggw, *a))
            dx3, dy3, _ = _ConvBwdFn.apply(gout, x, y, ggw, *a)
            g_x, g_y = _add(g_x, dx3), _add(g_y, dy3)
        return g_gout, g_x, g_y, g_w, None, None, None, None, None, None


def conv(x, y, w, plan: ConvPlan, ordering: EdgeOrdering, scale: float = 1.0, per_edge: bool = False):
    """out[n] = scale * sum_{e -> n} TP_uvu(x[sender(e)], y[e]; w[e]); `per_edge=True`
    returns the un-summed messages [E, d_out] instead (each edge is its own row)."""
    x, y, w = _tracked(x), _tracked(y), _tracked(w)
    if per_edge:
        rps, eids, _ = ordering.by_src  # backward gathers the gradient row of the edge itself
        return _ConvFn.apply(x, y, w, plan, ordering, ordering.per_edge(), (rps, eids, eids), ordering.n_edges, scale)
    return _ConvFn.apply(x, y, w, plan, ordering, ordering.by_dst, ordering.by_src, ordering.n_nodes, scale)


class _ScatterAddFn(torch.autograd.Function):
    """scatter_add over dim 0 as a fixed-order sum over each destination's rows; backward is a
    row gather (whose backward is this sum again: differentiable to any order)."""

    @staticmethod
    def forward(ctx, src, index, dim_size: int):
        src = _f32(src, "src")
        index = index.to(torch.int64).contiguous()
        e, d = src.shape
        ident = torch.arange(e, dtype=torch.int64, device=src.device)
        order = EdgeOrdering(torch.stack([index, ident]), max(int(dim_size), e))
        rp, eid, _ = order.by_src  # grouped by `index`
        out = torch.empty(order.n_nodes, d, dtype=torch.float32, device=src.device)
        _lib.check(_lib.lib().tfc_segment_sum(_ptr(src), _ptr(rp), _ptr(eid), order.n_nodes, d, _ptr(out), _stream()))
        ctx.save_for_backward(index)
        ctx.n_src = e
        return out[:dim_size]

    @staticmethod
    def backward(ctx, g):
        (index,) = ctx.saved_tensors
        return _GatherRowsFn.apply(g, index, ctx.n_src), None, None


class _GatherRowsFn(torch.autograd.Function):
    """out[r] = src[index[r]] (tfc_gather_rows)."""

    @staticmethod
    def forward(ctx, src, index, n_rows: int):
        src = _f32(src, "src")
        d = src.shape[1]
        out = torch.empty(n_rows, d, dtype=torch.float32, device=src.device)
        _lib.check(_lib.lib().tfc_gather_rows(_ptr(src), _ptr(index), n_rows, src.shape[0], d, _ptr(out), _stream()))
        ctx.save_for_backward(index)
        ctx.n_src = src.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        (index,) = ctx.saved_tensors
        return _ScatterAddFn.apply(g, index, ctx.n_src), None, None


def scatter_add(src, index, dim_size: int):
    return _ScatterAddFn.apply(src, index, dim_size)


# ======================================================================================
# per-edge encoders
# ======================================================================================


def _dvec_to_dpos_raw(dvec, ordering: EdgeOrdering):
    dvec = _f32(dvec, "dvec")
    dpos = torch.empty(ordering.n_nodes, 3, dtype=torch.float32, device=dvec.device)
    if ordering.n_edges == 0:
        return dpos.zero_()
    rps, eids, _ = ordering.by_src
    rpd, eidd, _ = ordering.by_dst
    _lib.check(
        _lib.lib().tfc_edge_vec_bwd(_ptr(dvec), _ptr(rps), _ptr(eids), _ptr(rpd), _ptr(eidd), ordering.n_nodes, _ptr(dpos), _stream())
    )
    return dpos


class _ScatterVecFn(torch.autograd.Function):
    """dpos[n] = sum_{e into n} dvec[e] - sum_{e out of n} dvec[e]: the transpose of pos -> pos[j] - pos[i]."""

    @staticmethod
    def forward(ctx, dvec, ordering: EdgeOrdering):
        ctx.ordering = ordering
        return _dvec_to_dpos_raw(dvec, ordering)

    @staticmethod
    def backward(ctx, g):
        return _GatherVecFn.apply(g, ctx.ordering), None


class _GatherVecFn(torch.autograd.Function):
    """vec[e] = pos[j] - pos[i] (no periodic shift: the linear part of with_edge_vectors)."""

    @staticmethod
    def forward(ctx, pos, ordering: EdgeOrdering):
        pos = _f32(pos, "pos")
        E = ordering.n_edges
        ctx.ordering = ordering
        vec = torch.empty(E, 3, dtype=torch.float32, device=pos.device)
        if E:
            length = torch.empty(E, dtype=torch.float32, device=pos.device)
            _lib.check(
                _lib.lib().tfc_edge_encode_fwd(
                    _ptr(pos), _ptr(ordering.edge_index), None, None, None, E, pos.shape[0], 1, 0, 1, None, 0, 1.0, 6.0,
                    _ptr(vec), _ptr(length), None, None, _stream(),
                )
            )
        return vec

    @staticmethod
    def backward(ctx, g):
        return _ScatterVecFn.apply(g, ctx.ordering), None


def _dvec_to_dpos(dvec, ordering: EdgeOrdering):
    if torch.is_grad_enabled():
        return _ScatterVecFn.apply(dvec, ordering)
    return _dvec_to_dpos_raw(dvec, ordering)


def _dvec_to_dcell(dvec, shift, batch, n_graphs: int, ordering: EdgeOrdering):
    """dcell [G,3,3] = sum over the edges of each structure of shift[e]^T dvec[e]: per sender node
    (tfc_edge_cell_bwd), then over the nodes of each structure (tfc_csr_build + tfc_segment_sum)."""
    dev = dvec.device
    N = ordering.n_nodes
    part = torch.empty(N, 9, dtype=torch.float32, device=dev)
    rps, eids, _ = ordering.by_src
    L = _lib.lib()
    _lib.check(L.tfc_edge_cell_bwd(_ptr(dvec), _ptr(shift), _ptr(rps), _ptr(eids), N, _ptr(part), _stream()))
    if n_graphs == 1:
        rp = torch.tensor([0, N], dtype=torch.int32, device=dev)
        nid = torch.arange(max(N, 1), dtype=torch.int32, device=dev)
    else:
        rp = torch.empty(n_graphs + 1, dtype=torch.int32, device=dev)
        nid = torch.empty(max(N, 1), dtype=torch.int32, device=dev)
        other = torch.empty(max(N, 1), dtype=torch.int32, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        ws = torch.empty(max(int(L.tfc_csr_workspace_bytes(n_graphs, N)), 8), dtype=torch.uint8, device=dev)
        _lib.check(L.tfc_csr_build(_ptr(batch), _ptr(batch), N, n_graphs, _ptr(rp), _ptr(nid), _ptr(other), _ptr(status), _ptr(ws), _stream()))
    dcell = torch.empty(n_graphs, 9, dtype=torch.float32, device=dev)
    _lib.check(L.tfc_segment_sum(_ptr(part), _ptr(rp), _ptr(nid), n_graphs, 9, _ptr(dcell), _stream()))
    return dcell.view(n_graphs, 3, 3)


def _sum_rows(part: torch.Tensor) -> torch.Tensor:
    """Column sums of the [blocks, B] partial results.  torch's reduction runs this shape in ONE workgroup (23 us on the
    critical path at config 3); the same sums as a matrix-vector product take 5."""
    return torch.mv(part.t(), torch.ones(part.shape[0], dtype=part.dtype, device=part.device))


def _edge_bwd_raw(vec, length, lmax, sh_normalize, bw, r_max, p, dsh, demb, dvec_in, dlen, need_dbw):
    """tfc_edge_encode_bwd: (dvec [E,3], dbw [B] or None) from the gradients of sh / emb / len / vec."""
    E = vec.shape[0]
    dvec = torch.empty_like(vec)
    # (contiguous copies are bound to names: a temporary would be freed, and its block possibly reused by the
    # next argument's copy, before the kernel that reads it has been queued)
    dsh = _f32(dsh, "dsh") if dsh is not None else None
    demb = _f32(demb, "demb") if demb is not None else None
    dvec_in = _f32(dvec_in, "dvec") if dvec_in is not None else None
    dlen = _f32(dlen, "dlen") if dlen is not None else None
    B = bw.shape[0] if bw is not None else 0
    if E == 0:
        return dvec, (torch.zeros_like(bw) if need_dbw else None)
    L = _lib.lib()
    part = torch.empty(int(L.tfc_edge_bwd_blocks(E)), B, dtype=torch.float32, device=vec.device) if (need_dbw and demb is not None) else None
    _lib.check(
        L.tfc_edge_encode_bwd(
            _ptr(vec), _ptr(length), E, max(lmax, 0), 1 if sh_normalize else 0, _ptr(bw) if demb is not None else None,
            B if demb is not None else 0, float(r_max), float(p), _ptr(dsh), _ptr(demb), _ptr(dvec_in), _ptr(dlen), _ptr(dvec),
            _ptr(part), _stream(),
        )
    )
    dbw = None
    if need_dbw:
        dbw = _sum_rows(part) if part is not None else torch.zeros_like(bw)  # second stage of the fixed-shape two-stage sum
    return dvec, dbw


class _EdgeBwdFn(torch.autograd.Function):
    """The backward of the per-edge encoders as a differentiable op (double backward, training on forces):
    (dvec, dbw) = gradient of phi = <dsh, Y(vec)> + <demb, emb(|vec|, bw)> + <dlen, |vec|> + <dvec_in, vec>.
    Its own backward is one dual-number pass (tfc_edge_encode_bwd2)."""

    @staticmethod
    def forward(ctx, vec, bw, dsh, demb, dlen, dvec_in, lmax: int, sh_normalize: bool, r_max: float, p: float):
        vec = _f32(vec, "edge_vectors")
        bw = _f32(bw, "bessel_weights") if bw is not None else None
        dvec, dbw = _edge_bwd_raw(vec, None, lmax, sh_normalize, bw, r_max, p, dsh, demb, dvec_in, dlen, bw is not None)
        ctx.save_for_backward(vec, bw, dsh, demb, dlen)
        ctx.args = (lmax, bool(sh_normalize), float(r_max), float(p), dvec_in is not None)
        ctx.set_materialize_grads(False)
        if dbw is None:
            dbw = torch.zeros(0, dtype=torch.float32, device=vec.device)
        return dvec, dbw

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gv, gb):
        vec, bw, dsh, demb, dlen = ctx.saved_tensors
        lmax, sh_normalize, r_max, p, has_dvec_in = ctx.args
        E = vec.shape[0]
        dev = vec.device
        if bw is None or demb is None or gb is None or gb.numel() == 0:
            gb = None
        if gv is None and gb is None:
            return (None,) * 10
        gv = _f32(gv, "gv") if gv is not None else None
        gb = _f32(gb, "gb") if gb is not None else None
        dsh_c = _f32(dsh, "dsh") if dsh is not None else None
        demb_c = _f32(demb, "demb") if demb is not None else None
        dlen_c = _f32(dlen, "dlen") if dlen is not None else None
        S = (max(lmax, 0) + 1) ** 2
        B = bw.shape[0] if (bw is not None and demb is not None) else 0
        L = _lib.lib()
        g_vec = torch.empty_like(vec)
        g_dsh = torch.empty(E, S, dtype=torch.float32, device=dev) if dsh is not None else None
        g_demb = torch.empty(E, B, dtype=torch.float32, device=dev) if demb is not None else None
        g_dlen = torch.empty(E, dtype=torch.float32, device=dev) if dlen is not None else None
        part = torch.empty(int(L.tfc_edge_bwd_blocks(E)), B, dtype=torch.float32, device=dev) if B else None
        if E:
            _lib.check(
                L.tfc_edge_encode_bwd2(
                    _ptr(vec), E, max(lmax, 0), 1 if sh_normalize else 0, _ptr(bw) if B else None, B, r_max, p, _ptr(dsh_c),
                    _ptr(demb_c), _ptr(dlen_c), _ptr(gv), _ptr(gb), _ptr(g_vec), _ptr(g_dsh), _ptr(g_demb), _ptr(g_dlen),
                    _ptr(part), _stream(),
                )
            )
        g_bw = None
        if bw is not None:
            g_bw = _sum_rows(part) if (part is not None and E) else torch.zeros_like(bw)
        return g_vec, g_bw, g_dsh, g_demb, g_dlen, (gv if has_dvec_in else None), None, None, None, None


def _edge_bwd(vec, length, lmax, sh_normalize, bw, r_max, p, dsh, demb, dvec_in, dlen, need_dbw):
    """(dvec, dbw): raw kernel, or - when the caller wants a graph of the backward - the differentiable op."""
    if torch.is_grad_enabled():
        dvec, dbw = _EdgeBwdFn.apply(vec, bw if demb is not None else None, dsh, demb, dlen, dvec_in, lmax, sh_normalize, r_max, p)
        if not need_dbw:
            dbw = None
        elif demb is None:
            dbw = torch.zeros_like(bw)
        return dvec, dbw
    return _edge_bwd_raw(vec, length, lmax, sh_normalize, bw, r_max, p, dsh, demb, dvec_in, dlen, need_dbw)


class _EdgeGeometryFn(torch.autograd.Function):
    """pos (+ periodic shifts) -> edge vectors, lengths and, optionally, spherical harmonics."""

    @staticmethod
    def forward(ctx, pos, shift, cell, batch, ordering: EdgeOrdering, lmax: int, sh_normalize: bool = True):
        pos = _f32(pos, "pos")
        ei = ordering.edge_index
        E, N = ordering.n_edges, pos.shape[0]
        dev = pos.device
        vec = torch.empty(E, 3, dtype=torch.float32, device=dev)
        length = torch.empty(E, dtype=torch.float32, device=dev)
        sh = torch.empty(E, (lmax + 1) ** 2, dtype=torch.float32, device=dev) if lmax >= 0 else None
        if shift is not None:
            shift = _f32(shift, "edge_cell_shift")
            cell = _f32(cell, "cell").reshape(-1, 3, 3)
            G = cell.shape[0]
            if G > 1:
                if batch is None:
                    raise ValueError("a batched cell needs `batch`")
                batch = batch.to(torch.int64).contiguous()
            else:
                batch = None
        else:
            cell, batch, G = None, None, 1
        if E:
            _lib.check(
                _lib.lib().tfc_edge_encode_fwd(
                    _ptr(pos), _ptr(ei), _ptr(shift), _ptr(cell), _ptr(batch), E, N, G, max(lmax, 0), 1 if sh_normalize else 0, None, 0,
                    1.0, 6.0, _ptr(vec), _ptr(length), _ptr(sh), None, _stream(),
                )
            )
        ctx.save_for_backward(vec, length)
        ctx.cell_args = (shift, batch, G, cell.shape) if shift is not None else None
        ctx.ordering, ctx.lmax, ctx.sh_normalize = ordering, lmax, bool(sh_normalize)
        ctx.n_nodes = N
        ctx.set_materialize_grads(False)
        if sh is None:
            return vec, length
        return vec, length, sh

    @staticmethod
    def backward(ctx, dvec_in, dlen, dsh=None):
        vec, length = ctx.saved_tensors
        E = vec.shape[0]
        dvec, _ = _edge_bwd(vec, length, ctx.lmax, ctx.sh_normalize, None, 1.0, 6.0, dsh, None, dvec_in, dlen, False)
        dcell = None
        if ctx.needs_input_grad[2] and ctx.cell_args is not None:
            shift, batch, G, _ = ctx.cell_args
            if not E:
                dcell = torch.zeros(G, 3, 3, dtype=torch.float32, device=vec.device)
            elif torch.is_grad_enabled():
                dcell = _DvecToDcellFn.apply(dvec, shift, batch, G, ctx.ordering, ctx.n_nodes)
            else:
                dcell = _dvec_to_dcell(dvec, shift, batch, G, ctx.ordering)
        return _dvec_to_dpos(dvec, ctx.ordering), None, dcell, None, None, None, None


class _DvecToDcellFn(torch.autograd.Function):
    """dcell[g] = sum over the edges of graph g of shift[e]^T (x) dvec[e], as a differentiable op.  It is linear in dvec
    and its transpose, dvec[e] = shift[e] . dcell[graph of e], is what the edge-vector kernel computes from zero
    positions - itself a differentiable op, so this goes to any order."""

    @staticmethod
    def forward(ctx, dvec, shift, batch, G: int, ordering: EdgeOrdering, n_nodes: int):
        ctx.args = (shift, batch, G, ordering, n_nodes)
        return _dvec_to_dcell(_f32(dvec, "dvec"), shift, batch, G, ordering)

    @staticmethod
    def backward(ctx, g):
        shift, batch, G, ordering, n_nodes = ctx.args
        zero = torch.zeros(n_nodes, 3, dtype=torch.float32, device=g.device)
        out = _EdgeGeometryFn.apply(zero, shift, g.reshape(-1, 3, 3), batch, ordering, -1, True)
        return out[0], None, None, None, None, None


def edge_geometry(pos, ordering: EdgeOrdering, shift=None, cell=None, batch=None, lmax: int = -1, sh_normalize: bool = True):
    """(vec, len) or (vec, len, sh) from positions; differentiable w.r.t. `pos` and, as the
    reference's graph is (eigenn/nn/_nequip.py:236-262), w.r.t. `cell` ([G,3,3])."""
    if cell is not None:
        cell = cell.reshape(-1, 3, 3)
    return _EdgeGeometryFn.apply(pos, shift, cell, batch, ordering, lmax, sh_normalize)


class _ShFn(torch.autograd.Function):
    """Spherical harmonics of caller-supplied edge vectors."""

    @staticmethod
    def forward(ctx, vec, lmax: int, normalize: bool = True):
        vec = _f32(vec, "edge_vectors")
        E = vec.shape[0]
        sh = torch.empty(E, (lmax + 1) ** 2, dtype=torch.float32, device=vec.device)
        if E:
            # len = null: the kernel takes the lengths from the vectors themselves
            _lib.check(
                _lib.lib().tfc_edge_encode_fwd(
                    None, None, None, None, None, E, 0, 1, lmax, 1 if normalize else 0, None, 0, 1.0, 6.0, _ptr(vec), None,
                    _ptr(sh), None, _stream(),
                )
            )
        ctx.save_for_backward(vec)
        ctx.lmax, ctx.normalize = lmax, bool(normalize)
        return sh

    @staticmethod
    def backward(ctx, dsh):
        (vec,) = ctx.saved_tensors
        dvec, _ = _edge_bwd(vec, None, ctx.lmax, ctx.normalize, None, 1.0, 6.0, dsh, None, None, None, False)
        return dvec, None, None


def spherical_harmonics(vec, lmax: int, normalize: bool = True):
    return _ShFn.apply(vec, lmax, normalize)


class _RadialFn(torch.autograd.Function):
    """Bessel basis x polynomial cutoff of the edge length.  The length enters through the
    edge vector, so the gradient is returned on `vec` directly."""

    @staticmethod
    def forward(ctx, vec, length, bessel_weights, r_max: float, p: float):
        vec, length, bw = _f32(vec, "edge_vectors"), _f32(length, "edge_lengths"), _f32(bessel_weights, "bessel_weights")
        E, B = vec.shape[0], bw.shape[0]
        emb = torch.empty(E, B, dtype=torch.float32, device=vec.device)
        if E:
            _lib.check(
                _lib.lib().tfc_edge_encode_fwd(
                    None, None, None, None, None, E, 0, 1, 0, 1, _ptr(bw), B, float(r_max), float(p), _ptr(vec),
                    _ptr(length), None, _ptr(emb), _stream(),
                )
            )
        ctx.save_for_backward(vec, length, bw)
        ctx.r_max, ctx.p = float(r_max), float(p)
        return emb

    @staticmethod
    def backward(ctx, demb):
        vec, length, bw = ctx.saved_tensors
        dvec, dbw = _edge_bwd(vec, length, 0, True, bw, ctx.r_max, ctx.p, None, demb, None, None, ctx.needs_input_grad[2])
        return dvec, None, dbw, None, None


def radial_embedding(vec, length, bessel_weights, r_max: float, p: float = 6.0):
    return _RadialFn.apply(vec, length, bessel_weights, r_max, p)


# ======================================================================================
# dense pieces on the fp32 matrix cores
# ======================================================================================


@functools.lru_cache(maxsize=4096)
def _imap(d1=1, d2=1, s0=0, s1=0, s2=0) -> IMap:
    """(cached: the maps are read-only once built, and a step builds thousands of them)"""
    return IMap(int(d1), int(d2), int(s0), int(s1), int(s2))


def _lin(stride) -> IMap:  # f(t) = t * stride
    return _imap(1, 1, stride, 0, 0)


@functools.lru_cache(maxsize=4096)
def _ws_floats(m: int, n: int, k: int) -> int:
    return int(_lib.lib().tfc_gemm_workspace_floats(m, n, k))


_GEMM_WS: "dict[tuple, torch.Tensor]" = _tensor_cache()


def _gemm_workspace(n_floats: int) -> torch.Tensor:
    """Scratch for split-K, one per (device, stream), reused by every call on that stream (calls on
    one stream are ordered; two streams must not share it)."""
    dev = torch.cuda.current_device()
    key = (dev, torch._C._cuda_getCurrentRawStream(dev))
    ws = _GEMM_WS.get(key)
    if ws is None or ws.numel() < n_floats:
        ws = torch.empty(max(n_floats, 1 << 22), dtype=torch.float32, device=torch.device("cuda", dev))
        _GEMM_WS[key] = ws
    return ws


# The radial network of a layer depends on the edge embedding only, not on the node features: its
# forward can run beside linear_1, and - because autograd runs a node's backward on the stream of
# its forward - its backward (three MFMA-bound products per layer) runs beside the HBM-bound
# convolution backward of the layer below.  (The self connection, which depends on the layer's
# input only, was tried there too: its products then share the CUs with the convolution backward,
# which slows down by more than is gained - config 3: 3.71 -> 3.85 ms.)  TFC_MLP_SIDE_STREAM=0 keeps
# everything on one stream.
MLP_SIDE_STREAM = _os_environ.get("TFC_MLP_SIDE_STREAM", "1") != "0"
# the convolution kernels also write the channel-contiguous copy of their result that the next o3.Linear reads
CONV_TRANSPOSED = _os_environ.get("TFC_CONV_TRANSPOSED", "1") != "0"
# the next layer's radial network right behind the current one's (TFC_MLP_PREFETCH=0: each layer its own)
MLP_PREFETCH = _os_environ.get("TFC_MLP_PREFETCH", "1") != "0"
# layers computed ahead.  1: more would not shorten the step (config 3: 3.44 ms either way) and moves the
# radial networks' backward, through autograd's ordering, onto the convolution backward of earlier layers
MLP_PREFETCH_AFTER_CONV = _os_environ.get("TFC_MLP_PREFETCH_AFTER_CONV", "0") != "0"
# who records the autograd node of a prefetched radial network: the prefetching layer (1, default: its backward then
# runs at the end of the pass, after the earlier layer's convolution backward) or the layer it belongs to (0: only
# the launches are issued early; the backward runs beside the next convolution backward, as without prefetch).
# Measured at config 3 (graph replay / eager): 3.39 / 3.53 ms against 3.50 / 3.69 - the convolution backward loses
# more to the shared CUs than the exposed tail of two small networks costs
MLP_PREFETCH_RECORD_EARLY = _os_environ.get("TFC_MLP_PREFETCH_RECORD_EARLY", "1") != "0"
# the radial networks of a pass read the edge embedding through one alias recorded under the side stream, so that
# their gradients to it are summed there and not, one by one, on the main stream (nn/point_conv.py:_embedding_hub)
MLP_EMBEDDING_HUB = _os_environ.get("TFC_MLP_EMBEDDING_HUB", "1") != "0"
MLP_PREFETCH_DEPTH = int(_os_environ.get("TFC_MLP_PREFETCH_DEPTH", "1"))
_SIDE_STREAMS: "dict[int, torch.cuda.Stream]" = {}


def side_stream(device: torch.device) -> "torch.cuda.Stream":
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _SIDE_STREAMS.get(idx)
    if st is None:
        st = _SIDE_STREAMS[idx] = torch.cuda.Stream(device=idx)
    return st


def _desc(m, n, k, a1, b, c, a1_r, a1_k, b_k, b_j, c_r, c_j, alpha=1.0, accumulate=False, a2=None, a2_r=None, a2_k=None,
          epilogue=_lib.EPI_NONE, aux=None, act_cst=1.0, tag=None, a_rep=1, a1_rs=0, a2_rs=0) -> GemmDesc:
    d = GemmDesc()
    d.a_rep, d.a1_rep_stride, d.a2_rep_stride = int(a_rep), int(a1_rs), int(a2_rs)
    d.tag = tag
    d.m, d.n, d.k = int(m), int(n), int(k)
    d.a1, d.b, d.c = a1, b, c
    d.a2 = a2
    d.a1_r, d.a1_k = a1_r, a1_k
    d.a2_r = a2_r if a2_r is not None else _lin(0)
    d.a2_k = a2_k if a2_k is not None else _lin(0)
    d.b_k, d.b_j, d.c_r, d.c_j = b_k, b_j, c_r, c_j
    d.alpha = float(alpha)
    d.accumulate = 1 if accumulate else 0
    d.epilogue = int(epilogue)
    d.aux = aux
    d.act_cst = float(act_cst)
    return d


def _gemm(*args, **kwargs):
    """One problem, launched now."""
    _gemm_group([_desc(*args, **kwargs)])


def _gemm_group(descs: "List[GemmDesc]"):
    """Independent problems (disjoint outputs) in as few launches as possible (tfc_gemm_group)."""
    if not descs:
        return
    L = _lib.lib()
    need = sum(_ws_floats(d.m, d.n, d.k) for d in descs)
    arr = (GemmDesc * len(descs))(*descs)
    if need > 0:
        ws = _gemm_workspace(need)
        for d in arr:
            d.workspace, d.workspace_floats = ws.data_ptr(), ws.numel()
    _lib.check(L.tfc_gemm_group(arr, len(descs), _stream()))


def _gemm_waves(items):
    """items: (output key, desc factory taking `accumulate`).  Problems that write the same output
    block must not share a launch: the n-th writer of a block goes to the n-th group."""
    waves: "List[List[GemmDesc]]" = []
    count = {}
    for key, make in items:
        w = count.get(key, 0)
        count[key] = w + 1
        while len(waves) <= w:
            waves.append([])
        waves[w].append(make(w > 0))
    for group in waves:
        _gemm_group(group)


_TRANSPOSED: "dict[tuple, tuple]" = _tensor_cache()


@functools.lru_cache(maxsize=256)
def _ir_blocks(blocks):
    if not blocks:
        return None
    arr = (IrBlock * len(blocks))()
    for a, (off, mul, dim) in zip(arr, blocks):
        a.off, a.mul, a.dim = off, mul, dim
    return arr


def _transposed_key(t: torch.Tensor):
    return (t.data_ptr(), t._version, t.shape[0], t.shape[1])


def _transposed_put(t: torch.Tensor, blocks, out: torch.Tensor):
    """Remember `out` as the channel-contiguous copy of the `blocks` of the rows of `t` (made by _ir_transpose or
    written by the kernel that produced `t`).  The last three are kept, each with its source, so that the
    source's address cannot be reused while the entry lives."""
    key = _transposed_key(t)
    _TRANSPOSED.pop(key, None)
    while len(_TRANSPOSED) >= 3:
        _TRANSPOSED.pop(next(iter(_TRANSPOSED)))
    cur = torch.cuda.current_stream(t.device)
    ev = None
    if MLP_SIDE_STREAM:  # the consumers of a tensor may run on different streams
        ev = torch.cuda.Event()
        ev.record(cur)
    _TRANSPOSED[key] = (t.detach(), out, ev, cur, frozenset(blocks))


def _transposed_get(t: torch.Tensor, blocks):
    hit = _TRANSPOSED.get(_transposed_key(t))
    if hit is None or not frozenset(blocks) <= hit[4]:
        return None
    cur = torch.cuda.current_stream(t.device)
    if hit[2] is not None and hit[3] != cur:  # made on another stream: order this one after it
        cur.wait_event(hit[2])
        hit[1].record_stream(cur)
    return hit[1]


def _ir_transpose(t: torch.Tensor, blocks) -> torch.Tensor:
    """Channel-contiguous copy of the (off, mul, dim) blocks of the rows of `t` (tfc_ir_transpose):
    block b of the flat result starts at off_b * N and is laid out [dim_b][N][mul_b].  Results are kept for a
    while (_transposed_put): the gradient that reaches the self connection and linear_2 of a layer is one
    tensor, transposed once, and the convolution kernels hand the copies of their own results over this way."""
    n, d_row = t.shape
    blocks = tuple(sorted(set(blocks)))
    hit = _transposed_get(t, blocks)
    if hit is not None:
        return hit
    arr = (IrBlock * len(blocks))()
    for a, (off, mul, dim) in zip(arr, blocks):
        a.off, a.mul, a.dim = off, mul, dim
    out = torch.empty(max(n * d_row, 1), dtype=torch.float32, device=t.device)
    _lib.check(_lib.lib().tfc_ir_transpose(_ptr(t), n, d_row, arr, len(blocks), _ptr(out), _stream()))
    _transposed_put(t, blocks, out)
    return out


def _p(t: torch.Tensor, off: int = 0):
    return t.data_ptr() + 4 * int(off)


# ------------------------------------------------------------------ o3.Linear


@dataclass(frozen=True)
class LinearSpec:
    """o3.Linear as a list of instructions (in_off, mul_in, out_off, mul_out, dim, w_off, path_weight)."""

    d_in: int
    d_out: int
    weight_numel: int
    instructions: Tuple[Tuple[int, int, int, int, int, int, float], ...]
    out_covered: bool  # every output column is written by some instruction
    in_covered: bool
    out_blocks: Tuple[Tuple[int, int, int], ...] = ()  # every (off, mul, dim) block of the output row


@functools.lru_cache(maxsize=256)
def _linear_ins(spec: "LinearSpec"):
    arr = (LinearIns * len(spec.instructions))()
    for a, (in_off, mi, out_off, mo, d, w_off, pw) in zip(arr, spec.instructions):
        a.in_off, a.mul_in, a.out_off, a.mul_out, a.dim, a.w_off, a.path_weight = in_off, mi, out_off, mo, d, w_off, pw
    return arr


def _linear_in_blocks(spec: "LinearSpec"):
    return tuple(sorted({(in_off, mi, d) for in_off, mi, _, _, d, _, _ in spec.instructions}))


def _linear_bwd_raw(spec: "LinearSpec", n: int, xt, g, weight, need_dx: bool, need_dw: bool):
    """(dx, dw) of o3.Linear from the channel-contiguous copy `xt` of its input and the output gradient."""
    g = _f32(g, "grad_output")
    dx = dw = None
    if need_dx:
        dx = (torch.empty if spec.in_covered else torch.zeros)(n, spec.d_in, dtype=torch.float32, device=g.device)
    if need_dw:
        dw = torch.empty(spec.weight_numel, dtype=torch.float32, device=g.device)
    if not spec.instructions:
        return dx, (dw.zero_() if dw is not None else None)
    # (all blocks of the row, not only the ones reached: the copy is then shared with the other
    # consumer of the same gradient, see _ir_transpose)
    gt = _ir_transpose(g, spec.out_blocks or [(out_off, mo, d) for _, _, out_off, mo, d, _, _ in spec.instructions])
    ins = _linear_ins(spec)
    L = _lib.lib()
    need = int(L.tfc_linear_workspace_floats(ins, len(ins), n)) if dw is not None else 0
    ws = _gemm_workspace(need) if need > 0 else None
    _lib.check(L.tfc_linear_bwd(ins, len(ins), _ptr(xt), None, _ptr(weight), n, spec.d_in, spec.d_out, spec.weight_numel,
                                _ptr(gt), 1, _ptr(dx), _ptr(dw), _ptr(ws), ws.numel() if ws is not None else 0, _stream()))
    return dx, dw


class _LinearFn(torch.autograd.Function):
    """o3.Linear through tfc_linear_fwd / tfc_linear_bwd: the products run on channel-contiguous copies
    of the input and of the output gradient, every instruction, whatever its l, being the same plain
    contraction over channels, all of them in one grouped launch."""

    @staticmethod
    def forward(ctx, x, weight, spec: LinearSpec):
        x, weight = _f32(x, "x"), _f32(weight, "weight")
        n = x.shape[0]
        if x.shape[1] != spec.d_in or weight.numel() != spec.weight_numel:
            raise ValueError(f"linear: x{tuple(x.shape)} / weight[{weight.numel()}] do not match the irreps ({spec.d_in}, {spec.weight_numel})")
        alloc = torch.empty if spec.out_covered else torch.zeros
        out = alloc(n, spec.d_out, dtype=torch.float32, device=x.device)
        xt = _transposed_get(x, _linear_in_blocks(spec)) if spec.instructions and n else None
        ready = xt is not None  # the kernel that produced x wrote the copy as well
        if xt is None:
            xt = torch.empty(max(n * spec.d_in, 1), dtype=torch.float32, device=x.device)
        if spec.instructions:
            ins = _linear_ins(spec)
            _lib.check(_lib.lib().tfc_linear_fwd_t(ins, len(ins), _ptr(x), _ptr(weight), n, spec.d_in, spec.d_out,
                                                   spec.weight_numel, _ptr(xt), 1 if ready else 0, _ptr(out), _stream()))
        ctx.save_for_backward(xt, weight, x)
        ctx.spec, ctx.n = spec, n
        return out

    @staticmethod
    def backward(ctx, g):
        xt, weight, x = ctx.saved_tensors
        spec, n = ctx.spec, ctx.n
        if torch.is_grad_enabled():  # create_graph=True
            dx, dw = _LinearBwdFn.apply(g, x, weight, spec, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
            return dx, (dw.view_as(weight) if dw is not None else None), None
        dx, dw = _linear_bwd_raw(spec, n, xt, g, weight, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dx, (dw.view_as(weight) if dw is not None else None), None


class _LinearBwdFn(torch.autograd.Function):
    """(dx, dw) of o3.Linear as a differentiable op.  The layer is bilinear in (x, W), so the derivatives of
    its two transposes are the layer and its transposes again (same kernels, any order)."""

    @staticmethod
    def forward(ctx, g, x, weight, spec: LinearSpec, need_dx: bool, need_dw: bool):
        g = _f32(g, "grad_output")
        n = g.shape[0]
        xt = None
        if need_dw:
            x = _f32(x, "x")
            xt = _ir_transpose(x, _linear_in_blocks(spec)) if spec.instructions else None
        if need_dx:
            weight = _f32(weight, "weight")
        dx, dw = _linear_bwd_raw(spec, n, xt, g, weight if need_dx else None, need_dx, need_dw)
        ctx.save_for_backward(g, x if need_dw else None, weight if need_dx else None)
        ctx.spec, ctx.need = spec, (need_dx, need_dw)
        ctx.set_materialize_grads(False)
        return dx, dw

    @staticmethod
    def backward(ctx, ggx, ggw):
        g, x, weight = ctx.saved_tensors
        spec = ctx.spec
        g_g = g_x = g_w = None
        if ggx is not None:  # dx = g W^T
            g_g = _add(g_g, _LinearFn.apply(ggx, weight, spec))
            g_w = _LinearBwdFn.apply(g, ggx, None, spec, False, True)[1].view_as(weight)
        if ggw is not None:  # dw = x^T g
            g_g = _add(g_g, _LinearFn.apply(x, ggw.reshape(-1), spec))
            g_x = _LinearBwdFn.apply(g, None, ggw.reshape(-1), spec, True, False)[0]
        return g_g, g_x, g_w, None, None, None


def linear(x, weight, spec: LinearSpec):
    return _LinearFn.apply(_tracked(x), _tracked(weight), spec)


# ------------------------------------------------------------------ radial FullyConnectedNet


def _radial_mlp_run(x, ws, act_cst: float):
    """The forward launches of the radial network: [x, h_1, ..., h_{L-1}, out] (no autograd)."""
    E = x.shape[0]
    hs = [x]
    dims = [x.shape[1]] + [w.shape[1] for w in ws]
    for i, w in enumerate(ws):
        if w.shape[0] != dims[i]:
            raise ValueError(f"radial mlp: layer {i} expects {w.shape[0]} inputs, got {dims[i]}")
    nl = len(ws)
    if 2 <= nl <= 6 and max(dims[:-1]) <= 64 and dims[-1] % 4 == 0 and ws[-1].data_ptr() % 16 == 0:
        # whole network in one launch, hidden activations kept in LDS (tfc_radial_mlp_fwd)
        outs = [torch.empty(E, d, dtype=torch.float32, device=x.device) for d in dims[1:]]
        c_dims = (C.c_int32 * (nl + 1))(*dims)
        c_w = (C.c_void_p * nl)(*[w.data_ptr() for w in ws])
        c_h = (C.c_void_p * nl)(*([o.data_ptr() for o in outs[:-1]] + [None]))
        need = int(_lib.lib().tfc_radial_mlp_fwd_workspace_floats(nl, c_dims))
        wsp = _gemm_workspace(need)  # images of W_L for the bf16-piece kernel
        _lib.check(_lib.lib().tfc_radial_mlp_fwd_ws(_ptr(x), E, nl, c_dims, c_w, float(act_cst), c_h, _ptr(outs[-1]), _ptr(wsp),
                                                    wsp.numel(), _stream()))
        hs += outs
    else:
        for i, w in enumerate(ws):
            hin, hout = w.shape
            out = torch.empty(E, hout, dtype=torch.float32, device=x.device)
            last = i == nl - 1
            _gemm(
                E, hout, hin, _p(hs[-1]), _p(w), _p(out), a1_r=_lin(hin), a1_k=_lin(1), b_k=_lin(hout), b_j=_lin(1),
                c_r=_lin(hout), c_j=_lin(1), alpha=1.0 / math.sqrt(hin),
                epilogue=_lib.EPI_NONE if last else _lib.EPI_SSP, act_cst=act_cst, tag=b"gemm_mlp_fwd",
            )
            hs.append(out)
    return hs


class _RadialMlpFn(torch.autograd.Function):
    """x -> act'(x W0/sqrt(h0)) -> ... -> h W_last/sqrt(h_last); act' = cst * ssp.

    `pre`: the activations [h_1, ..., out] of exactly this input and these weights, computed earlier by
    radial_mlp_precompute (e.g. on a side stream, ahead of time); the call then launches nothing and only
    records the autograd node - at the place in the forward order, and under the current stream, its
    backward should have."""

    @staticmethod
    def forward(ctx, x, act_cst: float, pre, *weights):
        x = _f32(x, "edge_embedding")
        ws = [_f32(w, "mlp weight") for w in weights]
        hs = [x] + list(pre) if pre is not None else _radial_mlp_run(x, ws, act_cst)
        if len(hs) != len(ws) + 1 or hs[-1].shape != (x.shape[0], ws[-1].shape[1]):
            raise ValueError("radial mlp: precomputed activations do not match the network")
        ctx.save_for_backward(*hs[:-1], *ws)
        ctx.n_layers, ctx.act_cst = len(ws), float(act_cst)
        return hs[-1]

    @staticmethod
    def backward(ctx, g):
        nl = ctx.n_layers
        hs, ws = ctx.saved_tensors[:nl], ctx.saved_tensors[nl:]  # hs[0] = x, hs[i] = output of layer i-1
        if torch.is_grad_enabled():
            # create_graph=True: rebuild the network from ops that are differentiable to any order (dense products
            # and the shifted softplus as a one-segment gate, all on the same kernels) and let autograd derive a
            # backward that can be differentiated again
            need = [i for i in range(1 + nl) if ctx.needs_input_grad[0 if i == 0 else 2 + i]]
            if not need:
                return (None,) * (3 + nl)
            with torch.enable_grad():
                out = _radial_mlp_composite(hs[0], ws, ctx.act_cst)
                grads = torch.autograd.grad(out, [([hs[0]] + list(ws))[i] for i in need], g, create_graph=True, allow_unused=True)
            res = [None] * (1 + nl)
            for i, gr in zip(need, grads):
                res[i] = gr
            return (res[0], None, None, *res[1:])
        g = _f32(g, "grad_output")
        E = g.shape[0]
        dev = g.device
        dims = [hs[0].shape[1]] + [w.shape[1] for w in ws]
        dws = [torch.empty_like(w) if ctx.needs_input_grad[3 + i] else None for i, w in enumerate(ws)]
        dhid = [torch.empty(E, dims[i + 1], dtype=torch.float32, device=dev) for i in range(nl - 1)]
        dx = torch.empty(E, dims[0], dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        L = _lib.lib()
        c_dims = (C.c_int32 * (nl + 1))(*dims)
        need = int(L.tfc_radial_mlp_bwd_workspace_floats(E, nl, c_dims))
        wsp = _gemm_workspace(need) if need > 0 else None
        vp = lambda ts: (C.c_void_p * max(len(ts), 1))(*[None if t is None else t.data_ptr() for t in ts])  # noqa: E731
        _lib.check(L.tfc_radial_mlp_bwd(
            _ptr(hs[0]), E, nl, c_dims, vp(ws), ctx.act_cst, vp(hs[1:]), _ptr(g), vp(dhid), _ptr(dx), vp(dws),
            _ptr(wsp), wsp.numel() if wsp is not None else 0, _stream(),
        ))
        return (dx, None, None, *dws)


class _MatmulFn(torch.autograd.Function):
    """C = alpha op(a) op(b) on the fp32 MFMA GEMM; a, b row-major 2-D, op = transpose when flagged.  Its
    backward is two more such products."""

    @staticmethod
    def forward(ctx, a, b, ta: bool, tb: bool, alpha: float):
        a, b = _f32(a, "a"), _f32(b, "b")
        m, k = (a.shape[1], a.shape[0]) if ta else (a.shape[0], a.shape[1])
        k2, n = (b.shape[1], b.shape[0]) if tb else (b.shape[0], b.shape[1])
        if k != k2:
            raise ValueError(f"matmul: inner sizes {k} and {k2} differ")
        c = torch.empty(m, n, dtype=torch.float32, device=a.device)
        if m and n:
            if k == 0:
                c.zero_()
            else:
                _gemm(
                    m, n, k, _p(a), _p(b), _p(c),
                    a1_r=_lin(1) if ta else _lin(a.shape[1]), a1_k=_lin(a.shape[1]) if ta else _lin(1),
                    b_k=_lin(1) if tb else _lin(b.shape[1]), b_j=_lin(b.shape[1]) if tb else _lin(1),
                    c_r=_lin(n), c_j=_lin(1), alpha=alpha, tag=b"gemm_matmul",
                )
        ctx.save_for_backward(a, b)
        ctx.args = (ta, tb, float(alpha))
        return c

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ta, tb, alpha = ctx.args
        da = db = None
        if ctx.needs_input_grad[0]:
            da = _MatmulFn.apply(b, g, tb, True, alpha) if ta else _MatmulFn.apply(g, b, False, not tb, alpha)
        if ctx.needs_input_grad[1]:
            db = _MatmulFn.apply(g, a, True, ta, alpha) if tb else _MatmulFn.apply(a, g, not ta, False, alpha)
        return da, db, None, None, None


def _radial_mlp_composite(x, ws, act_cst: float, act_id: Optional[int] = None):
    """The radial network out of ops that are differentiable to any order (used for double backward, and as the
    forward of networks whose activation is not the shifted softplus)."""
    act_id = _lib.ACT["ssp"] if act_id is None else int(act_id)
    h = x
    for i, w in enumerate(ws):
        h = _MatmulFn.apply(h, w, False, False, 1.0 / math.sqrt(w.shape[0]))
        if i < len(ws) - 1:
            d = w.shape[1]
            h = _GateFn.apply(h, GateSpec(d_in=d, d_out=d, segs=((0, 0, d, 1, -1, act_id, float(act_cst)),)))
    return h


def radial_mlp_general(x, weights: Sequence[torch.Tensor], act_cst: float, act_id: int):
    """FullyConnectedNet with any of the gate kernel's activations (eigenn/nn/utils.py:182 `mlp_activation`), layer by
    layer: dense product, then the activation as a one-segment gate."""
    return _radial_mlp_composite(x, list(weights), act_cst, act_id)


_SELF_ORDERINGS: "dict[tuple, tuple]" = _tensor_cache()


def self_ordering(n: int, device) -> "EdgeOrdering":
    """Ordering of the graph in which node i has the one edge i -> i (general FullyConnectedTensorProduct, nn/e3.py)."""
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), int(n))
    hit = _SELF_ORDERINGS.get(key)
    if hit is None:
        ar = torch.arange(n, dtype=torch.int64, device=device)
        ei = torch.stack([ar, ar])
        hit = (ei, EdgeOrdering(ei, n))
        if len(_SELF_ORDERINGS) >= 8:
            _SELF_ORDERINGS.pop(next(iter(_SELF_ORDERINGS)))
        _SELF_ORDERINGS[key] = hit
    return hit[1]


def radial_mlp(x, weights: Sequence[torch.Tensor], act_cst: float, pre=None):
    return _RadialMlpFn.apply(_tracked(x), act_cst, pre, *[_tracked(w) for w in weights])


def radial_mlp_precompute(x, weights: Sequence[torch.Tensor], act_cst: float):
    """The network's activations [h_1, ..., out] for `radial_mlp(..., pre=...)`: launches on the current stream,
    records nothing for autograd."""
    with torch.no_grad():
        return _radial_mlp_run(_f32(x.detach(), "edge_embedding"), [_f32(w.detach(), "mlp weight") for w in weights], act_cst)[1:]


# ------------------------------------------------------------------ FullyConnectedTensorProduct with scalar second input


@dataclass(frozen=True)
class FctpSpec:
    """Self connection: instructions (x_off, mul_u, a_off, mul_v, out_off, mul_w, dim, w_off, alpha)
    with out[n,w,k] += alpha * sum_{u,v} W[u,v,w] x[n,u,k] a[n,v]  (second input scalar, l2 = 0)."""

    d_x: int
    d_a: int
    d_out: int
    weight_numel: int
    instructions: Tuple[Tuple[int, int, int, int, int, int, int, int, float], ...]
    out_covered: bool
    x_covered: bool
    a_covered: bool
    out_blocks: Tuple[Tuple[int, int, int], ...] = ()  # every (off, mul, dim) block of the output row


@functools.lru_cache(maxsize=512)
def _fctp_ins(ins_list):
    arr = (FctpIns * len(ins_list))()
    for q, (xo, mu, ao, mv, oo, mw, d, wo, alpha) in zip(arr, ins_list):
        q.x_off, q.mul_u, q.a_off, q.mul_v, q.out_off, q.mul_w, q.dim, q.w_off, q.alpha = xo, mu, ao, mv, oo, mw, d, wo, alpha
    return arr


_FCTP_WS: "dict[tuple, int]" = {}


def _fctp_workspace(ins, n):
    key = (id(ins), n)  # (the instruction arrays are cached for the life of the process)
    need = _FCTP_WS.get(key)
    if need is None:
        need = _FCTP_WS[key] = int(_lib.lib().tfc_fctp_workspace_floats(ins, len(ins), n))
    return _gemm_workspace(need) if need > 0 else None


def _fctp_fwd(n, x, a, d_a, weight, out, spec, ins_list, on_base):
    """out[(n,k), w] (+)= alpha sum_{(u,v)} x[n,u,k] a[n,v] W[(u,v), w] for every instruction (tfc_fctp_fwd)."""
    ins = _fctp_ins(tuple(ins_list))
    ws = _fctp_workspace(ins, n)
    _lib.check(_lib.lib().tfc_fctp_fwd(ins, len(ins), _ptr(x), _ptr(a), _ptr(weight), n, spec.d_x, d_a, spec.d_out, weight.numel(),
                                       1 if on_base else 0, _ptr(out), _ptr(ws), ws.numel() if ws is not None else 0, _stream()))


def _fctp_bwd(n, x, a, d_a, weight, gt, spec, ins_list, dx=None, dw=None, da=None):
    """tfc_fctp_bwd on a ready-made channel-contiguous gradient copy `gt`: any of
    dx[(n,k), u] = alpha sum_{(v,w)} a[n,v] g[n,w,k] W[u,v,w],
    dW[(u,v), w] = alpha sum_{(n,k)} x[n,u,k] a[n,v] g[n,w,k],
    da[n, v]     = alpha sum_{(k,u,w)} x[n,u,k] g[n,w,k] W[u,v,w]  (per-instruction partials, ordered sum)."""
    ins = _fctp_ins(tuple(ins_list))
    L = _lib.lib()
    xt = part = None
    if da is not None:
        xt = torch.empty(max(n * spec.d_x, 1), dtype=torch.float32, device=x.device)
    if da is not None:
        part = torch.empty(max(len(ins) * n * d_a, 1), dtype=torch.float32, device=x.device)
    ws = _fctp_workspace(ins, n)
    _lib.check(L.tfc_fctp_bwd(ins, len(ins), _ptr(x), _ptr(a), _ptr(weight), None, n, spec.d_x, d_a, spec.d_out, weight.numel(),
                              _ptr(gt), 1, _ptr(dx), _ptr(dw), _ptr(da), _ptr(xt), _ptr(part), _ptr(ws),
                              ws.numel() if ws is not None else 0, _stream()))


ROUTE_COUNTS = {"fctp_dense": 0, "fctp_low_rank": 0}  # forward calls per self-connection route (read by tests)
_ATTR_FACTORS: "dict[tuple, tuple]" = _tensor_cache()
_ATTR_CLASSES: "dict[tuple, tuple]" = _tensor_cache()
_HASH_COEF: "dict[tuple, torch.Tensor]" = _tensor_cache()


class AttrFactors(NamedTuple):
    """a = C @ R (attr_factors).  `promised`: the caller supplied the classes and thereby promised
    that `a` is a per-class row table looked up by class, which lets the backward return the
    gradient of `a` projected onto such tensors (the class mean) instead of the per-node one."""
    C: torch.Tensor          # [N, r] one-hot, fp32
    R: torch.Tensor          # [r, d_a] distinct rows (zero padded)
    inv: torch.Tensor        # [N] int64 class of each row
    inv_count: torch.Tensor  # [r] fp32, 1 / rows in the class (0-size classes: 1)
    promised: bool


def _class_structure(labels: torch.Tensor):
    """(inv [N] dense class ids, rep [u] first row of each class, one-hot C [N, r], u, r, 1 / class sizes [r]) of
    an integer labelling; one host sync (the number of classes)."""
    n = labels.shape[0]
    uniq, inv = torch.unique(labels, return_inverse=True)  # sync
    u = int(uniq.numel())
    r = max(4, (u + 3) // 4 * 4)
    ar = torch.arange(n, device=labels.device)
    rep = torch.full((u,), n, dtype=torch.int64, device=labels.device)
    rep.scatter_reduce_(0, inv, ar, reduce="amin")
    Cm = torch.zeros(n, r, dtype=torch.float32, device=labels.device)
    Cm[ar, inv] = 1.0
    return inv, _rep_rows(rep, n, r), Cm, u, r, 1.0 / Cm.sum(0).clamp_min(1.0)


def _rep_rows(rep: torch.Tensor, n: int, r: int):
    """(row index [r], mask [r, 1]) from the first row of each class: classes without a member (rep == n) and the
    padding up to r point at row 0 with mask 0, so that the table of distinct rows is `a[idx] * mask` - one gather and
    one multiply per step instead of six small launches."""
    idx = torch.zeros(r, dtype=torch.int64, device=rep.device)
    idx[: rep.numel()] = rep.clamp_max(max(n - 1, 0))
    mask = torch.zeros(r, 1, dtype=torch.float32, device=rep.device)
    mask[: rep.numel(), 0] = (rep < n).to(torch.float32)
    return idx, mask


def _class_structure_static(labels: torch.Tensor, u: int):
    """_class_structure for labels known to lie in [0, u): one library call (tfc_class_table), no host
    synchronisation - it runs inside captured steps (classes without a member get a zero row and weight 1)."""
    n = labels.shape[0]
    r = max(4, (u + 3) // 4 * 4)
    dev = labels.device
    labels = labels.contiguous()
    work = torch.empty(2 * r, dtype=torch.int32, device=dev)
    Cm = torch.empty(n, r, dtype=torch.float32, device=dev)
    idx = torch.empty(r, dtype=torch.int64, device=dev)
    mask = torch.empty(r, 1, dtype=torch.float32, device=dev)
    inv_count = torch.empty(r, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().tfc_class_table(_ptr(labels), n, int(u), r, _ptr(work), _ptr(Cm), _ptr(idx), _ptr(mask), _ptr(inv_count),
                                          _stream()))
    return labels, (idx, mask), Cm, u, r, inv_count


def attr_factors(a: torch.Tensor, classes: Optional[torch.Tensor] = None):
    """(C [N, r], R [r, d_a]) with a == C @ R exactly, when the rows of `a` are copies of a few distinct
    rows (node attributes that are a species embedding: eigenn/nn/embedding.py:59-75 sets NODE_ATTRS
    to embedding(species), so at most n_species distinct rows) - else None.  C is the one-hot matrix
    of the row classes, R the distinct rows (zero padded to r, a multiple of 4).

    `classes` (keys.NODE_ATTRS_CLASSES): int64 [N], equal class => identical row, supplied by the
    caller and trusted; the class structure is then cached per `classes` tensor and a call costs
    one row gather.  Without it the classes are discovered from the bit patterns of `a` and
    verified (two host syncs), once per tensor: the result is cached on (address, version) with
    the tensor held alive, so the layers of a model, forward and backward, share it."""
    n, d_a = a.shape
    if n < 64 or d_a < 8 or not a.is_contiguous():
        return None
    if classes is not None:
        ckey = (classes.data_ptr(), classes._version, n)
        hit = _ATTR_CLASSES.get(ckey)
        if hit is None:
            n_known = getattr(classes, "tfc_num_classes", None)  # set by whoever made the classes (SpeciesEmbedding)
            with torch.no_grad():
                lab = classes.reshape(-1).to(torch.int64)
                st = _class_structure_static(lab, int(n_known)) if n_known else _class_structure(lab)
            if len(_ATTR_CLASSES) >= 8:
                _ATTR_CLASSES.pop(next(iter(_ATTR_CLASSES)))
            hit = (classes, st)
            _ATTR_CLASSES[ckey] = hit
        inv, rep, Cm, u, r, inv_count = hit[1]
        if 2 * r > d_a:
            return None
        key = (a.data_ptr(), a._version, tuple(a.shape), ckey)
        fhit = _ATTR_FACTORS.get(key)
        if fhit is not None:
            return fhit[1]
        with torch.no_grad():
            R = a.detach().index_select(0, rep[0]) * rep[1]  # (a class without a member: zero row)
        res = AttrFactors(Cm, R, inv, inv_count, True)  # (the class structure, sizes included, is cached per `classes`)
    else:
        key = (a.data_ptr(), a._version, tuple(a.shape), None)
        fhit = _ATTR_FACTORS.get(key)
        if fhit is not None:
            return fhit[1]
        res = None
        with torch.no_grad():
            ck = (a.device, d_a)
            coef = _HASH_COEF.get(ck)
            if coef is None:
                gen = torch.Generator().manual_seed(0x5EED)
                coef = (torch.randint(1, 1 << 40, (d_a,), generator=gen, dtype=torch.int64) * 2 + 1).to(a.device)
                _HASH_COEF[ck] = coef
            h = (a.detach().view(torch.int32).to(torch.int64) * coef).sum(dim=1)
            inv, rep, Cm, u, r, inv_count = _class_structure(h)
            if 2 * r <= d_a:
                R = a.detach().index_select(0, rep[0]) * rep[1]
                if bool((R[inv] == a.detach()).all().item()):  # sync; also rejects NaN rows
                    res = AttrFactors(Cm, R, inv, inv_count, False)
    if len(_ATTR_FACTORS) >= 8:
        _ATTR_FACTORS.pop(next(iter(_ATTR_FACTORS)))
    _ATTR_FACTORS[key] = (a, res)
    return res


def _rows_kernels_ok(spec) -> bool:
    # (measured: 9.5 against 13.4 us per launch at config 3, 1.6 M weights; 58 against 42 us at config 4, 16.8 M weights,
    # where the GEMM's tiles are no longer mostly padding - the short kernels are taken below 4 M weights)
    return (len(spec.instructions) <= 32 and all(ins[3] <= 256 for ins in spec.instructions)
            and spec.weight_numel <= (1 << 22))


def _lowrank_w(spec, low_ins, R, r, weight, wp):
    """W'[u, s, w] = sum_v R[s, v] W[u, v, w] per instruction (tfc_fctp_rows_w: one short launch; a grouped GEMM with r-row
    tiles beyond the kernel's limits)."""
    if _rows_kernels_ok(spec):
        c_ins = _fctp_ins(tuple(spec.instructions))
        _lib.check(_lib.lib().tfc_fctp_rows_w(c_ins, len(c_ins), _ptr(R), r, spec.d_a, _ptr(weight), weight.numel(), _ptr(wp), _stream()))
        return
    _gemm_group([
        _desc(
            r, mu * mw, mv, _p(R, ao), _p(weight, wo), _p(wp, lo[7]),
            a1_r=_lin(spec.d_a), a1_k=_lin(1), b_k=_lin(mw), b_j=_imap(1, mw, mv * mw, 0, 1),
            c_r=_lin(mw), c_j=_imap(1, mw, r * mw, 0, 1), tag=b"gemm_fctp_lowrank_w",
        )
        for (xo, mu, ao, mv, oo, mw, d, wo, alpha), lo in zip(spec.instructions, low_ins)
    ])


def _lowrank_dw(spec, low_ins, R, r, dwp, dw):
    """dW[u, v, w] = sum_s R[s, v] dW'[u, s, w] (tfc_fctp_rows_dw)."""
    if _rows_kernels_ok(spec):
        c_ins = _fctp_ins(tuple(spec.instructions))
        _lib.check(_lib.lib().tfc_fctp_rows_dw(c_ins, len(c_ins), _ptr(R), r, spec.d_a, _ptr(dwp), dw.numel(), _ptr(dw), _stream()))
        return
    _gemm_group([
        _desc(
            mv, mu * mw, r, _p(R, ao), _p(dwp, lo[7]), _p(dw, wo),
            a1_r=_lin(1), a1_k=_lin(spec.d_a), b_k=_lin(mw), b_j=_imap(1, mw, r * mw, 0, 1),
            c_r=_lin(mw), c_j=_imap(1, mw, mv * mw, 0, 1), tag=b"gemm_fctp_lowrank_w",
        )
        for (xo, mu, ao, mv, oo, mw, d, wo, alpha), lo in zip(spec.instructions, low_ins)
    ])


def _lowrank_da(spec, low_ins, r, dwp, weight, part):
    """part[i, s, v] = sum_{u,w} dW'[u, s, w] W[u, v, w]: the gradient of the distinct rows, per instruction (a split-K GEMM:
    a dedicated kernel was measured no faster, csrc/tfc_layout.hip)."""
    _gemm_group([
        _desc(
            r, mv, mu * mw, _p(dwp, lo[7]), _p(weight, wo), _p(part, i * r * spec.d_a + ao),
            a1_r=_lin(mw), a1_k=_imap(1, mw, r * mw, 0, 1),
            b_k=_imap(1, mw, mv * mw, 0, 1), b_j=_lin(mw), c_r=_lin(spec.d_a), c_j=_lin(1),
            tag=b"gemm_fctp_da_rows",
        )
        for i, ((xo, mu, ao, mv, oo, mw, d, wo, alpha), lo) in enumerate(zip(spec.instructions, low_ins))
    ])


class _FctpFn(torch.autograd.Function):
    """FullyConnectedTensorProduct(x, a) with scalar second input (the self connection).

    Dense route: one contraction over (u, v) per instruction.  Low-rank route (`factors` =
    attr_factors(a) = (C, R), a = C R): the same product is FCTP(x, C; W') with
    W'[u,s,w] = sum_v R[s,v] W[u,v,w], whose contraction index (u, s) is d_a / r times shorter; the
    forward, dx and dW take it, da is per node and stays dense.  Both routes are exact
    re-associations of the same sums."""

    @staticmethod
    def forward(ctx, x, a, weight, spec: FctpSpec, base, factors):
        x, a, weight = _f32(x, "x"), _f32(a, "node_attrs"), _f32(weight, "weight")
        n = x.shape[0]
        if x.shape[1] != spec.d_x or a.shape[1] != spec.d_a or weight.numel() != spec.weight_numel:
            raise ValueError("fctp: tensor shapes do not match the irreps")
        if base is not None:
            if not base.is_contiguous() or base.shape != (n, spec.d_out):
                raise ValueError("fctp: `base` must be a contiguous [N, d_out] tensor")
            out = base
            ctx.mark_dirty(base)
            on_base = True
        else:
            alloc = torch.empty if spec.out_covered else torch.zeros
            out = alloc(n, spec.d_out, dtype=torch.float32, device=x.device)
            on_base = False
        ROUTE_COUNTS["fctp_dense" if factors is None else "fctp_low_rank"] += 1
        if factors is None:
            _fctp_fwd(n, x, a, spec.d_a, weight, out, spec, spec.instructions, on_base)
            ctx.save_for_backward(x, a, weight)
            ctx.low = None
        else:
            Cm, R = factors.C, factors.R
            r = Cm.shape[1]
            # W'[u, s, w] = sum_v R[s, v] W[u, v, w], per instruction
            low_ins, off = [], 0
            for xo, mu, ao, mv, oo, mw, d, wo, alpha in spec.instructions:
                low_ins.append((xo, mu, 0, r, oo, mw, d, off, alpha))
                off += mu * r * mw
            wp = torch.empty(max(off, 1), dtype=torch.float32, device=x.device)
            _lowrank_w(spec, low_ins, R, r, weight, wp)
            _fctp_fwd(n, x, Cm, r, wp, out, spec, low_ins, on_base)
            ctx.save_for_backward(x, a, weight, Cm, R, wp)
            ctx.low = tuple(low_ins)
            ctx.factors = factors
        ctx.spec = spec
        ctx.has_base = base is not None
        return out

    @staticmethod
    def backward(ctx, g):
        spec = ctx.spec
        if torch.is_grad_enabled():
            # create_graph=True: the dense, differentiable form of the backward (exactly the same sums; with
            # promised classes the attribute gradient is then the per-node one, whose class sums are equal)
            x, a, weight = ctx.saved_tensors[:3]
            dx, da, dw = _FctpBwdFn.apply(g, x, a, weight, spec, *ctx.needs_input_grad[:3])
            return dx, da, (dw.view_as(weight) if dw is not None else None), None, (g if ctx.has_base else None), None
        g = _f32(g, "grad_output")
        if ctx.low is None:
            x, a, weight = ctx.saved_tensors
        else:
            x, a, weight, Cm, R, wp = ctx.saved_tensors
        n = x.shape[0]
        dx = da = dw = None
        # channel-contiguous copies ([dim][N][mul] per block): the contractions below all run over
        # channels, which e3nn's [mul][dim] blocks hold at stride dim
        gt = _ir_transpose(g, spec.out_blocks or [(oo, mw, d) for _, _, _, _, oo, mw, d, _, _ in spec.instructions])
        low = ctx.low is not None
        project_da = low and ctx.factors.promised and ctx.needs_input_grad[1]
        if ctx.needs_input_grad[0]:
            dx = (torch.empty if spec.x_covered else torch.zeros)(n, spec.d_x, dtype=torch.float32, device=x.device)
        if ctx.needs_input_grad[1] and not project_da:
            da = (torch.empty if spec.a_covered else torch.zeros)(n, spec.d_a, dtype=torch.float32, device=x.device)
        if not low:
            if ctx.needs_input_grad[2]:
                dw = torch.empty_like(weight)
            if dx is not None or dw is not None or da is not None:
                _fctp_bwd(n, x, a, spec.d_a, weight, gt, spec, spec.instructions, dx=dx, dw=dw, da=da)
        else:
            r = Cm.shape[1]
            dwp = torch.empty_like(wp) if (ctx.needs_input_grad[2] or project_da) else None
            if dx is not None or dwp is not None:
                _fctp_bwd(n, x, Cm, r, wp, gt, spec, ctx.low, dx=dx, dw=dwp)
            if da is not None:  # classes found by inspection: the per-node attribute gradient, dense
                _fctp_bwd(n, x, a, spec.d_a, weight, gt, spec, spec.instructions, da=da)
            if ctx.needs_input_grad[2]:
                # dW[u, v, w] = sum_s R[s, v] dW'[u, s, w]
                dw = torch.empty_like(weight)
                _lowrank_dw(spec, ctx.low, R, r, dwp, dw)
            if project_da:
                # dR[s, v] = sum_{u,w} dW'[u, s, w] W[u, v, w]: the gradient of the distinct rows; every row
                # of a class then receives its class mean (sums back to dR through any lookup a = R[class])
                ni = len(spec.instructions)
                part = torch.empty(ni, r, spec.d_a, dtype=torch.float32, device=x.device)
                if not spec.a_covered or any(ins[3] != spec.d_a for ins in spec.instructions):
                    part.zero_()
                _lowrank_da(spec, ctx.low, r, dwp, weight, part)
                da = torch.empty(n, spec.d_a, dtype=torch.float32, device=x.device)
                _lib.check(_lib.lib().tfc_fctp_class_rows(_ptr(part), ni, r, spec.d_a, _ptr(ctx.factors.inv_count),
                                                          _ptr(ctx.factors.inv), n, _ptr(da), _stream()))
        return dx, da, dw, None, (g if ctx.has_base else None), None


class _FctpBwdFn(torch.autograd.Function):
    """(dx, da, dw) of the self connection as a differentiable op (dense route).  The product is trilinear in
    (x, a, W): every derivative of its transposes is the product or a transpose again."""

    @staticmethod
    def forward(ctx, g, x, a, weight, spec: FctpSpec, need_dx: bool, need_da: bool, need_dw: bool):
        g = _f32(g, "grad_output")
        x, a, weight = _f32(x, "x"), _f32(a, "node_attrs"), _f32(weight, "weight")
        n = x.shape[0]
        dev = x.device
        gt = _ir_transpose(g, spec.out_blocks or [(oo, mw, d) for _, _, _, _, oo, mw, d, _, _ in spec.instructions])
        dx = (torch.empty if spec.x_covered else torch.zeros)(n, spec.d_x, dtype=torch.float32, device=dev) if need_dx else None
        da = (torch.empty if spec.a_covered else torch.zeros)(n, spec.d_a, dtype=torch.float32, device=dev) if need_da else None
        dw = torch.empty(spec.weight_numel, dtype=torch.float32, device=dev) if need_dw else None
        if dx is not None or da is not None or dw is not None:
            _fctp_bwd(n, x, a, spec.d_a, weight.reshape(-1), gt, spec, spec.instructions, dx=dx, dw=dw, da=da)
        ctx.save_for_backward(g, x, a, weight)
        ctx.spec = spec
        ctx.set_materialize_grads(False)
        return dx, da, dw

    @staticmethod
    def backward(ctx, ggx, gga, ggw):
        g, x, a, weight = ctx.saved_tensors
        spec = ctx.spec
        fwd = lambda x_, a_, w_: _FctpFn.apply(x_, a_, w_.reshape(-1), spec, None, None)  # noqa: E731
        bwd = lambda x_, a_, w_, *need: _FctpBwdFn.apply(g, x_, a_, w_.reshape(-1), spec, *need)  # noqa: E731
        g_g = g_x = g_a = g_w = None
        if ggx is not None:
            g_g = _add(g_g, fwd(ggx, a, weight))
            _, da1, dw1 = bwd(ggx, a, weight, False, True, True)
            g_a, g_w = _add(g_a, da1), _add(g_w, dw1)
        if gga is not None:
            g_g = _add(g_g, fwd(x, gga, weight))
            dx2, _, dw2 = bwd(x, gga, weight, True, False, True)
            g_x, g_w = _add(g_x, dx2), _add(g_w, dw2)
        if ggw is not None:
            g_g = _add(g_g, fwd(x, a, ggw))
            dx3, da3, _ = bwd(x, a, ggw, True, True, False)
            g_x, g_a = _add(g_x, dx3), _add(g_a, da3)
        return g_g, g_x, g_a, (g_w.view_as(weight) if g_w is not None else None), None, None, None, None


def fctp(x, a, weight, spec: FctpSpec, base: Optional[torch.Tensor] = None, low_rank: bool = True,
         classes: Optional[torch.Tensor] = None):
    """FullyConnectedTensorProduct(x, a) with scalar `a`; when `base` is given the result is
    accumulated into it in place (out = base + fctp), saving the separate add.  `low_rank`: take the
    shorter contraction when the rows of `a` are copies of a few distinct rows (attr_factors)."""
    x, a, weight = _tracked(x), _tracked(a), _tracked(weight)
    factors = attr_factors(a, classes) if low_rank else None
    return _FctpFn.apply(x, a, weight, spec, base, factors)


# ------------------------------------------------------------------ Gate


@dataclass(frozen=True)
class GateSpec:
    d_in: int
    d_out: int
    segs: Tuple[Tuple[int, int, int, int, int, int, float], ...]  # in_off,out_off,count,dim,gate_off,act,cst
    # (off, mul, dim) blocks of the input / output row: when given, the kernels also write the
    # channel-contiguous copy of dx / out for the o3.Linear that reads it next (tfc_gate_*_t)
    in_blocks: Tuple[Tuple[int, int, int], ...] = ()
    out_blocks: Tuple[Tuple[int, int, int], ...] = ()

    def c_array(self):
        arr = (GateSeg * len(self.segs))()
        for s, v in zip(arr, self.segs):
            s.in_off, s.out_off, s.count, s.dim, s.gate_off, s.act, s.cst = v
        return arr


class _GateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, spec: GateSpec):
        x = _f32(x, "x")
        if x.shape[1] != spec.d_in:
            raise ValueError(f"gate: expected {spec.d_in} columns, got {x.shape[1]}")
        n = x.shape[0]
        out = torch.empty(n, spec.d_out, dtype=torch.float32, device=x.device)
        blocks = spec.out_blocks if CONV_TRANSPOSED and n and len(spec.out_blocks) <= 32 else ()
        out_t = torch.empty(n * spec.d_out, dtype=torch.float32, device=x.device) if blocks else None
        _lib.check(_lib.lib().tfc_gate_fwd_t(spec.c_array(), len(spec.segs), _ptr(x), n, spec.d_in, spec.d_out, _ptr(out),
                                             _ir_blocks(blocks), len(blocks), _ptr(out_t), _stream()))
        if out_t is not None:
            _transposed_put(out, blocks, out_t)
        ctx.save_for_backward(x)
        ctx.spec = spec
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        if torch.is_grad_enabled():  # create_graph=True
            return _GateBwdFn.apply(x, g, ctx.spec), None
        return _gate_bwd_raw(x, g, ctx.spec), None


def _gate_bwd_raw(x, g, spec: GateSpec):
    g = _f32(g, "grad_output")
    dx = torch.empty_like(x)
    n = x.shape[0]
    blocks = spec.in_blocks if CONV_TRANSPOSED and n and len(spec.in_blocks) <= 32 else ()
    dx_t = torch.empty(x.numel(), dtype=torch.float32, device=x.device) if blocks else None
    _lib.check(_lib.lib().tfc_gate_bwd_t(spec.c_array(), len(spec.segs), _ptr(x), _ptr(g), n, spec.d_in, spec.d_out, _ptr(dx),
                                         _ir_blocks(blocks), len(blocks), _ptr(dx_t), _stream()))
    if dx_t is not None:
        _transposed_put(dx, blocks, dx_t)
    return dx


class _GateBwdFn(torch.autograd.Function):
    """dx = J(x)^T g of the gate as a differentiable op; its backward is tfc_gate_bwd2 (second derivatives of the
    activations).  Third order is not built."""

    @staticmethod
    def forward(ctx, x, g, spec: GateSpec):
        x = _f32(x, "x")
        g = _f32(g, "grad_output")
        ctx.save_for_backward(x, g)
        ctx.spec = spec
        return _gate_bwd_raw(x, g, spec)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, v):
        x, g = ctx.saved_tensors
        spec = ctx.spec
        v = _f32(v, "cotangent")
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gg = torch.empty_like(g) if ctx.needs_input_grad[1] else None
        if gx is not None or gg is not None:
            _lib.check(_lib.lib().tfc_gate_bwd2(spec.c_array(), len(spec.segs), _ptr(x), _ptr(g), _ptr(v), x.shape[0], spec.d_in,
                                                spec.d_out, _ptr(gg), _ptr(gx), _stream()))
        return gx, gg, None


def gate(x, spec: GateSpec):
    return _GateFn.apply(_tracked(x), spec)


class _NormActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, spec: GateSpec, eps_squared: float, normalize: bool):
        x = _f32(x, "x")
        if x.shape[1] != spec.d_in:
            raise ValueError(f"norm_activation: expected {spec.d_in} columns, got {x.shape[1]}")
        out = torch.empty_like(x)
        _lib.check(_lib.lib().tfc_norm_act_fwd(spec.c_array(), len(spec.segs), _ptr(x), x.shape[0], spec.d_in, eps_squared,
                                               1 if normalize else 0, _ptr(out), _stream()))
        ctx.save_for_backward(x)
        ctx.args = (spec, eps_squared, normalize)
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        spec, eps_squared, normalize = ctx.args
        if torch.is_grad_enabled():  # create_graph=True: the backward as a differentiable op
            return _NormActBwdFn.apply(x, g, spec, eps_squared, normalize), None, None, None
        return _norm_act_bwd_raw(x, _f32(g, "grad_output"), spec, eps_squared, normalize), None, None, None


def _norm_act_bwd_raw(x, g, spec: "GateSpec", eps_squared: float, normalize: bool):
    dx = torch.empty_like(x)
    _lib.check(_lib.lib().tfc_norm_act_bwd(spec.c_array(), len(spec.segs), _ptr(x), _ptr(g), x.shape[0], spec.d_in, eps_squared,
                                           1 if normalize else 0, _ptr(dx), _stream()))
    return dx


class _NormActBwdFn(torch.autograd.Function):
    """dx = J(x)^T g of the norm activation as a differentiable op (second order; third order is not built).  J is
    symmetric, so the cotangent of g is the same kernel applied to the incoming cotangent; the cotangent of x is
    tfc_norm_act_bwd2."""

    @staticmethod
    def forward(ctx, x, g, spec: GateSpec, eps_squared: float, normalize: bool):
        x, g = _f32(x, "x"), _f32(g, "grad_output")
        ctx.save_for_backward(x, g)
        ctx.args = (spec, eps_squared, normalize)
        return _norm_act_bwd_raw(x, g, spec, eps_squared, normalize)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, v):
        x, g = ctx.saved_tensors
        spec, eps_squared, normalize = ctx.args
        v = _f32(v, "v")
        gx = gg = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            _lib.check(_lib.lib().tfc_norm_act_bwd2(spec.c_array(), len(spec.segs), _ptr(x), _ptr(g), _ptr(v), x.shape[0], spec.d_in,
                                                    eps_squared, 1 if normalize else 0, _ptr(gx), _stream()))
        if ctx.needs_input_grad[1]:
            gg = _norm_act_bwd_raw(x, v, spec, eps_squared, normalize)
        return gx, gg, None, None, None


def norm_activation(x, spec: GateSpec, eps_squared: float = 0.0, normalize: bool = True):
    """e3nn.nn.NormActivation on [N, d] rows (tfc_norm_act_fwd/bwd); `spec.segs`: one
    (in_off, in_off, mul, 2l+1, -1, act, 1.0) per block."""
    return _NormActFn.apply(_tracked(x), spec, float(eps_squared), bool(normalize))
