"""The tensor-field convolution message-passing layer (eigenn/nn/point_conv.py), same
classes, constructor arguments, irreps bookkeeping and parameter names, computed by the
libtfc kernels.

Fusion boundary.  In the reference, `PointConvMessage` materialises EDGE_MESSAGE [E, D_mid]
and `PointConvUpdate` scatter-adds it.  Here, when the update function is the stock
`PointConvUpdate` (always, in the two target models), the message function runs the fused
kernel (sender gather + 'uvu' product + per-receiver sum + 1/sqrt(avg_num_neighbors)) and
hands the [N, D_mid] node sums to the update function; the per-edge tensor is never
written.  Nothing downstream reads EDGE_MESSAGE, but its irreps are still
published in `irreps_out` so chained irreps dicts equal the reference's.  With
`fuse_aggregation=False` (or a custom update class) the per-edge tensor is produced and
summed afterwards, as in the reference.
"""
from __future__ import annotations

from typing import Dict, Optional

import threading
import weakref

import torch

from .. import keys as DataKey
from .. import ops
from ..irreps import Irreps
from ..module_irreps import ModuleIrreps
from .e3 import FullyConnectedTensorProduct, Linear
from .utils import ActivationLayer, UVUTensorProduct, scatter_add

__all__ = ["PointConvMessage", "PointConvUpdate", "PointConvMessagePassing", "PointConv", "compile_mode"]

_CLASSES = DataKey.NODE_ATTRS_CLASSES
_NODE_MESSAGE = "_tfc_node_message"  # private hand-off: [N, D_mid] sums, already scaled


def compile_mode(mode: str):
    """Stand-in for e3nn.util.jit.compile_mode (a marker decorator, eigenn/nn/point_conv.py:197)."""

    def deco(cls):
        cls._e3nn_compile_mode = mode
        return cls

    return deco


# Prefetch state of the side-stream radial networks, kept outside the modules (they stay picklable
# and deep-copyable) and per host thread (two threads driving two models never see each other's
# state): which layer ran right after which IN THE SAME PASS (learnt from the previous pass), and
# the weights a layer's predecessor has already computed for it.
#
# A pass is the run of layers that see one edge-embedding tensor; a layer that shows up a second
# time starts a new pass, so the last layer of a model is never linked to the first one (the weights
# it would prefetch belong to a step whose parameters are not known yet).  Prefetched weights are
# keyed on the embedding tensor (identity, address, version), the grad mode AND the versions of the
# successor's radial-network parameters, so an optimiser step or load_state_dict between two passes
# on the same embedding tensor can never be answered with weights of the old parameters.
class _RadialChain(threading.local):
    def __init__(self):
        self.pass_key = None
        self.pass_layers = []  # weakrefs of the layers of the current pass, in order
        self.next_of = weakref.WeakKeyDictionary()  # layer -> weakref of its successor
        self.prefetched = weakref.WeakKeyDictionary()  # layer -> (key, embedding, weights, event)
        self.hub = None  # (embedding, its alias recorded under the side stream)


_CHAIN = _RadialChain()


def _radial_key(layer, emb):
    """Everything the radial network's output depends on besides the values of `emb`."""
    return (id(emb), emb.data_ptr(), emb._version, torch.is_grad_enabled(),
            tuple((p.data_ptr(), p._version) for p in layer.tp.weight_nn.parameters()))


def _embedding_hub(emb, side):
    """An alias of the edge embedding whose autograd node lives on the side stream, shared by the radial
    networks of a pass.  Every layer's network sends a gradient to the embedding; fed the tensor itself, autograd
    sums those on the embedding's own (main) stream as they arrive, which parks the main stream behind each radial
    network's backward in turn.  Through the alias the partial gradients meet on the side stream, and the main
    stream waits once, for the finished sum, when the edge encoders' backward finally needs it."""
    if not (ops.MLP_EMBEDDING_HUB and emb.requires_grad and torch.is_grad_enabled()):
        return emb
    hit = _CHAIN.hub
    if hit is not None and hit[0] is emb and hit[2] == emb._version:
        return hit[1]
    side.wait_stream(torch.cuda.current_stream(emb.device))
    with torch.cuda.stream(side):
        hub = emb.view(emb.shape)
    _CHAIN.hub = (emb, hub, emb._version)
    return hub


def _note_radial_chain(layer, emb_key):
    st = _CHAIN
    seen = [r() for r in st.pass_layers]
    if st.pass_key != emb_key or any(l is layer for l in seen):
        st.pass_key, st.pass_layers = emb_key, []  # a new pass: no link from the previous pass's last layer
    elif seen and seen[-1] is not None:
        st.next_of[seen[-1]] = weakref.ref(layer)
    st.pass_layers.append(weakref.ref(layer))


def reset_prefetch_state():
    """Forget the learnt layer order and drop prefetched weights (tests; also safe at any time)."""
    _CHAIN.__init__()


class PointConvMessage(ModuleIrreps, torch.nn.Module):
    """eigenn/nn/point_conv.py:11-87"""

    REQUIRED_KEYS_IRREPS_IN = [DataKey.NODE_FEATURES, DataKey.EDGE_EMBEDDING, DataKey.EDGE_ATTRS, DataKey.EDGE_INDEX]
    REQUIRED_TYPE_IRREPS_IN = {DataKey.EDGE_EMBEDDING: "0e"}

    def __init__(self, irreps_in: Dict[str, Irreps], irreps_out: Dict[str, Irreps], *, fc_num_hidden_layers: int = 1, fc_hidden_size: int = 8):
        super().__init__()
        self.init_irreps(irreps_in, irreps_out)
        node_in = self.irreps_in[DataKey.NODE_FEATURES]
        node_out = self.irreps_out[DataKey.NODE_FEATURES]
        self.linear_1 = Linear(node_in, node_in)
        self.tp = UVUTensorProduct(
            node_in,
            self.irreps_in[DataKey.EDGE_ATTRS],
            node_out,
            mlp_input_size=self.irreps_in[DataKey.EDGE_EMBEDDING].dim,
            mlp_hidden_size=fc_hidden_size,
            mlp_num_hidden_layers=fc_num_hidden_layers,
        )
        self.irreps_out[DataKey.EDGE_MESSAGE] = self.tp.irreps_out
        # set by PointConvMessagePassing when the update function can take node sums
        self.fuse_aggregation = False
        self.aggregation_scale = 1.0

    def _radial_on_side(self, emb, side, pre=None):
        """This layer's radial network under the side stream (so that its backward runs there too).  With
        `pre` (activations a predecessor computed ahead of time, _prefetch_next) nothing is launched: the
        call only records the autograd node, HERE in the forward order - had the prefetching layer recorded
        it, autograd would run its backward after that earlier layer's convolution backward, at the very end
        of the pass with nothing left to overlap it with."""
        src = _embedding_hub(emb, side)
        with torch.cuda.stream(side):
            w = self.tp.weight_nn(src, pre=pre)
            ready = None
            if pre is None:
                ready = torch.cuda.Event()
                ready.record(side)
        emb.record_stream(side)
        return w, ready

    def _precompute_on_side(self, emb, side):
        with torch.cuda.stream(side):
            pre = self.tp.weight_nn.precompute(emb)
            ready = torch.cuda.Event()
            ready.record(side)
        emb.record_stream(side)
        return pre, ready

    def _prefetch_next(self, emb, side):
        # (ops.MLP_PREFETCH_DEPTH layers ahead; one without gradients, where the weights of a layer
        # can be freed before the next ones are computed)
        cur = self
        done = [r() for r in _CHAIN.pass_layers]  # never prefetch for a layer that already ran in this pass
        for _ in range(ops.MLP_PREFETCH_DEPTH if torch.is_grad_enabled() else 1):
            nxt = _CHAIN.next_of.get(cur)
            nxt = nxt() if nxt is not None else None
            if (not ops.MLP_PREFETCH or nxt is None or any(nxt is l for l in done) or not nxt.fuse_aggregation
                    or nxt.tp.weight_nn is None):
                break
            key = _radial_key(nxt, emb)
            have = _CHAIN.prefetched.get(nxt)
            if have is None or have[0] != key or have[1] is not emb:
                if ops.MLP_PREFETCH_RECORD_EARLY:  # the autograd node recorded by the prefetching layer (round-1 order)
                    _CHAIN.prefetched[nxt] = (key, emb) + nxt._radial_on_side(emb, side) + (True,)
                else:
                    _CHAIN.prefetched[nxt] = (key, emb) + nxt._precompute_on_side(emb, side) + (False,)
            cur = nxt

    def forward(self, data: DataKey.Type) -> DataKey.Type:
        emb = data[DataKey.EDGE_EMBEDDING]
        w = ready = None
        if self.fuse_aggregation and ops.MLP_SIDE_STREAM and self.tp.weight_nn is not None and emb.is_cuda:
            # The radial network runs on the side stream beside linear_1 (and, in backward, beside the
            # next convolution backward).  It depends on the edge embedding only, so the layer that
            # follows this one on the same embedding tensor (learnt from the previous pass) has its
            # weights computed right behind ours, while this layer's convolution runs.
            main, side = torch.cuda.current_stream(emb.device), ops.side_stream(emb.device)
            key = _radial_key(self, emb)
            pre = _CHAIN.prefetched.pop(self, None)
            if pre is not None and pre[0] == key and pre[1] is emb:
                ready = pre[3]
                w = pre[2] if pre[4] else self._radial_on_side(emb, side, pre=pre[2])[0]
            else:
                side.wait_stream(main)
                w, ready = self._radial_on_side(emb, side)
            _note_radial_chain(self, key[:4])
            if not ops.MLP_PREFETCH_AFTER_CONV:
                self._prefetch_next(emb, side)
        x = self.linear_1(data[DataKey.NODE_FEATURES])
        if w is not None:
            main.wait_event(ready)
            w.record_stream(main)
        ordering = ops.edge_ordering(data[DataKey.EDGE_INDEX], x.shape[0])
        if self.fuse_aggregation:
            data[_NODE_MESSAGE] = self.tp(
                x, data[DataKey.EDGE_ATTRS], emb, ordering=ordering, scale=self.aggregation_scale, weight=w
            )
            data.pop(DataKey.EDGE_MESSAGE, None)
            if w is not None and ops.MLP_PREFETCH_AFTER_CONV:
                # behind this layer's convolution: beside the small products that follow it
                side.wait_stream(main)
                self._prefetch_next(emb, side)
        else:
            w = self.tp.weight_nn(data[DataKey.EDGE_EMBEDDING]) if self.tp.weight_nn is not None else None
            if w is None:
                w = self.tp.weight.unsqueeze(0).expand(ordering.n_edges, -1).contiguous()
            data[DataKey.EDGE_MESSAGE] = self.tp.tp(x, data[DataKey.EDGE_ATTRS], w, ordering, 1.0, per_edge=True)
        return data


class PointConvUpdate(ModuleIrreps, torch.nn.Module):
    """eigenn/nn/point_conv.py:90-194"""

    REQUIRED_KEYS_IRREPS_IN = [DataKey.NODE_FEATURES, DataKey.NODE_ATTRS, DataKey.EDGE_MESSAGE, DataKey.EDGE_INDEX]

    def __init__(self, irreps_in: Dict[str, Irreps], irreps_out: Dict[str, Irreps], *, use_self_connection: bool = True, avg_num_neighbors: Optional[float] = None):
        super().__init__()
        self.avg_num_neighbors = avg_num_neighbors
        self.init_irreps(irreps_in, irreps_out)
        node_in = self.irreps_in[DataKey.NODE_FEATURES]
        node_out = self.irreps_out[DataKey.NODE_FEATURES]
        self.linear_2 = Linear(self.irreps_in[DataKey.EDGE_MESSAGE], node_out)
        if use_self_connection:
            self.self_connection = FullyConnectedTensorProduct(node_in, self.irreps_in[DataKey.NODE_ATTRS], node_out)
        else:
            self.self_connection = None

    @property
    def aggregation_scale(self) -> float:
        return 1.0 if self.avg_num_neighbors is None else float(self.avg_num_neighbors) ** -0.5

    def forward(self, data: DataKey.Type) -> DataKey.Type:
        node_feats = data[DataKey.NODE_FEATURES]
        if _NODE_MESSAGE in data:  # summed and scaled inside the convolution kernel
            msg = data.pop(_NODE_MESSAGE)
        else:
            edge_src, edge_dst = data[DataKey.EDGE_INDEX]
            msg = scatter_add(data[DataKey.EDGE_MESSAGE], edge_dst, dim_size=len(node_feats))
            if self.avg_num_neighbors is not None:
                msg = msg * self.aggregation_scale
        msg = self.linear_2(msg)
        if self.self_connection is not None:
            # out = msg + self_connection(x, attrs), accumulated into msg's buffer by the GEMM
            node_feats = self.self_connection(node_feats, data[DataKey.NODE_ATTRS], base=msg, classes=data.get(_CLASSES))
        else:
            node_feats = msg
        data[DataKey.NODE_FEATURES] = node_feats
        return data


@compile_mode("script")
class PointConvMessagePassing(ModuleIrreps, torch.nn.Module):
    """eigenn/nn/point_conv.py:197-347: message -> update -> Gate -> optional residual."""

    def __init__(
        self,
        irreps_in: Dict[str, Irreps],
        conv_layer_irreps: Irreps,
        message=PointConvMessage,
        update=PointConvUpdate,
        message_kwargs: Optional[Dict] = None,
        update_kwargs: Optional[Dict] = None,
        activation_type: str = "gate",
        activation_scalars: Dict[str, str] = None,
        activation_gates: Dict[str, str] = None,
        use_resnet: bool = True,
    ):
        super().__init__()
        message_kwargs = {} if message_kwargs is None else dict(message_kwargs)
        update_kwargs = {} if update_kwargs is None else dict(update_kwargs)
        self.init_irreps(irreps_in)
        edge_attrs_irreps = self.irreps_in[DataKey.EDGE_ATTRS]
        node_in = self.irreps_in[DataKey.NODE_FEATURES]

        self.equivariant_nonlinear = ActivationLayer(
            tp_irreps_in1=node_in,
            tp_irreps_in2=edge_attrs_irreps,
            tp_irreps_out=conv_layer_irreps,
            activation_type=activation_type,
            activation_scalars=activation_scalars,
            activation_gates=activation_gates,
        )
        irreps_before_act = self.equivariant_nonlinear.irreps_in
        irreps_after_act = self.equivariant_nonlinear.irreps_out

        message_kwargs.pop("irreps_in", None)
        message_kwargs.pop("irreps_out", None)
        self.message_fn = message(irreps_in=self.irreps_in, irreps_out={DataKey.NODE_FEATURES: irreps_before_act}, **message_kwargs)

        update_fn_irreps_in = dict(self.irreps_in)
        update_fn_irreps_in[DataKey.EDGE_MESSAGE] = self.message_fn.irreps_out[DataKey.EDGE_MESSAGE]
        update_kwargs.pop("irreps_in", None)
        update_kwargs.pop("irreps_out", None)
        self.update_fn = update(irreps_in=update_fn_irreps_in, irreps_out={DataKey.NODE_FEATURES: irreps_before_act}, **update_kwargs)

        if isinstance(self.message_fn, PointConvMessage) and isinstance(self.update_fn, PointConvUpdate):
            self.message_fn.fuse_aggregation = True
            self.message_fn.aggregation_scale = self.update_fn.aggregation_scale

        self.use_resnet = bool(use_resnet and irreps_after_act == node_in)
        self.irreps_out.update(self.update_fn.irreps_out)
        self.irreps_out[DataKey.NODE_FEATURES] = irreps_after_act

    def forward(self, data: DataKey.Type) -> DataKey.Type:
        old_x = data[DataKey.NODE_FEATURES]
        data = self.message_fn(data)
        data = self.update_fn(data)
        x = self.equivariant_nonlinear(data[DataKey.NODE_FEATURES])
        data[DataKey.NODE_FEATURES] = old_x + x if self.use_resnet else x
        return data


class PointConv(ModuleIrreps, torch.nn.Module):
    """Legacy single-module form, eigenn/nn/point_conv.py:351-491 (explicit irreps_out, no
    activation).  Constructs for any irreps (as the reference's tests/nn/test_convolution.py
    does); calling it needs irreps the kernels cover."""

    REQUIRED_KEYS_IRREPS_IN = [
        DataKey.NODE_FEATURES,
        DataKey.NODE_ATTRS,
        DataKey.EDGE_EMBEDDING,
        DataKey.EDGE_ATTRS,
        DataKey.EDGE_INDEX,
    ]
    REQUIRED_TYPE_IRREPS_IN = {DataKey.EDGE_EMBEDDING: "0e"}

    def __init__(
        self,
        irreps_in: Dict[str, Irreps],
        irreps_out: Dict[str, Irreps],
        *,
        fc_num_hidden_layers: int = 1,
        fc_hidden_size: int = 8,
        use_self_connection: bool = True,
        avg_num_neighbors: Optional[float] = None,
    ):
        super().__init__()
        self.avg_num_neighbors = avg_num_neighbors
        self.init_irreps(irreps_in, irreps_out)
        node_in = self.irreps_in[DataKey.NODE_FEATURES]
        node_out = self.irreps_out[DataKey.NODE_FEATURES]
        self.linear_1 = Linear(node_in, node_in)
        self.tp = UVUTensorProduct(
            node_in,
            self.irreps_in[DataKey.EDGE_ATTRS],
            node_out,
            mlp_input_size=self.irreps_in[DataKey.EDGE_EMBEDDING].dim,
            mlp_hidden_size=fc_hidden_size,
            mlp_num_hidden_layers=fc_num_hidden_layers,
        )
        self.linear_2 = Linear(self.tp.irreps_out, node_out)
        if use_self_connection:
            self.self_connection = FullyConnectedTensorProduct(node_in, self.irreps_in[DataKey.NODE_ATTRS], node_out)
        else:
            self.self_connection = None

    def forward(self, data: DataKey.Type) -> DataKey.Type:
        data = data.copy()
        x_in = data[DataKey.NODE_FEATURES]
        x = self.linear_1(x_in)
        ordering = ops.edge_ordering(data[DataKey.EDGE_INDEX], x.shape[0])
        scale = 1.0 if self.avg_num_neighbors is None else float(self.avg_num_neighbors) ** -0.5
        msg = self.tp(x, data[DataKey.EDGE_ATTRS], data[DataKey.EDGE_EMBEDDING], ordering=ordering, scale=scale)
        msg = self.linear_2(msg)
        if self.self_connection is not None:
            msg = self.self_connection(x_in, data[DataKey.NODE_ATTRS], base=msg, classes=data.get(_CLASSES))
        data[DataKey.NODE_FEATURES] = msg
        return data
