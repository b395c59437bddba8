"""Drop-in encoders with the constructor / forward surface the reference calls
(analysisgnn/models/analysis.py:444-473 constructors, :576-579 forward; legacy positional forms
models/chord.py:590,601 and models/pitch_spelling.py:157,221, models/cadence.py:232-234,298-299).

THE BUILD SPEC.  The real classes live in graphmuse (not in the reference tree, unpinned, not
installable here), so their exact layer wiring cannot be read offline; this file isolates every
choice that is "parity unpinned" (SURVEY.md App. A.6):
  * GNN stack = per layer `trim_to_layer` -> HeteroConv{SAGEConv per edge type}(aggr) and, between
    layers, LayerNorm -> ReLU -> dropout   [in-tree analog: models/cadence.py:142-176]
  * `aggr` across relations defaults to 'sum' [models/cadence.py:151,158]
  * HGT stack = PyG HGTConv(heads) layers, ReLU + dropout between layers
  * hybrid branch = padded 2-layer bi-GRU (hidden H/2) over each subgraph's target notes ->
    LayerNorm -> MLP, concatenated with the GNN output -> Linear(2H, H)
    [models/cadence.py:248-303, models/analysis.py:527-537]
  * use_jk = bi-LSTM JumpingKnowledge over the per-layer target-note outputs [core/gnn.py:345-365]
  * MetricalGNN = GNN stack + MLP(H -> H -> output_channels)
Parameter names follow PyG (`convs.<i>.convs.<src___rel___dst>.lin_l/lin_r`, `kqv_lin.lins.<type>` ...)
so a graphmuse/PyG state_dict can be mapped by name once it can be inspected.

All message passing runs on the C-ABI kernels; dense projections are library GEMMs; nothing
here has a CPU path.
"""
from __future__ import annotations

import os

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops
from .gru import gru_forward
from .fused import FusedSequential, norm_act
from .params import sage_operands, sage_operands_cat
from .linear import Linear, deferring, flush_deferred, linear, set_home_stream
from .core_layers import JumpingKnowledge
from .graph import HeteroIndex, hetero_index

EdgeType = Tuple[str, str, str]


def et_key(et: EdgeType) -> str:
    return "<" + "___".join(et) + ">"


# ------------------------------------------------------------------------------------------
# neighbour-mask conventions: per-hop counts, per-element hop index, or None
# ------------------------------------------------------------------------------------------
def _hop_counts(mask, keys, num_layers):
    """Per-hop count lists.  Hop-index tensors (0 = target) are converted with one bincount per key;
    all keys share one list length (deepest hop seen + 1) so that PyG's `[-layer]` indexing lines up."""
    out = {}
    tens = {k: mask[k] for k in keys if k in mask and isinstance(mask[k], torch.Tensor)}
    if tens:
        depth = 1 + max((int(m.max()) if m.numel() else 0) for m in tens.values())   # host sync, once per batch
    for k in keys:
        if k not in mask:
            continue
        m = mask[k]
        if isinstance(m, torch.Tensor):
            out[k] = torch.bincount(m, minlength=depth).tolist()
        else:
            out[k] = [int(v) for v in m]
    return out


class TrimPlan:
    """Rows / COO prefixes each layer keeps (PyG trim_to_layer applied cumulatively,
    models/cadence.py:165-173).  With no masks every layer keeps everything."""

    def __init__(self, num_layers, x_dict, edge_index_dict, mask_node, mask_edge):
        self.n_keep: List[Dict[str, int]] = []
        self.e_keep: List[Dict[EdgeType, Optional[int]]] = []
        n = {k: int(v.shape[0]) for k, v in x_dict.items()}
        e = {k: int(v.shape[1]) for k, v in edge_index_dict.items()}
        if mask_node is None or mask_edge is None:
            for _ in range(num_layers):
                self.n_keep.append(dict(n))
                self.e_keep.append({k: None for k in e})
            return
        nodes = _hop_counts(mask_node, n.keys(), num_layers)
        edges = _hop_counts(mask_edge, e.keys(), num_layers)
        for layer in range(num_layers):
            if layer > 0:
                for k in n:
                    if k in nodes:
                        n[k] -= nodes[k][-layer]
                for k in e:
                    if k in edges:
                        e[k] -= edges[k][-layer]
            self.n_keep.append(dict(n))
            self.e_keep.append(dict(e))


# ------------------------------------------------------------------------------------------
# SAGE
# ------------------------------------------------------------------------------------------
class SAGEConv(nn.Module):
    """Parameter holder with PyG SAGEConv defaults (mean aggr, root weight, lin_l bias only)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin_l = nn.Linear(in_channels, out_channels, bias=True)
        self.lin_r = nn.Linear(in_channels, out_channels, bias=False)


class HeteroConv(nn.Module):
    """HeteroConv({edge_type: SAGEConv}, aggr) fused per destination type: one multi-relation
    gather-reduce (mean per relation, written as the [N, R*H] A operand) and two GEMMs
    (neighbour blocks, pre-summed root blocks)."""

    def __init__(self, edge_types: Sequence[EdgeType], in_channels: int, out_channels: int, aggr: str = "sum"):
        super().__init__()
        if aggr not in ("sum", "mean"):
            raise NotImplementedError(f"aggr={aggr!r}")
        self.edge_types = [tuple(et) for et in edge_types]
        self.aggr = aggr
        self.in_channels, self.out_channels = in_channels, out_channels
        self.convs = nn.ModuleDict({et_key(et): SAGEConv(in_channels, out_channels) for et in self.edge_types})

    def forward(self, x_dict, edge_index_dict, index: Optional[HeteroIndex] = None,
                n_keep: Optional[Dict[str, int]] = None, e_keep: Optional[Dict[EdgeType, Optional[int]]] = None):
        if index is None:
            index = hetero_index(edge_index_dict, {k: int(v.shape[0]) for k, v in x_dict.items()})
        by_dst: Dict[str, List[EdgeType]] = {}
        for et in self.edge_types:
            s, _, d = et
            if et in index.fwd and s in x_dict and d in x_dict:
                by_dst.setdefault(d, []).append(et)
        out = {}
        for d, ets in by_dst.items():
            n = n_keep[d] if n_keep is not None else int(x_dict[d].shape[0])
            src_types: List[str] = []
            src_id = []
            for et in ets:
                if et[0] not in src_types:
                    src_types.append(et[0])
                src_id.append(src_types.index(et[0]))
            convs = [self.convs[et_key(et)] for et in ets]
            plist = ([c.lin_l.weight for c in convs], [c.lin_l.bias for c in convs], [c.lin_r.weight for c in convs])
            no_edges = e_keep is not None and all(e_keep[et] is not None and e_keep[et] <= 0 for et in ets)
            # One GEMM per layer where the kernel carries the root operand along (agnn_spmm_root_f32: widths 256 / 512):
            # y = [mean_1 .. mean_R | x_dst] @ [W_l_1 .. W_l_R | sum_r W_r_r]^T + sum_r b_r
            fused = (no_edges or self.in_channels in ops.ROOT_WIDTHS) and x_dict[d].is_cuda
            cat = sage_operands_cat(*plist, with_l=not no_edges) if fused else None
            if cat is not None and no_edges:
                y = linear(_head(x_dict[d], n), cat[0], cat[1])           # see below: lin_l contributes its bias only
                out[d] = y / len(ets) if self.aggr == "mean" else y
                continue
            if cat is not None:
                spec = ops.AggSpec(fwd=[index.fwd[et] for et in ets], bwd=[index.bwd[et] for et in ets], src_id=src_id,
                                   n_rows=n, mean=True, shared_slot=False, root=True,
                                   e_limit=[e_keep[et] for et in ets] if e_keep is not None else None)
                A = ops.aggregate(spec, [x_dict[s] for s in src_types], self_t=x_dict[d])   # [n, (R+1)*H]
                y = linear(A, cat[0], cat[1])
                out[d] = y / len(ets) if self.aggr == "mean" else y
                continue
            W_l, b, W_r = sage_operands(*plist)                                          # [out, R*H], [out], [out, H]
            if no_edges:
                # No relation keeps an edge at this layer: every mean is zero, lin_l contributes its bias only.  This IS the
                # last layer of the reference's default setup — num_layers convolutions over num_layers - 1 sampled hops
                # (train/train_analysisgnn.py:154), trim_to_layer dropping hop 2's edges at layer 1 and hop 1's at layer 2
                # — so the [n, R*H] zero matrix, its GEMM, that GEMM's two gradients and the aggregation launches are not
                # issued at all (lin_l.weight gets the zero gradient it has).
                y = linear(_head(x_dict[d], n), W_r, b)
                out[d] = y / len(ets) if self.aggr == "mean" else y
                continue
            spec = ops.AggSpec(fwd=[index.fwd[et] for et in ets], bwd=[index.bwd[et] for et in ets], src_id=src_id,
                               n_rows=n, mean=True, shared_slot=False,
                               e_limit=[e_keep[et] for et in ets] if e_keep is not None else None)
            A = ops.aggregate(spec, [x_dict[s] for s in src_types])                      # [n, R*H]
            y = linear(_head(x_dict[d], n), W_r, None, acc=linear(A, W_l, b))                   # second GEMM accumulates (beta = 1)
            out[d] = y / len(ets) if self.aggr == "mean" else y
        return out


class HeteroSAGEStack(nn.Module):
    def __init__(self, edge_types, input_channels, hidden_channels, num_layers, dropout=0.5, aggr="sum"):
        super().__init__()
        self.num_layers = num_layers
        self.dropout = dropout
        self.convs = nn.ModuleList()
        self.layer_norms = nn.ModuleList()
        for i in range(num_layers):
            self.convs.append(HeteroConv(edge_types, input_channels if i == 0 else hidden_channels, hidden_channels, aggr))
            if i < num_layers - 1:
                self.layer_norms.append(nn.LayerNorm(hidden_channels))

    def forward(self, x_dict, edge_index_dict, plan: TrimPlan, collect: Optional[list] = None):
        index = hetero_index(edge_index_dict, {k: int(v.shape[0]) for k, v in x_dict.items()})
        self.last_index = index          # reused by the onset pooling that follows the encoder (models.py)
        index.prepare_trim(plan.e_keep)  # sampled batch: the row ends of all trimmed layers in one launch
        # A layer followed by one that keeps NO edge (the last layer of the reference's default setup, see HeteroConv) is
        # only read at the rows that layer keeps: the rows PyG's trim_to_layer would still carry for it as message sources
        # are dead, so they are not computed — and the next layer's input is the whole matrix, not a slice of it (a slice's
        # backward is a zero fill, a copy and a gradient add).
        n_out = [dict(k) for k in plan.n_keep]
        for i in range(self.num_layers - 2, -1, -1):
            nxt = plan.e_keep[i + 1]
            if nxt and all(v is not None and v <= 0 for v in nxt.values()):
                n_out[i] = {t: min(n, n_out[i + 1].get(t, n)) for t, n in n_out[i].items()}
        for i, conv in enumerate(self.convs):
            keep = n_out[i]
            x_dict = {k: v for k, v in x_dict.items()}
            if i == 1:
                from .gru import wait_for_projections
                wait_for_projections(x_dict["note"].device)      # the sequence branch's inner input projection gets the chip to itself
            x_dict = conv(x_dict, edge_index_dict, index, keep, plan.e_keep[i])
            if i < self.num_layers - 1:
                x_dict = {k: norm_act(v, self.layer_norms[i], post_relu=True, p=self.dropout, training=self.training)
                          for k, v in x_dict.items()}
            if collect is not None:
                collect.append(x_dict["note"])
        return x_dict


# ------------------------------------------------------------------------------------------
# hybrid (sequence) branch — persistent GRU kernels (gru.py) for hidden 128, library RNN otherwise
# ------------------------------------------------------------------------------------------
_LENGTHS_CACHE: Dict[tuple, tuple] = {}


def _sequence_lengths(batch: torch.Tensor):
    """Per-subgraph sequence lengths as a device tensor and a host list.  `bincount(...).tolist()` (what the
    reference does, models/analysis.py:529-530) is a device->host sync that stalls the launch queue behind the
    whole previous step; the result only depends on the batch-id tensor, so it is memoised on that tensor's
    identity + version (a fresh batch tensor pays the sync once, like the reference)."""
    known = getattr(batch, "agnn_target_lengths", None)         # set by the batch assembly (synth.torch_inputs, batching.py):
    if known is not None:                                        # the window sizes are host knowledge there, no sync at all
        lens = [int(v) for v in known]
        key = ("known", tuple(lens), str(batch.device))
        hit = _LENGTHS_CACHE.get(key)
        if hit is None:
            if len(_LENGTHS_CACHE) >= 16:
                _LENGTHS_CACHE.pop(next(iter(_LENGTHS_CACHE)))
            hit = _LENGTHS_CACHE[key] = (None, lens, None)
        return hit[0], hit[1]
    key = (batch.data_ptr(), batch._version, batch.numel(), str(batch.device))
    hit = _LENGTHS_CACHE.get(key)
    if hit is None:
        lengths = torch.bincount(batch)
        hit = (lengths, lengths.tolist(), batch)          # keep `batch` alive so the data_ptr stays unique
        if len(_LENGTHS_CACHE) >= 16:
            _LENGTHS_CACHE.pop(next(iter(_LENGTHS_CACHE)))
        _LENGTHS_CACHE[key] = hit
    return hit[0], hit[1]


def _head(t: torch.Tensor, n: int) -> torch.Tensor:
    """t[:n] without an autograd node when it is the whole tensor (SliceBackward costs a zero-fill, a copy and, for a
    tensor with a second consumer, a gradient add per use)."""
    return t if n >= t.shape[0] else t[:n]


_SIDE_STREAMS: Dict[int, "torch.cuda.Stream"] = {}


def _side_stream(dev: torch.device) -> "torch.cuda.Stream":
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _SIDE_STREAMS:
        _SIDE_STREAMS[idx] = torch.cuda.Stream(device=dev, priority=SIDE_STREAM_PRIORITY)
    return _SIDE_STREAMS[idx]


SIDE_STREAM_PRIORITY = 0        # A/B switch (bench.py --side-priority): -1 = the sequence branch's stream ahead of the main one


LATE_SEQUENCE_BACKWARD = True   # the sequence branch behind a late-created node (_LateNode); bench.py --schedule measures both


class _Stamp(torch.autograd.Function):
    """Measurement aid (AGNN_STAMPS=1, _lib.stamp): identity whose forward / backward leave a device time stamp on the stream
    they run on — where a branch's forward ends and where its backward begins in the replayed hipGraph."""

    @staticmethod
    def forward(ctx, x, name):
        ctx.name = name
        _lib.stamp(name + " fwd", x.device)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if g is not None:
            _lib.stamp(ctx.name + " bwd", g.device)
        return g, None


def _stamped(x, name):
    return _Stamp.apply(x, name) if _lib.STAMPS["on"] and x.requires_grad else x


class _FlushPoint(torch.autograd.Function):
    """Identity on the GNN stack's input.  Its backward runs where the stack's backward ends — where the main stream starts
    to idle until the sequence branch's backward arrives — and runs the weight gradients that the backward pass has deferred
    on this stream so far (linear.defer_weight_grads) right there."""

    @staticmethod
    def forward(ctx, x, side=None):
        ctx.side = side
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        _lib.stamp("flush (main) start", g.device)
        flush_deferred(spill_to=ctx.side)
        _lib.stamp("flush (main) end", g.device)
        return g, None


class _ForkInput(torch.autograd.Function):
    """The encoder input handed to both branches: (x for the GNN stack, x[:n] for the sequence branch).  Its backward IS the
    join of the two backward passes: the sequence branch's input gradient is added onto the first n rows of the GNN stack's
    (one launch over n rows, on the main stream), instead of autograd's SliceBackward (a full-size zero fill and a copy, on
    the branch's stream — as late dependents of the recurrence's last GEMM they started ~0.18 ms after it in the replayed
    graph) plus a full-size gradient add.  The sequence branch's deferred weight gradients are issued here as well, on the
    branch's stream, BEFORE the add is captured (see backward)."""

    @staticmethod
    def forward(ctx, x, n, side):
        ctx.n, ctx.side = n, side
        ctx.set_materialize_grads(False)
        return x.view_as(x), x.narrow(0, 0, n)

    @staticmethod
    def backward(ctx, g_gnn, g_seq):
        g = g_gnn
        # The branch's deferred work is captured BEFORE the join's add: the add is then a later dependent of the branch's last
        # kernel and lands in the main stream's node list (with the tail behind it) instead of extending the branch's list —
        # and the replay starts feeding the main stream's list when the branch's list is nearly through (3.40 vs 4.00 ms).
        if ctx.side is not None:
            with torch.cuda.stream(ctx.side):
                if g is not None:
                    _lib.stamp("sequence branch dX done (side)", g.device)
                flush_deferred(g.device if g is not None else None)
                if g is not None:
                    _lib.stamp("flush (side) end", g.device)
        if g is not None:
            _lib.stamp("join (main)", g.device)
        if g_seq is not None:
            if g is None:
                raise _lib.AgnnError("_ForkInput: the GNN stack produced no input gradient")   # both branches read the input
            # Out of place: autograd does not promise that `g` has no other holder (a backward that returns its incoming
            # gradient unchanged hands the same tensor to two consumers).  n is (nearly) all rows, so this moves the same
            # bytes as the in-place add over n rows would.
            out = torch.empty_like(g)
            torch.add(g[:ctx.n], g_seq, out=out[:ctx.n])
            if ctx.n < g.shape[0]:
                torch.mul(g[ctx.n:], 1.0, out=out[ctx.n:])       # (an element-wise launch: the runtime's copy kernel took 11 us for 335 rows)
            g = out
        return g, None, None


class _Inner:
    """The sequence branch's own autograd graph: its output and the detached leaf it was built on."""

    def __init__(self, z, x_leaf):
        self.z, self.x_leaf = z, x_leaf


class _LateNode(torch.autograd.Function):
    """Stands for the sequence branch in the outer autograd graph (deferred weight gradients only).  The branch's kernels are
    issued FIRST in the forward pass, but the autograd engine runs ready nodes NEWEST first: a branch recorded first is
    differentiated — and captured — after the whole GNN stack, and a replayed hipGraph feeds its per-stream node lists to
    the GPU one after the other, the next list once the previous one is down to its last ~12 nodes
    (scripts/graph_window_probe.py): the recurrence's backward, the step's longest chain, then starts ~0.5 ms after its
    input gradient exists.  So the branch is recorded on a detached leaf and this node — created after the GNN stack, hence
    the newest — differentiates it with a nested backward pass: its chain is captured first, and with every piece of
    optimizer-only work deferred it is 9 kernels long, so the GNN stack's list is fed right behind it."""

    @staticmethod
    def forward(ctx, x_seq, inner: _Inner):
        ctx.inner = inner
        return inner.z.detach()

    @staticmethod
    def backward(ctx, dz):
        inner, ctx.inner = ctx.inner, None
        if inner is None:
            raise _lib.AgnnError("the sequence branch's private autograd graph was freed by the first backward pass: a second "
                                 "backward (retain_graph=True, torch.autograd.grad twice) is not supported while "
                                 "dp.defer_weight_grads is on with encoders.LATE_SEQUENCE_BACKWARD — switch either off")
        if dz is None:
            return None, None
        torch.autograd.backward([inner.z], [dz])
        g = inner.x_leaf.grad if inner.x_leaf.requires_grad else None
        inner.x_leaf.grad = None
        return g, None


class _HybridMixin:
    def _init_hybrid(self, input_channels, hidden_channels, num_layers, dropout, use_jk):
        self.use_jk = bool(use_jk)
        if self.use_jk:
            self.jk = JumpingKnowledge(hidden_channels, num_layers)
        self.rnn = nn.GRU(input_size=input_channels, hidden_size=hidden_channels // 2, num_layers=2,
                          batch_first=True, bidirectional=True, dropout=dropout)
        self.rnn_norm = nn.LayerNorm(hidden_channels)
        self.rnn_mlp = FusedSequential(Linear(hidden_channels, hidden_channels), nn.ReLU(),
                                     nn.LayerNorm(hidden_channels), nn.Dropout(dropout),
                                     Linear(hidden_channels, hidden_channels))
        self.cat_proj = Linear(hidden_channels * 2, hidden_channels)

    def hybrid_forward(self, x, batch):
        lengths, lens = _sequence_lengths(batch)
        if len(set(lens)) == 1:                     # equal windows (the usual batch): a view, no padding
            y = x.view(len(lens), lens[0], x.shape[1])
            y = gru_forward(self.rnn, y, self.training)
            y = self.rnn_mlp(norm_act(y, self.rnn_norm))
            return y.reshape(-1, y.shape[-1])
        seqs = nn.utils.rnn.pad_sequence(x.split(lens), batch_first=True, padding_value=0.0)
        y = gru_forward(self.rnn, seqs, self.training)
        y = self.rnn_mlp(norm_act(y, self.rnn_norm))
        return torch.cat(nn.utils.rnn.unpad_sequence(y, batch_first=True, lengths=torch.tensor(lens)), dim=0)

    # The sequence branch only needs the encoder INPUT, so it runs on a second HIP stream beside the
    # GNN stack (the persistent GRU kernels occupy 2*B of the 256 CUs and are latency bound); autograd
    # replays the backward on the same streams.  Set to False to serialise (debugging).
    overlap_sequence_branch = True

    def _start_branch(self, x_in, batch_dict, batch_size):
        dev = x_in.device
        if batch_dict is None:
            batch_note = torch.zeros(batch_size, dtype=torch.long, device=dev)
        else:
            full = batch_dict["note"]
            batch_note = full[:batch_size]
            known = getattr(full, "agnn_target_lengths", None)      # per-subgraph target counts known on the host
            if known is not None and sum(int(v) for v in known) == batch_size:
                batch_note.agnn_target_lengths = known
        if not (self.overlap_sequence_branch and x_in.is_cuda):
            return self.hybrid_forward(_head(x_in, batch_size), batch_note), None, None
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev)
        gnn_note = None
        x_seq = None
        if torch.is_grad_enabled() and x_in.requires_grad and deferring(x_in):
            # deferred weight gradients: the two branches hang off one node whose backward is their join (_ForkInput), and the
            # GNN stack's input passes through the main stream's flush point (_FlushPoint)
            x_gnn, x_seq = _ForkInput.apply(x_in, batch_size, side)
            gnn_note = _FlushPoint.apply(x_gnn, side)
        side.wait_stream(main)
        # x_in was allocated on the main stream and is READ on the side stream — in forward, and again in backward by the
        # first recurrent layer's weight gradient (its saved input is a view of x_in).  Without this the allocator hands the
        # block back to the main stream's pool the moment the last Python reference dies, while the side stream's kernels
        # that read it are only queued: a main-stream allocation then overwrites it under them (seen as a wrong
        # rnn.weight_ih_l0 gradient with FlatGradBuffer(views=True) + deferral, tests/test_gpu_step.py).
        x_in.record_stream(side)

        set_home_stream(main)
        if x_seq is not None and LATE_SEQUENCE_BACKWARD:
            with torch.cuda.stream(side):
                x_leaf = x_seq.detach().requires_grad_(True)
                inner = self.hybrid_forward(x_leaf, batch_note)
            if inner.requires_grad:
                return (x_seq, _Inner(inner, x_leaf)), side, gnn_note   # the node is created in `_finish`, after the GNN stack
            return inner, side, gnn_note
        with torch.cuda.stream(side):
            z = self.hybrid_forward(x_seq if x_seq is not None else _head(x_in, batch_size), batch_note)
        return z, side, gnn_note

    @staticmethod
    def _gnn_input(x_dict, gnn_note):
        """The GNN stack's input: the dict itself, or (deferred weight gradients) the note matrix behind the flush point.
        (`gnn_note` travels through the caller's frame, not through the module: forward stays re-entrant.)"""
        return x_dict if gnn_note is None else {**x_dict, "note": gnn_note}

    def _finish(self, x_note, outs, z, side, batch_size):
        x = _head(x_note, batch_size)
        if self.use_jk:
            x = self.jk([_head(o, batch_size) for o in outs])
        x = _stamped(x, "GNN stack")
        if isinstance(z, tuple):
            with torch.cuda.stream(side):                # the node belongs to the side stream, like the branch it stands for
                z = _stamped(_LateNode.apply(*z), "sequence branch")
        elif side is not None:
            with torch.cuda.stream(side):
                z = _stamped(z, "sequence branch")
        if side is not None:
            torch.cuda.current_stream(x.device).wait_stream(side)
            z.record_stream(torch.cuda.current_stream(x.device))
        return _stamped(self.cat_proj(torch.cat((x, z), dim=-1)), "cat_proj")


class HybridGNN(nn.Module, _HybridMixin):
    """analysis.py:455-462 constructor; forward keywords analysis.py:576-579; legacy positional
    order (x_dict, edge_index_dict, batch_dict, batch_size, masks...) chord.py:601."""

    def __init__(self, metadata=None, input_channels=None, hidden_channels=None, num_layers=2, dropout=0.5,
                 use_jk=False, aggr="sum", edge_types=None, **kwargs):
        super().__init__()
        if metadata is None:
            if edge_types is None:
                raise TypeError("HybridGNN needs metadata=(node_types, edge_types)")
            nodes = sorted({t for et in edge_types for t in (et[0], et[2])})
            metadata = (nodes, list(edge_types))          # older signature, models/chord.py:590
        self.metadata = (list(metadata[0]), [tuple(e) for e in metadata[1]])
        self.num_layers = num_layers
        self.gnn = HeteroSAGEStack(self.metadata[1], input_channels, hidden_channels, num_layers, dropout, aggr)
        self._init_hybrid(input_channels, hidden_channels, num_layers, dropout, use_jk)

    def forward(self, x_dict, edge_index_dict, batch_dict=None, batch_size=None, neighbor_mask_node=None,
                neighbor_mask_edge=None, return_edge_index=False, edge_attr_dict=None):
        _lib.require_gpu(*x_dict.values())
        if batch_size is None:
            batch_size = int(x_dict["note"].shape[0])
        plan = TrimPlan(self.num_layers, x_dict, edge_index_dict, neighbor_mask_node, neighbor_mask_edge)
        outs: list = []
        z, side, gnn_note = self._start_branch(x_dict["note"], batch_dict, batch_size)
        h = self.gnn(self._gnn_input(x_dict, gnn_note), edge_index_dict, plan, outs)
        out = self._finish(h["note"], outs, z, side, batch_size)
        return (out, edge_index_dict) if return_edge_index else out


class MetricalGNN(nn.Module):
    """analysis.py:464-473 (keywords), pitch_spelling.py:157 (positional), cadence.py:232-234."""

    def __init__(self, input_channels, hidden_channels, output_channels, num_layers, metadata, dropout=0.5,
                 use_jk=False, fast=False, aggr="sum", **kwargs):
        super().__init__()
        self.metadata = (list(metadata[0]), [tuple(e) for e in metadata[1]])
        self.num_layers = num_layers
        self.fast = fast               # graphmuse's grouped-GEMM switch; this path is always fused
        self.gnn = HeteroSAGEStack(self.metadata[1], input_channels, hidden_channels, num_layers, dropout, aggr)
        self.use_jk = bool(use_jk)
        if self.use_jk:
            self.jk = JumpingKnowledge(hidden_channels, num_layers)
        self.mlp = FusedSequential(Linear(hidden_channels, hidden_channels), nn.ReLU(),
                                 nn.LayerNorm(hidden_channels), nn.Dropout(dropout),
                                 Linear(hidden_channels, output_channels))

    def forward(self, x_dict, edge_index_dict, neighbor_mask_node=None, neighbor_mask_edge=None, batch_dict=None,
                batch_size=None, return_edge_index=False, edge_attr_dict=None):
        _lib.require_gpu(*x_dict.values())
        plan = TrimPlan(self.num_layers, x_dict, edge_index_dict, neighbor_mask_node, neighbor_mask_edge)
        outs: list = []
        h = self.gnn(x_dict, edge_index_dict, plan, outs)["note"]
        if batch_size is not None:
            h = _head(h, batch_size)
            outs = [_head(o, batch_size) for o in outs]
        if self.use_jk:
            h = self.jk(outs)
        out = self.mlp(h)
        return (out, edge_index_dict) if return_edge_index else out
