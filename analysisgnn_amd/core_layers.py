"""HIP-backed mirrors of the reference's in-tree message-passing layers.

Same class names, constructor arguments, parameter names (`state_dict` compatible) and forward
signatures as analysisgnn/models/core/gnn.py and core/hgnn.py, so the reference's callers
(models/chord.py:506-583 `MetricalChordEncoder`, models/cadence.py, core/hgnn.py stacks) can
switch imports.  The arithmetic runs on the C-ABI kernels (`ops.aggregate`): one CSR build per
graph, one multi-relation gather-reduce per layer, the R per-relation projections fused into
two GEMMs.  In these layers messages flow from `edge_index[1]` to `edge_index[0]`
(core/gnn.py:70,74).  There is no CPU path.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops
from .graph import Csr, SegSpec, build_csr

EdgeInput = Union[torch.Tensor, Dict[str, torch.Tensor]]


class TypedIndex:
    """CSR (rows = edge_index[0]) and its transpose for every relation code of an in-tree
    homogeneous-with-types graph: `edge_index [2,E]`, `edge_type [E]`, `etypes {name: code}`
    (core/hgnn.py:128-140).  Replaces the R boolean masks + compactions per layer."""

    def __init__(self, n_nodes: int, edge_index: EdgeInput, edge_type: Optional[torch.Tensor],
                 etypes: Dict[str, int]):
        self.names: List[str] = list(etypes.keys())
        self.n_nodes = int(n_nodes)
        specs: List[SegSpec] = []
        R = len(self.names)
        if isinstance(edge_index, dict):
            eis = [edge_index[k] for k in self.names]
            for ei in eis:
                specs.append(SegSpec(row=ei[0], col=ei[1], n_rows=self.n_nodes))
            for ei in eis:
                specs.append(SegSpec(row=ei[1], col=ei[0], n_rows=self.n_nodes))
            self.n_edges = [int(ei.shape[1]) for ei in eis]
        else:
            if edge_type is None:
                raise ValueError("Edge type must be specified")        # core/hgnn.py:134-135
            for k in self.names:
                specs.append(SegSpec(row=edge_index[0], col=edge_index[1], n_rows=self.n_nodes,
                                     etype=edge_type, code=int(etypes[k])))
            for k in self.names:
                specs.append(SegSpec(row=edge_index[1], col=edge_index[0], n_rows=self.n_nodes,
                                     etype=edge_type, code=int(etypes[k])))
            self.n_edges = [int(edge_index.shape[1])] * R
        csrs = build_csr(specs)
        self.fwd: List[Csr] = csrs[:R]
        self.bwd: List[Csr] = csrs[R:]
        # device-side "relation r has no edge" flags, no host sync (empty branch, core/gnn.py:67-69)
        self.empty = torch.stack([c.rowptr[-1] == c.rowptr[0] for c in self.fwd])   # bool [R]


_TYPED_CACHE: Dict[tuple, TypedIndex] = {}


def typed_index(n_nodes: int, edge_index: EdgeInput, edge_type, etypes: Dict[str, int]) -> TypedIndex:
    if isinstance(edge_index, dict):
        key = ("d", n_nodes) + tuple((k, v.data_ptr(), tuple(v.shape), v._version) for k, v in edge_index.items())
    else:
        key = ("t", n_nodes, edge_index.data_ptr(), tuple(edge_index.shape), edge_index._version,
               edge_type.data_ptr() if edge_type is not None else 0,
               edge_type._version if edge_type is not None else 0, tuple(etypes.items()))
    hit = _TYPED_CACHE.get(key)
    if hit is None:
        hit = TypedIndex(n_nodes, edge_index, edge_type, etypes)
        hit._keepalive = (edge_index, edge_type)
        if len(_TYPED_CACHE) >= 8:
            _TYPED_CACHE.pop(next(iter(_TYPED_CACHE)))
        _TYPED_CACHE[key] = hit
    return hit


def _xavier_relu_(lin: nn.Linear) -> None:
    nn.init.xavier_uniform_(lin.weight, gain=nn.init.calculate_gain("relu"))
    if lin.bias is not None:
        nn.init.constant_(lin.bias, 0.0)


class SageConvScatter(nn.Module):
    """core/gnn.py:39-76.  z = W[x || s] + b,  s_i = (x_i + sum_{(i,j)} (W_n x_j + b_n) [+ W_e e_ij]) / max(deg_i, 1);
    with no edges s = W_n x + b_n (the reference's empty branch, :67-69)."""

    def __init__(self, in_features, out_features, bias=True, in_edge_features=None):
        super().__init__()
        self.neigh_linear = nn.Linear(in_features, in_features, bias=bias)
        self.linear = nn.Linear(in_features * 2, out_features, bias=bias)
        self.in_edge_features = in_edge_features
        if in_edge_features is not None:
            self.edge_linear = nn.Linear(in_edge_features, in_features, bias=bias)
        self.reset_parameters()

    def reset_parameters(self):
        _xavier_relu_(self.linear)
        _xavier_relu_(self.neigh_linear)
        if self.in_edge_features is not None:
            _xavier_relu_(self.edge_linear)

    def forward(self, features, edge_index, edge_features=None, neigh_feats=None):
        _lib.require_gpu(features)
        src_feats = features if neigh_feats is None else neigh_feats
        h = self.neigh_linear(src_feats)
        if edge_index is None or edge_index.shape[1] == 0:
            return self.linear(torch.cat([features, h], dim=-1))
        n = features.shape[0]
        fwd, bwd = build_csr([SegSpec(edge_index[0], edge_index[1], n),
                              SegSpec(edge_index[1], edge_index[0], h.shape[0])])
        hp, F0 = ops.pad4(h)
        xp, _ = ops.pad4(features)
        srcs = [hp]
        spec = ops.AggSpec(fwd=[fwd], bwd=[bwd], src_id=[0], n_rows=n, mean=True, shared_slot=True)
        if self.in_edge_features is not None and edge_features is not None:
            # per-edge term W_e e_ij joins the numerator: a second "relation" whose gathered rows
            # are the E edge vectors themselves (col = edge id = the CSR's perm array)
            ef, _ = ops.pad4(self.edge_linear(edge_features))
            e_fwd = Csr(rowptr=fwd.rowptr, col=fwd.perm, perm=fwd.perm, n_rows=fwd.n_rows, n_edges=fwd.n_edges)
            E = edge_index.shape[1]
            ident = torch.arange(E + 1, dtype=torch.int32, device=features.device)
            e_bwd = Csr(rowptr=ident, col=edge_index[0].to(torch.int32), perm=ident, n_rows=E, n_edges=E)
            # mean over (node + edge) messages shares ONE count: aggregate sums, divide once
            tot = ops.aggregate(ops.AggSpec(fwd=[fwd, e_fwd], bwd=[bwd, e_bwd], src_id=[0, 1], n_rows=n,
                                            mean=False, shared_slot=True), [hp, ef])
            deg = (fwd.rowptr[1:] - fwd.rowptr[:-1]).clamp(min=1).to(torch.float32).unsqueeze(-1)
            s = (xp + tot) / deg
        else:
            s = ops.aggregate(spec, srcs, self_t=xp)
        s = s[:, :F0] if s.shape[1] != F0 else s
        return self.linear(torch.cat([features, s], dim=-1))


class HeteroAttention(nn.Module):
    """core/hgnn.py:8-23 (the `lstm` reduction of the hetero wrappers, :41, :115, :462): on the relation stack x [R, N, H]
    the batch_first bi-LSTM takes the R relations as the batch and the N NODES as the sequence, the softmax runs over the
    last axis of [R, N] (the nodes), and the result is the weighted sum over relations — reproduced literally.  The LSTM
    is the library's (MIOpen); hidden (R*H)//2 per direction."""

    def __init__(self, n_hidden, n_layers):
        super().__init__()
        self.lstm = nn.LSTM(n_hidden, (n_layers * n_hidden) // 2, bidirectional=True, batch_first=True)
        self.att = nn.Linear(2 * ((n_layers * n_hidden) // 2), 1)
        self.reset_parameters()

    def reset_parameters(self):
        self.lstm.reset_parameters()
        nn.init.xavier_uniform_(self.att.weight, gain=nn.init.calculate_gain("relu"))

    def forward(self, x):
        alpha, _ = self.lstm(x)
        alpha = torch.softmax(self.att(alpha).squeeze(-1), dim=-1)
        return (x * alpha.unsqueeze(-1)).sum(dim=0)


def _make_reduction(reduction: str, out_features: Optional[int] = None, n_rel: Optional[int] = None, allow_none: bool = False):
    """The reductions of core/hgnn.py:30-46 / :102-116 / :451-464 that run in the reference: 'mean', 'sum', 'lstm'
    (HeteroAttention, a module: returned as such so that its parameters sit under `reduction.*` as in the reference) and,
    for the ResGated wrapper only, 'none'.  'max' / 'min' hand a (values, indices) pair to `.to()` and 'concat' calls
    torch.cat on a tensor: both raise in the reference, so they raise here as well."""
    if reduction in ("mean", "sum"):
        return reduction
    if reduction == "lstm":
        return HeteroAttention(out_features, n_rel)
    if reduction == "none" and allow_none:
        return reduction
    raise NotImplementedError(f"reduction={reduction!r}: not runnable in the reference either (core/hgnn.py:108-112)")


def _reduce_stack(reduction, st: torch.Tensor) -> torch.Tensor:
    if isinstance(reduction, nn.Module):
        return reduction(st)
    if reduction == "mean":
        return st.mean(dim=0)
    if reduction == "sum":
        return st.sum(dim=0)
    return st                                            # 'none'


class HeteroSageConvLayer(nn.Module):
    """core/hgnn.py:98-140.  R SageConvScatter modules, one per relation, outputs reduced over
    the R slots.  Fused here: ONE GEMM for all W_n^r, ONE multi-relation gather-reduce, ONE GEMM
    over [x || s_1 .. s_R] with the root blocks of W^r pre-summed."""

    def __init__(self, in_features, out_features, etypes, bias=True, reduction="mean"):
        super().__init__()
        self.out_features = out_features
        self.in_features = in_features
        self.etypes = etypes
        self.reduction = _make_reduction(reduction, out_features, len(etypes))
        self.conv = nn.ModuleDict({k: SageConvScatter(in_features, out_features, bias=bias) for k in etypes.keys()})

    def reset_parameters(self):
        for c in self.conv.values():
            c.reset_parameters()

    def _per_relation(self, x, tix: TypedIndex):
        """[R, N, out]: every relation's own output (needed by the reductions that are not linear in the slots).  Same
        fused aggregation; the output projection is one batched GEMM over the relations."""
        names = tix.names
        R = len(names)
        Fin = self.in_features
        if Fin % 4 != 0:
            raise _lib.AgnnError("HeteroSageConvLayer on HIP needs in_features % 4 == 0")
        convs = [self.conv[k] for k in names]
        Wn = torch.cat([c.neigh_linear.weight for c in convs], dim=0)
        bn = torch.cat([c.neigh_linear.bias for c in convs]) if convs[0].neigh_linear.bias is not None else None
        Hcat = F.linear(x, Wn, bn)
        srcs = [Hcat[:, r * Fin:(r + 1) * Fin] for r in range(R)]
        spec = ops.AggSpec(fwd=tix.fwd, bwd=tix.bwd, src_id=list(range(R)), n_rows=x.shape[0], mean=True, shared_slot=False)
        S = ops.aggregate(spec, srcs, self_t=x)
        S = torch.where(tix.empty.repeat_interleave(Fin).unsqueeze(0), Hcat, S)          # empty relation: s_r = h_r
        W = torch.stack([c.linear.weight for c in convs])                                 # [R, out, 2F]
        xs = torch.cat([x.unsqueeze(0).expand(R, -1, -1), S.view(-1, R, Fin).transpose(0, 1)], dim=-1)   # [R, N, 2F]
        out = torch.bmm(xs, W.transpose(1, 2))
        if convs[0].linear.bias is not None:
            out = out + torch.stack([c.linear.bias for c in convs]).unsqueeze(1)
        return out

    def _fused(self, x, tix: TypedIndex):
        if not isinstance(self.reduction, str) or self.reduction == "none":
            return _reduce_stack(self.reduction, HeteroSageConvLayer._per_relation(self, x, tix))
        names = tix.names
        R = len(names)
        Fin = self.in_features
        if Fin % 4 != 0:
            raise _lib.AgnnError("HeteroSageConvLayer on HIP needs in_features % 4 == 0")
        convs = [self.conv[k] for k in names]
        Wn = torch.cat([c.neigh_linear.weight for c in convs], dim=0)                     # [R*F, F]
        bn = torch.cat([c.neigh_linear.bias for c in convs]) if convs[0].neigh_linear.bias is not None else None
        Hcat = F.linear(x, Wn, bn)                                                       # [N, R*F]
        srcs = [Hcat[:, r * Fin:(r + 1) * Fin] for r in range(R)]
        spec = ops.AggSpec(fwd=tix.fwd, bwd=tix.bwd, src_id=list(range(R)), n_rows=x.shape[0], mean=True,
                           shared_slot=False)
        S = ops.aggregate(spec, srcs, self_t=x)                                           # [N, R*F]
        # empty relations take s_r = h_r (core/gnn.py:67-69); flags live on the device
        S = torch.where(tix.empty.repeat_interleave(Fin).unsqueeze(0), Hcat, S)
        W_root = sum(c.linear.weight[:, :Fin] for c in convs)                             # [out, F]
        W_nb = torch.cat([c.linear.weight[:, Fin:] for c in convs], dim=1)                # [out, R*F]
        out = F.linear(torch.cat([x, S], dim=-1), torch.cat([W_root, W_nb], dim=1))
        if convs[0].linear.bias is not None:
            out = out + sum(c.linear.bias for c in convs)
        return out / R if self.reduction == "mean" else out

    def forward(self, x, edge_index, edge_type=None):
        _lib.require_gpu(x)
        if edge_type is None and not isinstance(edge_index, dict):
            raise ValueError("Edge type must be specified")
        return self._fused(x, typed_index(x.shape[0], edge_index, edge_type, self.etypes))


class JumpingKnowledge(nn.Module):
    """core/gnn.py:345-365: bi-LSTM attention over the per-layer outputs (MIOpen LSTM)."""

    def __init__(self, n_hidden, n_layers):
        super().__init__()
        self.lstm = nn.LSTM(n_hidden, (n_layers * n_hidden) // 2, bidirectional=True, batch_first=True)
        self.att = nn.Linear(2 * ((n_layers * n_hidden) // 2), 1)
        self.reset_parameters()

    def reset_parameters(self):
        self.lstm.reset_parameters()
        nn.init.xavier_uniform_(self.att.weight, gain=nn.init.calculate_gain("relu"))

    def forward(self, xs):
        x = torch.stack(xs, dim=1)
        alpha, _ = self.lstm(x)
        alpha = torch.softmax(self.att(alpha).squeeze(-1), dim=-1)
        return (x * alpha.unsqueeze(-1)).sum(dim=1)


_DEFAULT_ETYPES = {"onset": 0, "consecutive": 1, "during": 2, "rests": 3, "consecutive_rev": 4,
                   "during_rev": 5, "rests_rev": 6}


class HGCN(nn.Module):
    """core/hgnn.py:144-179: n_layers+1 hetero SAGE layers; hidden ones followed by
    relu -> L2 normalise -> dropout; optional JumpingKnowledge before the last layer."""

    def __init__(self, in_feats, n_hidden, out_feats, n_layers, etypes=None, activation=F.relu, dropout=0.5,
                 jk=False):
        super().__init__()
        etypes = dict(_DEFAULT_ETYPES) if etypes is None else etypes
        self.n_hidden = n_hidden
        self.layers = nn.ModuleList()
        self.normalize = F.normalize
        self.activation = activation
        self.dropout = nn.Dropout(dropout)
        self.layers.append(HeteroSageConvLayer(in_feats, n_hidden, etypes=etypes))
        for _ in range(n_layers - 1):
            self.layers.append(HeteroSageConvLayer(n_hidden, n_hidden, etypes=etypes))
        self.use_knowledge = bool(jk)
        if jk:
            self.jk = JumpingKnowledge(n_hidden=n_hidden, n_layers=n_layers)
        self.layers.append(HeteroSageConvLayer(n_hidden, out_feats, etypes=etypes))

    def reset_parameters(self):
        for conv in self.layers:
            conv.reset_parameters()

    def forward(self, x, edge_index, edge_type):
        h, hs = x, []
        for conv in self.layers[:-1]:
            h = self.dropout(self.normalize(self.activation(conv(h, edge_index, edge_type))))
            hs.append(h)
        if self.use_knowledge:
            h = self.jk(hs)
        return self.layers[-1](h, edge_index, edge_type)


# ------------------------------------------------------------------------------------------
# ResGatedGraphConv (core/gnn.py:212-258) on the edge-gated aggregation kernels
# ------------------------------------------------------------------------------------------
class _GatedAggregate(torch.autograd.Function):
    """S_i = sum_{(i,j)} sigmoid(a_i + b_j [+ c_e]) * h_j   (agnn_gated_*; c in COO edge order)."""

    @staticmethod
    def forward(ctx, fwd: Csr, bwd: Csr, a, b, h, c):
        dev = _lib.require_gpu(a, b, h, c)
        lib = _lib.load()
        a, b, h = a.contiguous(), b.contiguous(), h.contiguous()
        c = c.contiguous() if c is not None else None
        n, H = a.shape
        out = torch.empty_like(a)
        g = _lib.Gated(fwd.rowptr.data_ptr(), fwd.col.data_ptr(), fwd.perm.data_ptr(), a.data_ptr(), b.data_ptr(),
                       h.data_ptr(), _lib.ptr(c), a.stride(0), c.stride(0) if c is not None else 0, n, H)
        _lib.check(lib.agnn_gated_fwd_f32(g, out.data_ptr(), out.stride(0), _lib.stream_ptr(dev)), "agnn_gated_fwd_f32")
        ctx.csr = (fwd, bwd)
        ctx.has_c = c is not None
        ctx.save_for_backward(a, b, h, *([c] if c is not None else []))
        return out

    @staticmethod
    def backward(ctx, ds):
        fwd, bwd = ctx.csr
        a, b, h, *rest = ctx.saved_tensors
        c = rest[0] if ctx.has_c else None
        dev = ds.device
        lib = _lib.load()
        ds = ds.contiguous()
        n, H = a.shape
        da = torch.empty_like(a)
        db = torch.empty_like(b)
        dh = torch.empty_like(h)
        dc = torch.zeros_like(c) if c is not None else None
        ldc = c.stride(0) if c is not None else 0
        g1 = _lib.Gated(fwd.rowptr.data_ptr(), fwd.col.data_ptr(), fwd.perm.data_ptr(), a.data_ptr(), b.data_ptr(),
                        h.data_ptr(), _lib.ptr(c), a.stride(0), ldc, n, H)
        _lib.check(lib.agnn_gated_bwd_dst_f32(g1, ds.data_ptr(), ds.stride(0), da.data_ptr(), _lib.ptr(dc),
                                              _lib.stream_ptr(dev)), "agnn_gated_bwd_dst_f32")
        g2 = _lib.Gated(bwd.rowptr.data_ptr(), bwd.col.data_ptr(), bwd.perm.data_ptr(), a.data_ptr(), b.data_ptr(),
                        h.data_ptr(), _lib.ptr(c), a.stride(0), ldc, b.shape[0], H)
        _lib.check(lib.agnn_gated_bwd_src_f32(g2, ds.data_ptr(), ds.stride(0), db.data_ptr(), dh.data_ptr(),
                                              _lib.stream_ptr(dev)), "agnn_gated_bwd_src_f32")
        return None, None, da, db, dh, dc


class ResGatedGraphConv(nn.Module):
    """core/gnn.py:212-258.  out = 2 * W1 x + sum_j sigmoid(W3 x_i + W4 x_j [+ W5 e_ij]) * W2 x_j
    (the reference scatters into `out=h1.clone()` and then adds h1 again, :256-257)."""

    def __init__(self, in_features, out_features, bias=True, in_edge_features=None):
        super().__init__()
        self.W1 = nn.Linear(in_features, out_features, bias=bias)
        self.W2 = nn.Linear(in_features, out_features, bias=bias)
        self.W3 = nn.Linear(in_features, out_features, bias=bias)
        self.W4 = nn.Linear(in_features, out_features, bias=bias)
        self.in_edge_features = in_edge_features
        if in_edge_features is not None:
            self.W5 = nn.Linear(in_edge_features, out_features, bias=bias)
        self.out_features = out_features
        self.reset_parameters()

    def reset_parameters(self):
        for lin in (self.W1, self.W2, self.W3, self.W4):
            _xavier_relu_(lin)
        if self.in_edge_features is not None:
            _xavier_relu_(self.W5)

    def _projections(self, features, neigh_feats=None):
        nf = features if neigh_feats is None else neigh_feats
        W = torch.cat([self.W1.weight, self.W3.weight, self.W4.weight] + ([self.W2.weight] if neigh_feats is None else []), 0)
        bias = None
        if self.W1.bias is not None:
            bias = torch.cat([self.W1.bias, self.W3.bias, self.W4.bias] + ([self.W2.bias] if neigh_feats is None else []), 0)
        P = F.linear(features, W, bias)                     # one GEMM for W1, W3, W4 (and W2)
        O = self.out_features
        h1, a, b = P[:, :O], P[:, O:2 * O], P[:, 2 * O:3 * O]
        h2 = P[:, 3 * O:4 * O] if neigh_feats is None else self.W2(nf)
        return h1, a, b, h2

    def forward_csr(self, features, fwd: Csr, bwd: Csr, c=None, neigh_feats=None):
        h1, a, b, h2 = self._projections(features, neigh_feats)
        (a4, O), (b4, _), (h4, _) = ops.pad4(a), ops.pad4(b), ops.pad4(h2)
        c4 = ops.pad4(c)[0] if c is not None else None
        s = _GatedAggregate.apply(fwd, bwd, a4, b4, h4, c4)
        return 2.0 * h1 + (s[:, :O] if s.shape[1] != O else s)

    def forward(self, features, edge_index, edge_features=None, neigh_feats=None):
        _lib.require_gpu(features)
        n = features.shape[0]
        fwd, bwd = build_csr([SegSpec(edge_index[0], edge_index[1], n), SegSpec(edge_index[1], edge_index[0], n)])
        c = None
        if edge_features is not None and self.in_edge_features is not None:
            c = self.W5(edge_features)
        return self.forward_csr(features, fwd, bwd, c, neigh_feats)


class _HeteroPerRelation(nn.Module):
    """Shared forward of the in-tree hetero wrappers: one conv per relation slot, reduced over the R slots
    (core/hgnn.py:58-63, :479-484).  One CSR build for all relations; SAGE convs take the fused path."""

    def _run(self, x, edge_index, edge_type, edge_features=None):
        _lib.require_gpu(x)
        tix = typed_index(x.shape[0], edge_index, edge_type, self.etypes)
        convs = [self.conv[k] for k in tix.names]
        if all(isinstance(c, SageConvScatter) and c.in_edge_features is None for c in convs) and edge_features is None:
            return HeteroSageConvLayer._fused(self, x, tix)
        outs = []
        for r, conv in enumerate(convs):
            if isinstance(conv, ResGatedGraphConv):
                c = None
                if edge_features is not None and conv.in_edge_features is not None:
                    c = conv.W5(edge_features)           # all edges; the kernel reads the rows of this relation's edges
                outs.append(conv.forward_csr(x, tix.fwd[r], tix.bwd[r], c))
            elif isinstance(conv, (RelEdgeConv, SageConvScatter)):
                # per-edge messages of data-dependent size: the relation's edges are compacted as the reference does
                # (boolean mask, core/hgnn.py:481-483 — a host sync; this block is off the benchmarked configurations)
                if isinstance(edge_index, dict):
                    sub, ef = edge_index[tix.names[r]], None
                else:
                    m = edge_type == self.etypes[tix.names[r]]
                    sub = edge_index[:, m]
                    ef = edge_features[m] if edge_features is not None else None
                outs.append(conv(x, sub, ef))
            else:
                raise NotImplementedError(f"{type(conv).__name__} inside a hetero wrapper is not on the HIP path")
        return _reduce_stack(self.reduction, torch.stack(outs, dim=0))


class HeteroResGatedGraphConvLayer(_HeteroPerRelation):
    """core/hgnn.py:26-63."""

    def __init__(self, in_features, out_features, etypes, bias=True, reduction="mean"):
        super().__init__()
        self.out_features, self.in_features, self.etypes = out_features, in_features, etypes
        self.reduction = _make_reduction(reduction, out_features, len(etypes), allow_none=True)
        self.conv = nn.ModuleDict({k: ResGatedGraphConv(in_features, out_features, bias=bias) for k in etypes.keys()})

    def reset_parameters(self):
        for c in self.conv.values():
            c.reset_parameters()

    def forward(self, x, edge_index, edge_type):
        return self._run(x, edge_index, edge_type)


class HeteroConv(_HeteroPerRelation):
    """core/hgnn.py:435-484: wrap a conv class into a per-relation module with a reduction over relations."""

    def __init__(self, in_features, out_features, etypes, in_edge_features=None, module=SageConvScatter, bias=True,
                 reduction="mean"):
        super().__init__()
        self.out_features, self.in_features, self.etypes = out_features, in_features, etypes
        self.reduction = _make_reduction(reduction, out_features, len(etypes))
        self.conv = nn.ModuleDict({k: module(in_features, out_features, bias=bias, in_edge_features=in_edge_features)
                                   for k in etypes.keys()})

    def reset_parameters(self):
        for c in self.conv.values():
            c.reset_parameters()

    def forward(self, x, edge_index, edge_type, edge_features=None):
        return self._run(x, edge_index, edge_type, edge_features)


class _RelEdgeAggregate(torch.autograd.Function):
    """(S, D): S_i = sum_{(i,j)} h_j,  D_i = sum_{(i,j)} |h_i - h_j|, side by side in ONE [N, 2F] matrix (agnn_absdiff_*).
    `want_d=False`: S only ([N, F]) — the caller brings its own per-edge features."""

    @staticmethod
    def forward(ctx, fwd: Csr, bwd: Csr, h, want_d: bool):
        dev = _lib.require_gpu(h)
        lib = _lib.load()
        h = h.contiguous()
        n, F_ = h.shape
        out = torch.empty((n, 2 * F_ if want_d else F_), dtype=torch.float32, device=dev)
        _lib.check(lib.agnn_absdiff_fwd_f32(fwd.rowptr.data_ptr(), fwd.col.data_ptr(), h.data_ptr(), h.stride(0), n, F_,
                                            out.data_ptr(), out.data_ptr() + 4 * F_ if want_d else None, out.stride(0),
                                            _lib.stream_ptr(dev)), "agnn_absdiff_fwd_f32")
        ctx.csr, ctx.want_d = (fwd, bwd), want_d
        ctx.save_for_backward(h)
        return out

    @staticmethod
    def backward(ctx, g):
        fwd, bwd = ctx.csr
        (h,) = ctx.saved_tensors
        dev = g.device
        lib = _lib.load()
        g = g if (g.stride(1) == 1 and g.stride(0) % 4 == 0 and g.data_ptr() % 16 == 0) else g.contiguous()
        n, F_ = h.shape
        dh = torch.empty_like(h)
        gs, gd = g.data_ptr(), (g.data_ptr() + 4 * F_ if ctx.want_d else None)
        st = _lib.stream_ptr(dev)
        if ctx.want_d:       # the aggregating end: gD_i * sum_j sign(h_i - h_j)
            _lib.check(lib.agnn_absdiff_bwd_f32(fwd.rowptr.data_ptr(), fwd.col.data_ptr(), h.data_ptr(), h.stride(0), n, F_, gd, None,
                                                g.stride(0), 0, 0, dh.data_ptr(), dh.stride(0), st), "agnn_absdiff_bwd_f32")
        # the gathered end (transposed CSR): sum_i [ sign(h_j - h_i) * gD_i + gS_i ]
        _lib.check(lib.agnn_absdiff_bwd_f32(bwd.rowptr.data_ptr(), bwd.col.data_ptr(), h.data_ptr(), h.stride(0), n, F_, gd, gs,
                                            g.stride(0), 1, 1 if ctx.want_d else 0, dh.data_ptr(), dh.stride(0), st), "agnn_absdiff_bwd_f32")
        return None, None, dh, None


class RelEdgeConv(nn.Module):
    """core/gnn.py:79-106.  h = W_n x + b;  e_ij = |h_i - h_j| unless edge features are given;  m_ij = W_e [h_j || e_ij] + b_e;
    s_i = (h_i + sum_{(i,j)} m_ij) / max(deg_i, 1)  (scatter onto edge row 0 with `out=h.clone()`, mean);  z = W [x || s] + b.
    Re-derived, not transcribed: W_e is linear, so the row sum moves inside it —
        sum_j m_ij = W_e [ sum_j h_j || sum_j e_ij ] + deg_i b_e
    — two per-row aggregates from one pass over the neighbours (csrc/reledge.hip) and a NODE-level GEMM [N, 2F] x [2F, F];
    the reference's [E, F] gathers, the [E, 2F] cat and the edge-level GEMM do not exist here.  Given per-edge features take
    the same route (their row sums come from the gather-reduce kernel, `scatter.scatter`)."""

    def __init__(self, in_node_features, out_features, bias=True, in_edge_features=None):
        super().__init__()
        self.neigh_linear = nn.Linear(in_node_features, in_node_features, bias=bias)
        self.edge_linear = nn.Linear(in_node_features * 2 if in_edge_features is None else in_node_features + in_edge_features,
                                     in_node_features, bias=bias)
        self.linear = nn.Linear(in_node_features * 2, out_features, bias=bias)
        self.in_edge_features = in_edge_features
        self.reset_parameters()

    def reset_parameters(self):
        for lin in (self.linear, self.neigh_linear, self.edge_linear):
            _xavier_relu_(lin)

    def forward(self, features, edge_index, edge_features=None):
        _lib.require_gpu(features)
        n = features.shape[0]
        h = self.neigh_linear(features)
        Fh = h.shape[1]
        if Fh % 4 or Fh > 1024:
            raise _lib.AgnnError(f"RelEdgeConv: feature width {Fh} (the aggregation kernel takes multiples of 4 up to 1024)")
        fwd, bwd = build_csr([SegSpec(edge_index[0], edge_index[1], n), SegSpec(edge_index[1], edge_index[0], n)])
        deg = (fwd.rowptr[1:n + 1] - fwd.rowptr[:n]).to(torch.float32).unsqueeze(1)
        if edge_features is None:
            agg = _RelEdgeAggregate.apply(fwd, bwd, h, True)                               # [N, 2F] = [S | D]
        else:
            from .scatter import scatter
            e_sum = scatter(edge_features, edge_index[0], 0, dim_size=n, reduce="sum")     # [N, F_e]
            agg = torch.cat((_RelEdgeAggregate.apply(fwd, bwd, h, False), e_sum), dim=-1)
        msg = F.linear(agg, self.edge_linear.weight)
        if self.edge_linear.bias is not None:
            msg = torch.addcmul(msg, deg, self.edge_linear.bias.unsqueeze(0))              # + deg_i * b_e
        s = (h + msg) / deg.clamp(min=1.0)
        return self.linear(torch.cat([features, s], dim=-1))


class HeteroRelEdgeConvLayer(_HeteroPerRelation):
    """core/hgnn.py:66-95: one RelEdgeConv per relation, mean over the relations.  Node-level edge features [N, F_e] become
    |f_i - f_j| per edge (:82-83), per-edge ones [E, F_e] are used as they are (:84-85), anything else is ignored (:86-87)."""

    def __init__(self, in_features, out_features, etypes, bias=True, in_edge_features=None):
        super().__init__()
        self.out_features, self.in_features, self.etypes = out_features, in_features, etypes
        self.reduction = "mean"
        self.conv = nn.ModuleDict({k: RelEdgeConv(in_features, out_features, bias=bias, in_edge_features=in_edge_features)
                                   for k in etypes.keys()})

    def reset_parameters(self):
        for c in self.conv.values():
            c.reset_parameters()

    def forward(self, x, edge_index, edge_type, edge_features=None):
        if edge_features is not None and edge_features.shape[0] == x.shape[0]:
            edge_features = torch.abs(edge_features[edge_index[0]] - edge_features[edge_index[1]])
        elif edge_features is not None and edge_features.shape[0] == edge_index.shape[1]:
            pass
        else:
            edge_features = None
        return self._run(x, edge_index, edge_type, edge_features)


class GATConvLayer(nn.Module):
    """core/gnn.py:154-209.  The reference's softmax runs over the HEAD axis (its own remark at :205) and is
    then averaged over heads, so every edge weight is 1/heads up to rounding; reproduced literally: per-edge
    weights from the per-node scores, then a weighted gather-sum on the SpMM kernel."""

    def __init__(self, in_features, out_features, num_heads=3, bias=True, dropout=0.3, negative_slope=0.2,
                 in_edge_features=None):
        super().__init__()
        if in_edge_features is not None:
            raise NotImplementedError("GATConvLayer(in_edge_features=...) is broken in the reference itself (attne never initialised)")
        self.num_heads, self.in_features, self.out_features = num_heads, in_features, out_features
        self.linear = nn.Linear(in_features, out_features, bias=bias)
        self.el = nn.Linear(in_features, in_features * num_heads, bias=bias)
        self.er = nn.Linear(in_features, in_features * num_heads, bias=bias)
        self.attnl = nn.Parameter(torch.empty(1, num_heads, in_features))
        self.attnr = nn.Parameter(torch.empty(1, num_heads, in_features))
        self.negative_slope = negative_slope
        self.attndrop = nn.Dropout(dropout)
        self.reset_parameters()

    def reset_parameters(self):
        gain = nn.init.calculate_gain("relu")
        for w in (self.linear.weight, self.el.weight, self.er.weight, self.attnl, self.attnr):
            nn.init.xavier_normal_(w, gain=gain)
        for lin in (self.linear, self.el, self.er):
            if lin.bias is not None:
                nn.init.constant_(lin.bias, 0.0)

    def forward(self, features, edge_index, edge_features=None):
        _lib.require_gpu(features)
        n = features.shape[0]
        s_l = (self.el(features).view(n, self.num_heads, self.in_features) * self.attnl).sum(-1)      # [N, heads]
        s_r = (self.er(features).view(n, self.num_heads, self.in_features) * self.attnr).sum(-1)
        e = F.leaky_relu(s_l[edge_index[0]] + s_r[edge_index[1]], self.negative_slope)
        a = torch.softmax(self.attndrop(e), dim=1).mean(dim=1)                                          # [E]
        h = self.linear(features)
        hp, O = ops.pad4(h)
        fwd, bwd = build_csr([SegSpec(edge_index[0], edge_index[1], n), SegSpec(edge_index[1], edge_index[0], n)])
        spec = ops.AggSpec(fwd=[fwd], bwd=[bwd], src_id=[0], n_rows=n, mean=False, shared_slot=True,
                           edge_weight=[a.detach()])
        s = ops.aggregate(spec, [hp])
        # d(out)/d(a) vanishes analytically (softmax over heads sums to one); keep `a` in the graph at zero cost
        return h + (s[:, :O] if s.shape[1] != O else s) + 0.0 * a.sum()


# ------------------------------------------------------------------------------------------
# MetricalConvLayer (core/gnn.py:488-540) and the in-tree MetricalGNN (core/hgnn.py:323-433)
# ------------------------------------------------------------------------------------------
def _scatter_rows(src: torch.Tensor, dst_index: torch.Tensor, src_index: torch.Tensor, n_out: int) -> torch.Tensor:
    """out[i] = sum_{e: dst_index[e] == i} src[src_index[e]]  (zero-initialised scatter_add of gathered rows)."""
    fwd, bwd = build_csr([SegSpec(dst_index, src_index, n_out), SegSpec(src_index, dst_index, src.shape[0])])
    sp, O = ops.pad4(src)
    spec = ops.AggSpec(fwd=[fwd], bwd=[bwd], src_id=[0], n_rows=n_out, mean=False, shared_slot=True)
    out = ops.aggregate(spec, [sp])
    return out[:, :O] if out.shape[1] != O else out


class MetricalConvLayer(nn.Module):
    def __init__(self, in_dim, out_dim, activation=None, dropout=0.2, bias=True):
        super().__init__()
        self.input_dim, self.output_dim = in_dim, out_dim
        self.activation = nn.Identity() if activation is None else activation
        self.dropout = nn.Dropout(dropout)
        self.normalize = nn.BatchNorm1d(out_dim)
        self.neigh = nn.Linear(in_dim, in_dim, bias=bias)
        self.conv_out = nn.Linear(4 * in_dim, out_dim, bias=bias)
        self.seq = nn.GRU(in_dim, in_dim, batch_first=True, bias=bias, bidirectional=True)

    def reset_parameters(self):
        self.neigh.reset_parameters()
        self.conv_out.reset_parameters()
        self.seq.reset_parameters()

    def forward(self, x_metrical, x, edge_index, lengths):
        """edge_index[0] = note, edge_index[1] = beat/measure.  Returns (per-note output, per-beat state)."""
        _lib.require_gpu(x_metrical, x)
        from .gru import gru_forward
        nm = x_metrical.size(0)
        if lengths is None:
            lengths = torch.tensor([nm], dtype=torch.long, device=x.device)
        ragged = not bool(torch.all(lengths == lengths[0]))
        h_scatter = _scatter_rows(self.neigh(x), edge_index[1], edge_index[0], nm)               # notes -> beats, :511
        z_s = torch.cat((h_scatter, x_metrical), dim=-1)
        if ragged:
            sizes = torch.diff(lengths).tolist()
            from torch.nn.utils.rnn import pad_sequence
            z_s = pad_sequence(torch.split(z_s, sizes), batch_first=True)
            h_metrical = pad_sequence(torch.split(h_scatter, sizes), batch_first=True)
        else:
            L0 = int(lengths[0])
            h_metrical = h_scatter.view(-1, L0, h_scatter.shape[1])
            z_s = z_s.view(-1, L0, z_s.shape[1])
        h_seq = gru_forward(self.seq, h_metrical, self.training)
        h = self.activation(self.conv_out(torch.cat([z_s, h_seq], dim=-1)))
        h = self.dropout(self.normalize(h.transpose(1, 2))).transpose(1, 2)                        # BatchNorm over [B, C, T]
        if ragged:
            keep = torch.arange(h.shape[1], device=h.device).unsqueeze(0) < torch.tensor(sizes, device=h.device).unsqueeze(1)
            h = h[keep].view(-1, h.shape[-1])
        else:
            h = h.reshape(-1, h.shape[-1])          # the reference's .view() raises here for > 1 sequence (gnn.py:538)
        out = _scatter_rows(h, edge_index[0], edge_index[1], x.size(0))                            # beats -> notes, :539
        return out, h


class MetricalGNN(nn.Module):
    """In-tree MetricalGNN (core/hgnn.py:323-433): metrical=True/False, any in-tree conv block (SageConvScatter,
    ResGatedGraphConv, RelEdgeConv: models/chord.py:521-528), `use_reledge` (edge features into the first layer, :344-346,
    :416-417) and `jk` (the reference builds JumpingKnowledge(n_layers=hidden_features), :340 — kept as it is)."""

    def __init__(self, input_features, hidden_features, output_features, etypes, num_layers=2, dropout=0.5,
                 use_reledge=False, jk=False, in_edge_features=None, metrical=False, conv_block=SageConvScatter):
        super().__init__()
        self.dropout, self.num_layers, self.num_hidden = dropout, num_layers, hidden_features
        self.use_metrical = metrical
        self.use_reledge = use_reledge
        self.use_knowledge = bool(jk)
        self.convs = nn.ModuleList()
        self.emb_beats = nn.Linear(input_features, hidden_features)
        self.emb_measures = nn.Linear(input_features, hidden_features)
        self.beat_convs, self.measure_convs, self.project_metrical = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        if jk:
            self.jk = JumpingKnowledge(n_hidden=hidden_features, n_layers=hidden_features)
        self.convs.append(HeteroConv(input_features, hidden_features, etypes=etypes,
                                     in_edge_features=in_edge_features if use_reledge else None, module=conv_block))
        for _ in range(max(num_layers - 2, 0)):
            self.convs.append(HeteroConv(hidden_features, hidden_features, etypes=etypes, module=conv_block))
            if metrical:
                self.beat_convs.append(MetricalConvLayer(hidden_features, hidden_features, activation=F.relu, dropout=dropout))
                self.measure_convs.append(MetricalConvLayer(hidden_features, hidden_features, activation=F.relu, dropout=dropout))
                self.project_metrical.append(nn.Linear(hidden_features * 3, hidden_features))
        self.convs.append(HeteroConv(hidden_features, hidden_features, etypes=etypes, module=conv_block))
        if metrical:
            self.beat_convs.append(MetricalConvLayer(hidden_features, output_features, activation=F.relu, dropout=dropout))
            self.measure_convs.append(MetricalConvLayer(hidden_features, output_features, activation=F.relu, dropout=dropout))
            self.project_metrical.append(nn.Linear(output_features * 3, output_features))

    def forward(self, x, edge_index, edge_type, beat_nodes, measure_nodes, beat_edges, measure_edges, rel_edge=None,
                beat_lengths=None, measure_lengths=None, **kwargs):
        _lib.require_gpu(x)
        if self.use_metrical:
            h_beat = _scatter_rows(self.emb_beats(x), beat_edges[1], beat_edges[0], beat_nodes.size(0))
            h_measure = _scatter_rows(self.emb_measures(x), measure_edges[1], measure_edges[0], measure_nodes.size(0))

        def metrical(k, h, h_beat, h_measure):
            bc, h_beat = self.beat_convs[k](h_beat, h, beat_edges, beat_lengths)
            mc, h_measure = self.measure_convs[k](h_measure, h, measure_edges, measure_lengths)
            h = self.project_metrical[k](torch.cat([h, bc, mc], dim=-1))
            return F.normalize(F.relu(h), p=2.0, dim=-1), h_beat, h_measure

        h = x
        hs = []
        for i in range(len(self.convs) - 1):
            if i != 0 and self.use_metrical:
                h, h_beat, h_measure = metrical(i - 1, h, h_beat, h_measure)
            if i == 0 and self.use_reledge:                                   # first layer only: :416-417
                h = self.convs[i](h, edge_index, edge_type, edge_features=rel_edge)
            else:
                h = self.convs[i](h, edge_index, edge_type)
            h = F.dropout(F.relu(F.normalize(h, p=2.0, dim=-1)), p=self.dropout, training=self.training)
            hs.append(h)
        if self.use_knowledge:
            h = self.jk(hs)
        if self.use_metrical:
            h, h_beat, h_measure = metrical(len(self.beat_convs) - 1, h, h_beat, h_measure)
        return self.convs[-1](h, edge_index, edge_type)
