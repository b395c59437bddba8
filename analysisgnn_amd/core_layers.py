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


def _make_reduction(reduction: str):
    if reduction not in ("mean", "sum"):
        raise NotImplementedError(f"reduction={reduction!r}: only 'mean' and 'sum' run on the HIP path")
    return reduction


class HeteroSageConvLayer(nn.Module):
    """core/hgnn.py:98-140.  R SageConvScatter modules, one per relation, outputs reduced over
    the R slots.  Fused here: ONE GEMM for all W_n^r, ONE multi-relation gather-reduce, ONE GEMM
    over [x || s_1 .. s_R] with the root blocks of W^r pre-summed."""

    def __init__(self, in_features, out_features, etypes, bias=True, reduction="mean"):
        super().__init__()
        self.out_features = out_features
        self.in_features = in_features
        self.etypes = etypes
        self.reduction = _make_reduction(reduction)
        self.conv = nn.ModuleDict({k: SageConvScatter(in_features, out_features, bias=bias) for k in etypes.keys()})

    def reset_parameters(self):
        for c in self.conv.values():
            c.reset_parameters()

    def _fused(self, x, tix: TypedIndex):
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
