"""Autograd wrappers over the C-ABI kernels (include/agnn.h).  Everything here runs on the HIP
library; tensors on CPU raise (`_lib.require_gpu`).

`aggregate` is the one primitive behind every gather/scatter site of the reference's hot path
(analysisgnn/models/core/gnn.py:70-74,:511,:539; core/hgnn.py:406-407; models/analysis.py:586;
PyG SAGEConv/HeteroConv via models/cadence.py:147-159): a multi-relation segmented
gather-reduce with torch_scatter's `out=` numerator semantics as an option.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import torch

from . import _lib
from .graph import Csr


@dataclass
class AggSpec:
    """Static description of one aggregation call (no tensors that need gradients)."""
    fwd: List[Csr]                       # per relation: rows = output rows, col = gathered rows
    bwd: List[Csr]                       # per relation: rows = gathered rows, col = output rows
    src_id: List[int]                    # relation -> position of its source matrix in `srcs`
    n_rows: int                          # output rows to compute (a prefix of the CSR rows)
    mean: bool = True
    shared_slot: bool = False            # True: relations summed into [n_rows, H]; False: [n_rows, R*H]
    e_limit: Optional[List[Optional[int]]] = None   # per relation COO prefix (trim_to_layer)
    skip_self: bool = False
    col_limit: int = _lib.INT32_MAX
    edge_weight: Optional[List[Optional[torch.Tensor]]] = None   # per relation, COO order, no grad
    root: bool = False                   # `self` is a SAGE layer's root operand: output [n_rows, (R+1)*H], last slot = self[:n_rows]

    def limit(self, r: int) -> Optional[int]:
        return None if self.e_limit is None else self.e_limit[r]


def _view_ok(t: torch.Tensor) -> bool:
    return (t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0
            and t.data_ptr() % 16 == 0)


# When set to a list, every SpMM launch is bracketed by HIP events on the launch stream and
# (tag, start_event, end_event, n_rel, n_rows, H, rel_stride) is appended — bench.py's live kernel timing.
SPMM_TRACE: Optional[list] = None
ROOT_WIDTHS = (256, 512)   # agnn_spmm_root_f32 exists for the widths of the specialised kernel
SPMM_VARIANT = 0           # tests / A-B timing: 1024 = AGNN_SPMM_GENERIC (never take the fast path)


def _launch(rels: Sequence[dict], n_rows: int, H: int, out: torch.Tensor, rel_stride: int,
            self_t: Optional[torch.Tensor], inv_cnt: Optional[torch.Tensor], col_limit: int, flags: int,
            tag: str = "spmm", root_rows: Optional[int] = None):
    lib = _lib.load()
    dev = out.device
    arr = _lib.make_rels(rels)
    trace = SPMM_TRACE
    if trace is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(dev))
    if root_rows is not None:
        rc = lib.agnn_spmm_root_f32(len(rels), arr, n_rows, H, out.data_ptr(), out.stride(0), rel_stride, self_t.data_ptr(),
                                    self_t.stride(0), root_rows, _lib.ptr(inv_cnt), min(int(col_limit), _lib.INT32_MAX), flags,
                                    _lib.stream_ptr(dev))
    else:
        rc = lib.agnn_spmm_f32(len(rels), arr, n_rows, H, out.data_ptr(), out.stride(0), rel_stride,
                               _lib.ptr(self_t), self_t.stride(0) if self_t is not None else 0,
                               _lib.ptr(inv_cnt), min(int(col_limit), _lib.INT32_MAX), flags | SPMM_VARIANT,
                               _lib.stream_ptr(dev))
    if trace is not None:
        e1.record(torch.cuda.current_stream(dev))
        trace.append((tag, e0, e1, len(rels), n_rows, H, rel_stride))
    _lib.check(rc, "agnn_spmm_root_f32" if root_rows is not None else "agnn_spmm_f32")


def _perm_weights(csr: Csr, w: torch.Tensor) -> torch.Tensor:
    """Per-edge weights (COO order of the segment) -> CSR position order."""
    full = torch.zeros(csr.col.numel(), dtype=torch.float32, device=w.device)
    lo = int(csr.rowptr[0])
    hi = int(csr.rowptr[-1])
    full[lo:hi] = w.index_select(0, csr.perm[lo:hi].long())
    return full


class _Aggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec: AggSpec, self_t: Optional[torch.Tensor], *srcs: torch.Tensor):
        dev = _lib.require_gpu(*srcs, self_t)
        H = srcs[0].shape[1]
        if H % 4 != 0:
            raise _lib.AgnnError(f"feature width {H} must be a multiple of 4 (pad with ops.pad4)")
        srcs_c = [(_lib.f32c(s) if not _view_ok(s) else s) for s in srcs]
        for s in srcs_c:
            if s.shape[1] != H:
                raise _lib.AgnnError("all source matrices must have the same width")
        R = len(spec.fwd)
        n = spec.n_rows
        self_c = None
        if self_t is not None:
            self_c = self_t if _view_ok(self_t) else _lib.f32c(self_t)
            if self_c.shape[0] < n or self_c.shape[1] != H:
                raise _lib.AgnnError("`self` must have at least n_rows rows and width H")
        if spec.root:
            if spec.shared_slot or self_c is None or spec.edge_weight is not None or H not in ROOT_WIDTHS:
                raise _lib.AgnnError("root aggregation: per-relation slots, a root operand, no edge weights, H in (256, 512)")
            width = (R + 1) * H
        else:
            width = H if spec.shared_slot else R * H
        out = torch.empty((n, width), dtype=torch.float32, device=dev)
        inv_cnt = torch.empty((R, max(n, 1)), dtype=torch.float32, device=dev) if spec.mean else None
        rels = []
        keep = []
        for r in range(R):
            csr = spec.fwd[r]
            if csr.n_rows < n:
                raise _lib.AgnnError(f"relation {r}: CSR has {csr.n_rows} rows < n_rows={n}")
            src = srcs_c[spec.src_id[r]]
            re = csr.rowend(spec.limit(r))
            ew = None
            if spec.edge_weight is not None and spec.edge_weight[r] is not None:
                ew = _perm_weights(csr, spec.edge_weight[r])
                keep.append(ew)
            rels.append(dict(src=src.data_ptr(), rowptr=csr.rowptr.data_ptr(), rowend=_lib.ptr(re),
                             col=csr.col.data_ptr(), ew=_lib.ptr(ew), ld_src=src.stride(0)))
        flags = (_lib.SPMM_MEAN if spec.mean else 0) | (_lib.SPMM_SKIP_SELF if spec.skip_self else 0)
        if n > 0:
            _launch(rels, n, H, out, 0 if spec.shared_slot else H, self_c, inv_cnt, spec.col_limit, flags, tag="fwd",
                    root_rows=n if spec.root else None)
        ctx.spec = spec
        ctx.H = H
        ctx.src_rows = [s.shape[0] for s in srcs]
        ctx.has_self = self_t is not None
        ctx.self_rows = self_t.shape[0] if self_t is not None else 0
        # `self` and a source are the same matrix (onset pooling): its two gradients are produced as one
        ctx.self_src = -1
        if self_t is not None:
            for k, s_ in enumerate(srcs):
                if s_.data_ptr() == self_t.data_ptr() and s_.shape == self_t.shape and s_.stride() == self_t.stride():
                    ctx.self_src = k
                    break
        ctx.save_for_backward(*( [inv_cnt] if inv_cnt is not None else [] ))
        return out

    @staticmethod
    def backward(ctx, dout: torch.Tensor):
        spec: AggSpec = ctx.spec
        H = ctx.H
        R = len(spec.fwd)
        n = spec.n_rows
        inv_cnt = ctx.saved_tensors[0] if spec.mean else None
        dout = dout if _view_ok(dout) else _lib.f32c(dout)
        dev = dout.device
        grads: List[Optional[torch.Tensor]] = []
        root_folded = False
        n_srcs = len(ctx.src_rows)
        for k in range(n_srcs):
            if not ctx.needs_input_grad[2 + k]:
                grads.append(None)
                continue
            n_src = ctx.src_rows[k]
            rows_t = min(n_src, int(spec.col_limit))
            g = (torch.empty if rows_t == n_src else torch.zeros)((n_src, H), dtype=torch.float32, device=dev)
            rels = []
            keep = []
            for r in range(R):
                if spec.src_id[r] != k:
                    continue
                csr = spec.bwd[r]
                if csr.n_rows < rows_t:
                    raise _lib.AgnnError("transposed CSR smaller than the source matrix")
                slot = dout if spec.shared_slot else dout[:, r * H:(r + 1) * H]
                re = csr.rowend(spec.limit(r))
                ew = None
                if spec.edge_weight is not None and spec.edge_weight[r] is not None:
                    ew = _perm_weights(csr, spec.edge_weight[r])
                    keep.append(ew)
                rels.append(dict(src=slot.data_ptr(), rowptr=csr.rowptr.data_ptr(), rowend=_lib.ptr(re),
                                 col=csr.col.data_ptr(), ew=_lib.ptr(ew),
                                 colscale=(inv_cnt[r].data_ptr() if inv_cnt is not None else None),
                                 ld_src=dout.stride(0)))
            # root aggregation: the root slot's gradient is one more addend of this launch when the root operand IS this
            # source (x_dst of a note -> note relation), so neither a padded copy nor a gradient add is ever issued
            fold_root = spec.root and k == ctx.self_src and ctx.needs_input_grad[1] and rels and rows_t >= n > 0
            if not rels or rows_t == 0 or n == 0:
                g.zero_()
            else:
                flags = _lib.SPMM_SKIP_SELF if spec.skip_self else 0
                # only a trimmed forward (fewer output rows than the CSR has) needs the column filter
                lim = n if any(c.n_rows > n for c in spec.fwd) else _lib.INT32_MAX
                if fold_root:
                    _launch(rels, rows_t, H, g, 0, dout[:, R * H:], None, lim, flags, tag="bwd", root_rows=n)
                    root_folded = True
                else:
                    _launch(rels, rows_t, H, g, 0, None, None, lim, flags, tag="bwd")
            grads.append(g)
        gself = None
        if spec.root:
            if ctx.needs_input_grad[1] and not root_folded:
                gself = torch.zeros((ctx.self_rows, H), dtype=torch.float32, device=dev)
                gself[:n] = dout[:, R * H:]
        elif ctx.has_self and ctx.needs_input_grad[1] and n > 0:
            # d/d self = sum_r dout_r * (1/cnt_r | 1): one launch (agnn_spmm_self_grad_f32), added straight onto the source's
            # gradient when self and that source are one matrix (autograd would otherwise add the two with another launch)
            lib = _lib.load()
            k = ctx.self_src
            fold = k >= 0 and grads[k] is not None
            if fold:
                tgt, acc = grads[k], 1
            else:
                tgt = (torch.zeros if ctx.self_rows > n else torch.empty)((ctx.self_rows, H), dtype=torch.float32, device=dev)
                acc = 0
            _lib.check(lib.agnn_spmm_self_grad_f32(dout.data_ptr(), dout.stride(0), 0 if spec.shared_slot else H, R, _lib.ptr(inv_cnt),
                                                   inv_cnt.stride(0) if inv_cnt is not None else 0, n, H, tgt.data_ptr(), tgt.stride(0),
                                                   acc, _lib.stream_ptr(dev)), "agnn_spmm_self_grad_f32")
            gself = None if fold else tgt
        elif ctx.has_self and ctx.needs_input_grad[1]:
            gself = torch.zeros((ctx.self_rows, H), dtype=torch.float32, device=dev)
        return (None, gself, *grads)


def aggregate(spec: AggSpec, srcs: Sequence[torch.Tensor], self_t: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[i, r*H:(r+1)*H] (or summed over r) = reduce_{p in row i of relation r} srcs[src_id[r]][col[p]]
    with optional torch_scatter-style `self` numerator.  Differentiable w.r.t. srcs and self."""
    return _Aggregate.apply(spec, self_t, *srcs)


def pad4(x: torch.Tensor):
    """Pad the feature dim to a multiple of 4 floats (kernel rows are 16-byte vectors)."""
    H = x.shape[1]
    Hp = (H + 3) & ~3
    if Hp == H:
        return x, H
    return torch.nn.functional.pad(x, (0, Hp - H)), H
