"""Lookup of the small pitch-spelling / key-signature tables (ref: analysisgnn/models/analysis.py:399-400, :566-569)
with a sort-free, sync-free backward.

`torch.nn.functional.embedding`'s backward sorts the indices and runs a chain of select / segmented-reduce kernels
(~20 launches, ~0.2 ms per step here, and its data-dependent segment bookkeeping does not belong in a captured
hipGraph).  The tables have 35 and 15 rows, so the gradient is simply  dW = onehot(idx)^T @ dY : one comparison kernel
and the split-N weight-gradient GEMM (`agnn_wgrad_f32`), summed in a fixed order."""
from __future__ import annotations

import torch
import torch.nn.functional as F

import ctypes as C
from typing import Sequence

from . import _lib
from .linear import all_steal, weight_grad, wgrad_stream

MAX_ROWS = 256      # beyond this a one-hot operand is the wrong tool; fall back to the library op


class _SmallEmbedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, idx, weight):
        ctx.save_for_backward(idx)
        ctx.rows = weight.shape[0]
        ctx.leaf = weight.is_leaf
        ctx.weight_ref = weight if weight.is_leaf else None
        return F.embedding(idx, weight)

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        V = ctx.rows
        Vp = V + (V & 1)                                                   # even width for the kernel
        with wgrad_stream(dy.device, dy, idx, active=ctx.leaf and all_steal((ctx.weight_ref,)), kind="embed"):   # optimizer-only output (linear.py)
            flat_idx = idx.reshape(-1)
            onehot = (flat_idx.unsqueeze(1) == torch.arange(Vp, device=idx.device)).to(dy.dtype)   # [N, Vp]
            dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
            dw, _ = weight_grad(onehot, dy2, False)                        # [Vp, D]
            dw = dw[:V]
        return None, dw


def embedding(idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    if weight.is_cuda and weight.requires_grad and torch.is_grad_enabled() and weight.shape[0] <= MAX_ROWS:
        return _SmallEmbedding.apply(idx, weight)
    return F.embedding(idx, weight)


# ------------------------------------------------------------------------------------------------------------
# Fused note-input assembly: cat([x, table_0[idx_0], table_1[idx_1], ...]) in ONE launch, rows padded to 16 bytes, and the
# tables' gradients in two (csrc/embed.hip).  ref: models/analysis.py:574.
# ------------------------------------------------------------------------------------------------------------
MAX_TABLES = 4


def _ptr_array(ts):
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


class _EmbedCat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n_tab, *rest):
        idxs, tabs = rest[:n_tab], rest[n_tab:]
        lib = _lib.load()
        dev = x.device
        n, in_x = int(x.shape[0]), int(x.shape[1])
        dim = int(tabs[0].shape[1])
        width = in_x + n_tab * dim
        ld = (width + 3) & ~3                                   # rows padded to 16 bytes; the spare columns are zero
        xc = x if (x.dtype == torch.float32 and x.stride(1) == 1) else x.float().contiguous()
        idc = [i if (i.dtype == torch.int64 and i.is_contiguous()) else i.long().contiguous() for i in idxs]
        tc = [t if t.is_contiguous() else t.contiguous() for t in tabs]
        vocab = (C.c_int32 * n_tab)(*[int(t.shape[0]) for t in tabs])
        out = torch.empty((n, ld), dtype=torch.float32, device=dev)
        _lib.check(lib.agnn_embed_cat_fwd_f32(xc.data_ptr(), xc.stride(0), in_x, n, n_tab, _ptr_array(idc), _ptr_array(tc), vocab,
                                              dim, out.data_ptr(), ld, _lib.stream_ptr(dev)), "agnn_embed_cat_fwd_f32")
        ctx.save_for_backward(*idc)
        ctx.meta = (n, in_x, n_tab, dim, [int(t.shape[0]) for t in tabs], all(t.is_leaf for t in tabs))
        ctx.tab_refs = [t for t in tabs if t.is_leaf]
        return out[:, :width]

    @staticmethod
    def backward(ctx, dout):
        idc = ctx.saved_tensors
        n, in_x, n_tab, dim, vocabs, leaf = ctx.meta
        lib = _lib.load()
        dev = dout.device
        if dout.stride(1) != 1 or dout.dtype != torch.float32:
            dout = dout.float().contiguous()
        with wgrad_stream(dev, dout, *idc, active=leaf and all_steal(ctx.tab_refs), kind="embed"):   # optimizer-only outputs (linear.py)
            vocab = (C.c_int32 * n_tab)(*vocabs)
            dtab = torch.empty((sum(vocabs), dim), dtype=torch.float32, device=dev)
            nws = int(lib.agnn_embed_workspace_bytes(n_tab, vocab, dim))
            ws = torch.empty(nws, dtype=torch.uint8, device=dev)
            _lib.check(lib.agnn_embed_cat_bwd_f32(dout.data_ptr(), dout.stride(0), in_x, n, n_tab, _ptr_array(idc), vocab, dim,
                                                  dtab.data_ptr(), ws.data_ptr(), nws, _lib.stream_ptr(dev)), "agnn_embed_cat_bwd_f32")
            grads = list(torch.split(dtab, vocabs, dim=0))          # views of one buffer: no kernel
        return (None, None, *([None] * n_tab), *grads)


def embed_cat(x: torch.Tensor, idxs: Sequence[torch.Tensor], tables: Sequence[torch.Tensor]) -> torch.Tensor:
    """cat([x, tables[0][idxs[0]], tables[1][idxs[1]], ...], dim=-1) on the HIP kernels.  The result is a column view of a
    buffer whose rows are padded to a multiple of 4 floats (zeros), which is what the projection that follows wants.
    The feature matrix x gets no gradient (it is an input)."""
    n_tab = len(tables)
    fits = (x.is_cuda and x.dim() == 2 and 1 <= n_tab <= MAX_TABLES and len(idxs) == n_tab and not x.requires_grad
            and all(t.dim() == 2 and t.shape[1] == tables[0].shape[1] and t.dtype == torch.float32 for t in tables)
            and tables[0].shape[1] % 2 == 0 and all(i.dim() == 1 and i.shape[0] == x.shape[0] for i in idxs))
    if not fits:
        return torch.cat([x] + [embedding(i, t) for i, t in zip(idxs, tables)], dim=-1)
    r = _EmbedCat.apply(x, n_tab, *idxs, *tables)
    # x gets no gradient: whoever consumes r only owes the gradient of the embedding columns (linear._LinearFn computes dX for
    # these columns alone: a [N, 256] x [256, 128] product instead of x [256, 281])
    r._agnn_grad_cols = (int(x.shape[1]), int(r.shape[1]))
    return r
