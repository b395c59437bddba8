"""Lookup of the small pitch-spelling / key-signature tables (ref: analysisgnn/models/analysis.py:399-400, :566-569)
with a sort-free, sync-free backward.

`torch.nn.functional.embedding`'s backward sorts the indices and runs a chain of select / segmented-reduce kernels
(~20 launches, ~0.2 ms per step here, and its data-dependent segment bookkeeping does not belong in a captured
hipGraph).  The tables have 35 and 15 rows, so the gradient is simply  dW = onehot(idx)^T @ dY : one comparison kernel
and the split-N weight-gradient GEMM (`agnn_wgrad_f32`), summed in a fixed order."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .linear import weight_grad, wgrad_stream

MAX_ROWS = 256      # beyond this a one-hot operand is the wrong tool; fall back to the library op


class _SmallEmbedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, idx, weight):
        ctx.save_for_backward(idx)
        ctx.rows = weight.shape[0]
        ctx.leaf = weight.is_leaf
        return F.embedding(idx, weight)

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        V = ctx.rows
        Vp = V + (V & 1)                                                   # even width for the kernel
        with wgrad_stream(dy.device, dy, idx, active=ctx.leaf):             # optimizer-only output (linear.py)
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
