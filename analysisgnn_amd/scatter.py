"""`torch_scatter`-compatible operator surface on the HIP kernels — the second drop-in point of
SURVEY.md §8(b): `scatter`, `scatter_add`, `scatter_sum`, `scatter_mean` with the `out=` accumulation
semantics of App. A.1 (sum accumulates into `out`; mean divides the WHOLE `out` by max(count, 1)).
Only dim=0 with a 1-D index (every call site of the reference: core/gnn.py:74,104,149,208,256,309,511,539;
core/hgnn.py:406-407; models/analysis.py:66,586,1239).  Returns a new tensor (the reference always uses the
return value); differentiable w.r.t. `src` and `out`.  No CPU path."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib, ops
from .graph import Csr, SegSpec, build_csr


def _scatter(src, index, dim, out, dim_size, mean: bool):
    _lib.require_gpu(src, index)
    if dim != 0 or index.dim() != 1 or index.numel() != src.shape[0]:
        raise NotImplementedError("analysisgnn_amd.scatter: dim=0 with a 1-D index over dim 0 only")
    shape = src.shape
    src2 = src.reshape(shape[0], -1)
    if out is not None:
        n = out.shape[0]
    elif dim_size is not None:
        n = int(dim_size)
    else:
        n = int(index.max()) + 1 if index.numel() else 0
    E = index.numel()
    ident = torch.arange(E, dtype=torch.int64, device=src.device)
    fwd, bwd = build_csr([SegSpec(index, ident, n), SegSpec(ident, index, E)])
    sp, W = ops.pad4(src2)
    self_t = None
    if out is not None:
        self_t, _ = ops.pad4(out.reshape(n, -1))
    spec = ops.AggSpec(fwd=[fwd], bwd=[bwd], src_id=[0], n_rows=n, mean=mean, shared_slot=True)
    res = ops.aggregate(spec, [sp], self_t=self_t)
    res = res[:, :W] if res.shape[1] != W else res
    return res.reshape((n,) + tuple(shape[1:]))


def scatter_sum(src, index, dim: int = 0, out: Optional[torch.Tensor] = None, dim_size: Optional[int] = None):
    return _scatter(src, index, dim, out, dim_size, mean=False)


scatter_add = scatter_sum


def scatter_mean(src, index, dim: int = 0, out: Optional[torch.Tensor] = None, dim_size: Optional[int] = None):
    return _scatter(src, index, dim, out, dim_size, mean=True)


def scatter(src, index, dim: int = 0, out: Optional[torch.Tensor] = None, dim_size: Optional[int] = None,
            reduce: str = "sum"):
    if reduce in ("sum", "add"):
        return scatter_sum(src, index, dim, out, dim_size)
    if reduce == "mean":
        return scatter_mean(src, index, dim, out, dim_size)
    raise NotImplementedError(f"reduce={reduce!r} is not on the hot path")
