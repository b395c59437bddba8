"""`torch_scatter`-compatible operator surface on the HIP kernels — the second drop-in point of
SURVEY.md §8(b): `scatter`, `scatter_add`, `scatter_sum`, `scatter_mean` with the `out=` accumulation
semantics of App. A.1 (sum accumulates into `out`; mean divides the WHOLE `out` by max(count, 1)).
Only dim=0 with a 1-D index (every call site of the reference: core/gnn.py:74,104,149,208,256,309,511,539;
core/hgnn.py:406-407; models/analysis.py:66,586,1239).  Returns a new tensor (the reference always uses the
return value); differentiable w.r.t. `src` and `out`.  No CPU path.

Registered with `torch.library` as `analysisgnn_amd::scatter_reduce` (fake / meta kernel + autograd formula), so a
`torch.compile(model, dynamic=True)` of code that calls these names (reference: train/train_analysisgnn.py:202-203 compiles
the model; the in-tree layers call `scatter(...)`) traces THROUGH the call — one opaque node in the FX graph — instead of
breaking the graph at it.  Eager calls take the same op.  (The encoders themselves stay behind `torch._dynamo.disable`:
their host side — trim plans, CSR memo, stream forks — is not traceable Python.)"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib, ops
from .graph import Csr, SegSpec, build_csr


def _scatter(src, index, dim, out, dim_size, mean: bool):
    if dim != 0 or index.dim() != 1 or index.numel() != src.shape[0]:
        raise NotImplementedError("analysisgnn_amd.scatter: dim=0 with a 1-D index over dim 0 only")
    if out is not None:
        n = out.shape[0]
    elif dim_size is not None:
        n = int(dim_size)
    else:
        n = int(index.max()) + 1 if index.numel() else 0      # data dependent: under torch.compile pass dim_size / out
    return _scatter_op(src, index, out, n, mean)


def _scatter_hip(src, index, out, n: int, mean: bool):
    """The launch sequence (CSR by destination, multi-relation gather-reduce kernel); called below autograd by the custom op."""
    _lib.require_gpu(src, index)
    shape = src.shape
    src2 = src.reshape(shape[0], -1)
    E = index.numel()
    ident = torch.arange(E, dtype=torch.int64, device=src.device)
    fwd, bwd = build_csr([SegSpec(index, ident, n), SegSpec(ident, index, E)])
    sp, W = ops.pad4(src2)
    self_t = None
    if out is not None:
        self_t, _ = ops.pad4(out.reshape(n, -1))
    spec = ops.AggSpec(fwd=[fwd], bwd=[bwd], src_id=[0], n_rows=n, mean=mean, shared_slot=True)
    with torch.no_grad():
        res = ops.aggregate(spec, [sp], self_t=self_t)
    res = res[:, :W].contiguous() if res.shape[1] != W else res
    return res.reshape((n,) + tuple(shape[1:]))


@torch.library.custom_op("analysisgnn_amd::scatter_reduce", mutates_args=())
def _scatter_op(src: torch.Tensor, index: torch.Tensor, out: Optional[torch.Tensor], n: int, mean: bool) -> torch.Tensor:
    return _scatter_hip(src, index, out, n, mean)


@_scatter_op.register_fake
def _(src, index, out, n, mean):
    return src.new_empty((n,) + tuple(src.shape[1:]))


def _scatter_ctx(ctx, inputs, output):
    src, index, out, n, mean = inputs
    ctx.mean, ctx.n, ctx.has_out = mean, n, out is not None
    ctx.save_for_backward(index)


def _scatter_bwd(ctx, g):
    """d/d src = the gathered rows of the (scaled) output gradient, d/d out = the (scaled) output gradient itself: result =
    (out + sum) / max(count, 1) for mean (App. A.1: the WHOLE accumulator is divided), out + sum otherwise."""
    (index,) = ctx.saved_tensors
    if ctx.mean:
        cnt = torch.bincount(index, minlength=ctx.n).clamp(min=1).to(g.dtype)
        g = g / cnt.reshape((ctx.n,) + (1,) * (g.dim() - 1))
    return g.index_select(0, index), None, (g if ctx.has_out else None), None, None


_scatter_op.register_autograd(_scatter_bwd, setup_context=_scatter_ctx)




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
