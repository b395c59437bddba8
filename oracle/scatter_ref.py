"""CPU restatement of the three `torch_scatter` functions the reference's hot path calls.

TEST INFRASTRUCTURE (see oracle/__init__.py).

`torch_scatter` (pinned `>=2.1.0`, /root/reference/requirements.txt:5) is a third-party
compiled package that is absent from /root/reference and from this image.  Its published
semantics are restated here:

* call sites: core/gnn.py:74,104,149,208,256,309,511,539; core/hgnn.py:406-407;
  models/analysis.py:66,586,1239; models/cadence.py:204,329.
* `sum`/`add`: when `out` is given, ACCUMULATE into its existing contents; otherwise zeros of
  `dim_size` rows (or `index.max()+1`).
* `mean`: `out = scatter_sum(src, index, out=out)`; `count = scatter_sum(ones)`;
  `count.clamp_(min=1)`; `out /= count` — the whole `out`, pre-filled part included, is
  divided by the neighbour count (true division for floating point).

PARITY: pinned only by the hand-computed known-answer tests in tests/test_oracle_scatter.py
(the reference holds no vector for these functions) — "unpinned vs the real torch_scatter".
Only dim=0 with a 1-D index is supported: that is every call site listed above.
"""
from __future__ import annotations

from typing import Optional

import torch


def _prep(src: torch.Tensor, index: torch.Tensor, dim: int, out, dim_size):
    if dim != 0:
        raise NotImplementedError("oracle scatter: dim=0 only (all reference call sites)")
    if index.dim() != 1 or index.numel() != src.shape[0]:
        raise ValueError("oracle scatter: index must be 1-D over dim 0")
    if out is None:
        if dim_size is None:
            dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
        out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    return out


def scatter_sum(src, index, dim: int = 0, out: Optional[torch.Tensor] = None,
                dim_size: Optional[int] = None) -> torch.Tensor:
    out = _prep(src, index, dim, out, dim_size)
    if index.numel() == 0:
        return out
    return out.index_add_(0, index, src) if not out.requires_grad else out.index_add(0, index, src)


scatter_add = scatter_sum


def scatter_mean(src, index, dim: int = 0, out: Optional[torch.Tensor] = None,
                 dim_size: Optional[int] = None) -> torch.Tensor:
    out = scatter_sum(src, index, dim, out, dim_size)
    n = out.shape[0]
    count = torch.zeros(n, dtype=src.dtype, device=src.device)
    if index.numel() > 0:
        count.index_add_(0, index, torch.ones(index.numel(), dtype=src.dtype, device=src.device))
    count.clamp_(min=1)
    shape = (n,) + (1,) * (out.dim() - 1)
    if out.requires_grad:
        return out / count.view(shape)
    out.div_(count.view(shape))
    return out


def scatter(src, index, dim: int = 0, out: Optional[torch.Tensor] = None,
            dim_size: Optional[int] = None, reduce: str = "sum") -> torch.Tensor:
    if reduce in ("sum", "add"):
        return scatter_sum(src, index, dim, out, dim_size)
    if reduce == "mean":
        return scatter_mean(src, index, dim, out, dim_size)
    raise NotImplementedError(f"oracle scatter: reduce={reduce!r} is not on the hot path")
