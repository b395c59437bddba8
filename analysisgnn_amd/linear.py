"""Projection layers whose weight / bias gradients run on the split-N fp32-MFMA kernel `agnn_wgrad_f32`.

Forward and input-gradient GEMMs are ordinary library GEMMs (well shaped: N x in x out with N = 16 000); only
dW = dY^T X (tiny output, reduction over N) is mis-served by the library heuristics — see csrc/wgrad.hip.
`Linear` subclasses `nn.Linear`: same parameters, same `state_dict`."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

MIN_ROWS = 2048          # below this the library GEMM is fine
MAX_OUT_IN = 512 * 1024  # above this output size the library GEMM (large tiles, no slabs) is faster (scripts/bench_wgrad.py)
ENABLED = True           # A/B switch for benchmarking


def _ok(t: torch.Tensor) -> bool:
    return (t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 2 == 0
            and t.shape[1] % 2 == 0 and t.data_ptr() % 8 == 0)


def weight_grad(dy: torch.Tensor, x: torch.Tensor, want_bias: bool):
    """(dW [out, in], db [out] or None) for dy [N, out], x [N, in] on the HIP kernel; library GEMM when the
    shape / alignment does not fit the kernel."""
    n, out_f = dy.shape
    in_f = x.shape[1]
    if not (ENABLED and dy.is_cuda and n >= MIN_ROWS and out_f * in_f <= MAX_OUT_IN and _ok(dy) and _ok(x)):
        return dy.t() @ x, (dy.sum(dim=0) if want_bias else None)
    lib = _lib.load()
    dev = dy.device
    dw = torch.empty((out_f, in_f), dtype=torch.float32, device=dev)
    db = torch.empty((out_f,), dtype=torch.float32, device=dev) if want_bias else None
    nws = int(lib.agnn_wgrad_workspace_bytes(n, out_f, in_f))
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    _lib.check(lib.agnn_wgrad_f32(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), n, out_f, in_f, dw.data_ptr(),
                                  dw.stride(0), _lib.ptr(db), ws.data_ptr(), nws, _lib.stream_ptr(dev)), "agnn_wgrad_f32")
    return dw, db


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, acc):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        if acc is not None:                       # y = acc + x W^T (+ b): the GEMM's beta = 1 epilogue, no separate add
            y = torch.addmm(acc, x, w.t())
            return y + b if b is not None else y
        return torch.addmm(b, x, w.t()) if b is not None else x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = dy @ w if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = weight_grad(dy, x, ctx.has_bias and ctx.needs_input_grad[2])
        return dx, dw, db, (dy if ctx.needs_input_grad[3] else None)


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None, acc: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x W^T (+ b) (+ acc)."""
    if x.is_cuda and torch.is_grad_enabled() and (w.requires_grad or (b is not None and b.requires_grad)):
        if x.dim() == 2 and x.shape[0] >= MIN_ROWS:
            return _LinearFn.apply(x, w, b, acc)
        if x.dim() == 3 and x.shape[0] * x.shape[1] >= MIN_ROWS and acc is None:
            return _LinearFn.apply(x.reshape(-1, x.shape[-1]), w, b, None).view(x.shape[0], x.shape[1], -1)
    y = F.linear(x, w, b)
    return y + acc if acc is not None else y


class Linear(nn.Linear):
    def forward(self, x):
        return linear(x, self.weight, self.bias)
