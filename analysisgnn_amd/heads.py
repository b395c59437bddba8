"""Fused task heads and fused multi-task cross entropy (the wrapper epilogue).

The reference runs 21 independent `Linear -> ReLU -> LayerNorm -> Linear` heads
(analysisgnn/models/analysis.py:486-496, :546-548) and 21 `CrossEntropyLoss(ignore_index=-1,
label_smoothing=0.1)` terms (:881-888, models/chord.py:39-49): ~600 tiny launches per step whose
weight-gradient GEMMs are 64 x 128 outputs with K = N (4 workgroups on a 256-CU chip).  Same parameters
(`clf_dict.<task>.{0,2,3}` stay ordinary modules, `state_dict` unchanged), different schedule:
  * first layers stacked into ONE GEMM  [N, o] x [o, T*h2]
  * ReLU + LayerNorm over each task's h2-wide group in one pass on the [N, T, h2] view
  * second layers as ONE grouped projection (`agnn_gproj_*`: task t reads its own h2 columns and writes its own C_t
    logit columns side by side [N, sum C]; a block-diagonal library GEMM did 21x the useful FLOPs)
  * one C-ABI kernel for all T cross-entropy terms and their gradient (`agnn_multitask_ce_f32`).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .fused import grouped_norm_act
from .linear import linear, mark_wgrad_async, wgrad_stream
from .params import cat_rows, stack_rows


def fused_head_logits(clf_dict: nn.ModuleDict, x: torch.Tensor, tasks: Sequence[str]) -> Tuple[torch.Tensor, List[int]]:
    """Concatenated logits [N, sum_t C_t] of the given tasks and the segment offsets (len T+1)."""
    mods = [clf_dict[t] for t in tasks]
    T = len(mods)
    h2 = mods[0][0].out_features
    # cat / stack of leaf parameters: their backward is narrow / unbind (views), so these gradients may arrive late;
    # the cats themselves are views when the parameters are adjacent in memory (dp.plan_parameters), launches otherwise
    W1 = mark_wgrad_async(cat_rows([m[0].weight for m in mods]))          # [T*h2, o]
    b1 = mark_wgrad_async(cat_rows([m[0].bias for m in mods]))
    a = linear(x, W1, b1)                                                 # [N, T*h2]
    gamma = stack_rows([m[2].weight for m in mods])                       # [T, h2]
    beta = stack_rows([m[2].bias for m in mods])
    a = grouped_norm_act(a.view(-1, T, h2), gamma, beta, mods[0][2].eps, pre_relu=True)   # ReLU + per-task LayerNorm, one launch
    offs = [0]
    for m in mods:
        offs.append(offs[-1] + m[3].out_features)
    b2 = cat_rows([m[3].bias for m in mods])
    a = a.reshape(-1, T * h2)
    if a.is_cuda and h2 in GPROJ_K and T <= _lib.MAX_SEG and GPROJ_ENABLED:
        W2 = mark_wgrad_async(cat_rows([m[3].weight for m in mods]))           # [sum C, h2]
        logits = grouped_projection(a, W2, mark_wgrad_async(b2), offs, h2)
    else:                                                                 # widths the kernel is not built for
        W2 = torch.block_diag(*[m[3].weight for m in mods])               # [sum C, T*h2]
        logits = linear(a, W2, b2)
    return logits, offs


GPROJ_K = (32, 64, 128)
GPROJ_ENABLED = True     # A/B switch for benchmarking


def _offs_tensor(offs: Sequence[int], device) -> torch.Tensor:
    key = (tuple(offs), str(device))
    t = _OFFS_CACHE.get(key)
    if t is None:                            # host -> device once per head layout (keeps the step graph-capturable)
        t = _OFFS_CACHE[key] = torch.tensor(list(offs), dtype=torch.int32, device=device)
    return t


class _GroupedProj(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, w, b, offs_t, offs, K):
        dev = _lib.require_gpu(a, w, offs_t)
        ctx._wg_async_in = all(t is None or t.is_leaf or getattr(t, "_agnn_wgrad_async", False) for t in (w, b))
        a = _lib.f32c(a)
        w = _lib.f32c(w)
        if w.data_ptr() % 16:
            w = w.clone()
        G = len(offs) - 1
        sum_c = offs[-1]
        tiles = sum((offs[i + 1] - offs[i] + 31) // 32 for i in range(G))
        N = a.shape[0]
        out = torch.empty((N, sum_c), dtype=torch.float32, device=dev)
        lib = _lib.load()
        bb = b.float().contiguous() if b is not None else None
        _lib.check(lib.agnn_gproj_fwd_f32(a.data_ptr(), a.stride(0), w.data_ptr(), _lib.ptr(bb), offs_t.data_ptr(), G, K, tiles, N,
                                          out.data_ptr(), out.stride(0), _lib.stream_ptr(dev)), "agnn_gproj_fwd_f32")
        ctx.save_for_backward(a, w, offs_t)
        ctx.meta = (G, K, tiles, sum_c, b is not None)
        ctx.wg_async = ctx._wg_async_in
        return out

    @staticmethod
    def backward(ctx, dout):
        a, w, offs_t = ctx.saved_tensors
        G, K, tiles, sum_c, has_b = ctx.meta
        dev = a.device
        dout = _lib.f32c(dout)
        N = a.shape[0]
        lib = _lib.load()
        need_w = ctx.needs_input_grad[1] or (has_b and ctx.needs_input_grad[2])
        dw = db = None
        if need_w:                                   # optimizer-only outputs: weight-gradient stream (linear.py)
            with wgrad_stream(dev, dout, a, active=ctx.wg_async):
                dw = torch.empty_like(w)
                db = torch.empty((sum_c,), dtype=torch.float32, device=dev) if has_b else None
                nws = int(lib.agnn_gproj_workspace_bytes(N, sum_c, K, tiles))
                ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=dev)
                _lib.check(lib.agnn_gproj_bwd_f32(dout.data_ptr(), dout.stride(0), a.data_ptr(), a.stride(0), w.data_ptr(),
                                                  offs_t.data_ptr(), G, K, tiles, sum_c, N, None, 0, dw.data_ptr(), _lib.ptr(db),
                                                  ws.data_ptr(), nws, _lib.stream_ptr(dev)), "agnn_gproj_bwd_f32")
        da = None
        if ctx.needs_input_grad[0]:
            da = torch.empty_like(a)
            _lib.check(lib.agnn_gproj_bwd_f32(dout.data_ptr(), dout.stride(0), a.data_ptr(), a.stride(0), w.data_ptr(),
                                              offs_t.data_ptr(), G, K, tiles, sum_c, N, da.data_ptr(), da.stride(0), None, None,
                                              None, 0, _lib.stream_ptr(dev)), "agnn_gproj_bwd_f32")
        return da, dw, db, None, None, None


def grouped_projection(a: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], offs: Sequence[int], K: int) -> torch.Tensor:
    """out[:, offs[g]:offs[g+1]] = a[:, g*K:(g+1)*K] @ w[offs[g]:offs[g+1]].T + b[offs[g]:offs[g+1]]  for every group g."""
    return _GroupedProj.apply(a, w, b, _offs_tensor(offs, a.device), tuple(int(o) for o in offs), int(K))


class _MultiTaskCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, offs_t, eps: float, ignore_index: int, full_cover: bool = False):
        dev = _lib.require_gpu(logits, labels, offs_t)
        if logits.dtype != torch.float32 or logits.stride(1) != 1:
            logits = logits.float().contiguous()
        N = logits.shape[0]
        T = offs_t.numel() - 1
        labels = labels.contiguous()
        row_loss = torch.empty((T, N), dtype=torch.float32, device=dev)
        loss = torch.zeros((T,), dtype=torch.float32, device=dev)
        inv_cnt = torch.ones((T,), dtype=torch.float32, device=dev)
        # the kernel writes every column of every segment: zero-fill only when the segments leave columns uncovered
        dlogits = torch.empty_like(logits) if full_cover else torch.zeros_like(logits)
        lib = _lib.load()
        _lib.check(lib.agnn_multitask_ce_f32(logits.data_ptr(), logits.stride(0), offs_t.data_ptr(), T, labels.data_ptr(), N,
                                             float(eps), int(ignore_index), row_loss.data_ptr(), dlogits.data_ptr(),
                                             loss.data_ptr(), inv_cnt.data_ptr(), _lib.stream_ptr(dev)), "agnn_multitask_ce_f32")
        ctx.save_for_backward(dlogits, offs_t, inv_cnt)
        return loss                                                        # [T] mean loss per task

    @staticmethod
    def backward(ctx, g):
        dlogits, offs_t, inv_cnt = ctx.saved_tensors
        dev = dlogits.device
        T = offs_t.numel() - 1
        scale = (g.to(torch.float32) * inv_cnt).contiguous()
        out = torch.empty_like(dlogits)
        lib = _lib.load()
        _lib.check(lib.agnn_multitask_ce_scale_f32(dlogits.data_ptr(), dlogits.stride(0), offs_t.data_ptr(), T, dlogits.shape[0],
                                                   dlogits.shape[1], scale.data_ptr(), out.data_ptr(), out.stride(0),
                                                   _lib.stream_ptr(dev)),
                   "agnn_multitask_ce_scale_f32")
        return out, None, None, None, None, None


def multitask_cross_entropy(logits: torch.Tensor, offs: Sequence[int], labels: torch.Tensor, label_smoothing: float = 0.1,
                            ignore_index: int = -1) -> torch.Tensor:
    """Per-task mean losses [T] for side-by-side logits [N, sum C]; labels int64 [T, N]."""
    full = len(offs) > 1 and offs[0] == 0 and offs[-1] == logits.shape[1] and all(offs[i] < offs[i + 1] for i in range(len(offs) - 1))
    return _MultiTaskCE.apply(logits, labels, _offs_tensor(offs, logits.device), label_smoothing, ignore_index, full)


_OFFS_CACHE: Dict[tuple, torch.Tensor] = {}
_LOSS_WS: Dict[str, torch.Tensor] = {}


class _TrainLoss(torch.autograd.Function):
    """total = sum_t CE_t + lam * mean(feat^2) (agnn_train_loss_f32: two launches) and its gradient w.r.t. the logits and
    feat (agnn_train_loss_bwd_f32: one launch)."""

    @staticmethod
    def forward(ctx, logits, labels, offs_t, feat, eps: float, ignore_index: int, lam: float):
        dev = _lib.require_gpu(logits, labels, offs_t, feat)
        if logits.dtype != torch.float32 or logits.stride(1) != 1:
            logits = logits.float().contiguous()
        if feat.dtype != torch.float32 or feat.stride(1) != 1:
            feat = feat.float().contiguous()
        N = logits.shape[0]
        T = offs_t.numel() - 1
        labels = labels.contiguous()
        lib = _lib.load()
        ws = _LOSS_WS.get(str(dev))                           # one per device: calls on one device are assumed not to overlap
        if ws is None:                                        # zero-filled once; every call leaves it zero-filled
            ws = _LOSS_WS[str(dev)] = torch.zeros(int(lib.agnn_train_loss_workspace_bytes()) + 256, dtype=torch.uint8, device=dev)
        wsp = (ws.data_ptr() + 255) & ~255
        row_loss = torch.empty((T, N), dtype=torch.float32, device=dev)
        out = torch.empty((2 * T + 1,), dtype=torch.float32, device=dev)      # loss[T] | inv_cnt[T] | total
        dlogits = torch.empty_like(logits)
        _lib.check(lib.agnn_train_loss_f32(logits.data_ptr(), logits.stride(0), offs_t.data_ptr(), T, labels.data_ptr(), N, float(eps),
                                           int(ignore_index), feat.data_ptr(), feat.stride(0), feat.shape[1], float(lam),
                                           row_loss.data_ptr(), dlogits.data_ptr(), out.data_ptr(), out[T:].data_ptr(),
                                           out[2 * T:].data_ptr(), wsp, int(lib.agnn_train_loss_workspace_bytes()),
                                           _lib.stream_ptr(dev)), "agnn_train_loss_f32")
        ctx.save_for_backward(dlogits, offs_t, out, feat)
        ctx.lam = float(lam)
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)                      # no zero-filled gradient for the logging output
        return out[2 * T], out

    @staticmethod
    def backward(ctx, g, _g_parts):
        dlogits, offs_t, out, feat = ctx.saved_tensors
        dev = dlogits.device
        T = offs_t.numel() - 1
        g = g.to(torch.float32).contiguous()
        dl = torch.empty_like(dlogits)
        dfeat = torch.empty_like(feat) if ctx.needs_input_grad[3] else None
        lib = _lib.load()
        _lib.check(lib.agnn_train_loss_bwd_f32(dlogits.data_ptr(), dlogits.stride(0), offs_t.data_ptr(), T, dlogits.shape[0],
                                               dlogits.shape[1], out[T:].data_ptr(), g.data_ptr(), dl.data_ptr(), dl.stride(0),
                                               feat.data_ptr(), feat.stride(0), feat.shape[1], ctx.lam, _lib.ptr(dfeat),
                                               dfeat.stride(0) if dfeat is not None else 0, _lib.stream_ptr(dev)),
                   "agnn_train_loss_bwd_f32")
        return dl, None, None, dfeat, None, None, None


def training_loss(logits: torch.Tensor, offs: Sequence[int], labels: torch.Tensor, feat: torch.Tensor, lambda_feat: float = 0.1,
                  label_smoothing: float = 0.1, ignore_index: int = -1) -> Tuple[torch.Tensor, torch.Tensor]:
    """(total, per_task) with total = sum_t CrossEntropy_t + lambda_feat * feat.pow(2).mean() — the reference's training
    objective without its optional terms (models/analysis.py:881-888, :984, :1072) — in two launches forward and one
    backward.  per_task [T] are the mean losses per task (for logging; not differentiable).  Requires segments that
    cover the logits' columns side by side (what fused_head_logits produces)."""
    full = len(offs) > 1 and offs[0] == 0 and offs[-1] == logits.shape[1] and all(offs[i] < offs[i + 1] for i in range(len(offs) - 1))
    if not full:
        per_task = multitask_cross_entropy(logits, offs, labels, label_smoothing, ignore_index)
        return per_task.sum() + lambda_feat * feat.pow(2).mean(), per_task.detach()
    total, out = _TrainLoss.apply(logits, labels, _offs_tensor(offs, logits.device), feat, label_smoothing, ignore_index, lambda_feat)
    return total, out[:len(offs) - 1]

