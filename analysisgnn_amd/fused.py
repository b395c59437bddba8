"""Fused [ReLU ->] LayerNorm [-> ReLU] [-> dropout] (C-ABI `agnn_norm_act_*`) and an `nn.Sequential` drop-in that
recognises those chains.  Parameters stay in the ordinary `nn.LayerNorm` / `nn.Linear` modules (`state_dict`
unchanged); only the schedule differs.  Dropout uses the library's own counter-based generator: masks are a pure
function of (seed, training step, call site, element), so backward regenerates them instead of storing them."""
from __future__ import annotations

import itertools
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .linear import ColsumItem, all_steal, defer, deferring, leaf_refs

PRE_RELU, POST_RELU = 1, 2
ENABLED = True
_RNG: Dict[str, torch.Tensor] = {}
_CALL_IDS = itertools.count(1)


def rng_state(device) -> torch.Tensor:
    """Device int64[2] = (seed, step).  `advance_rng` bumps the step (one tiny in-stream launch, graph-safe)."""
    key = str(device)
    if key not in _RNG:
        _RNG[key] = torch.tensor([int(torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=device)
    return _RNG[key]


def advance_rng(device) -> None:
    rng_state(device)[1:2].add_(1)          # one in-place launch (indexing with [1] += 1 expands into three)


def _al16(t: torch.Tensor) -> torch.Tensor:
    """Parameters can be views into a flat optimizer buffer at any element offset; the kernel reads float4."""
    return t if t.data_ptr() % 16 == 0 else t.clone()


class _NormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, p, flags, call_id, seg=0, defer_ok=False, steal_refs=None, pre=None):
        """x [n, H]; gamma / beta [H]; statistics over segments of `seg` floats (0 = the whole row).  `defer_ok`: gamma / beta
        are leaves, or reach their leaves through views only (their gradients may be produced late: linear.defer_weight_grads).
        `pre` = [y, mean, rstd] already computed by a fused producer (heads.fused_head_logits): nothing is launched here, the
        node only carries the backward pass."""
        dev = _lib.require_gpu(x, gamma, beta)
        ctx.defer_ok = bool(defer_ok) or (gamma.is_leaf and beta.is_leaf)
        ctx.steal_refs = leaf_refs(gamma, beta) if steal_refs is None else tuple(steal_refs)   # reshaped operands: the caller names the leaves
        lib = _lib.load()
        x = x if (x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0) else x.contiguous()
        n, H = x.shape
        seg = int(seg) if seg else H
        rng = rng_state(dev) if p > 0 else None
        # the (seed, step) pair this call draws from is saved with it: the live counter moves on with every training forward
        # (models.encode -> advance_rng), and backward must regenerate THIS call's masks
        used = torch.empty(2, dtype=torch.int64, device=dev) if p > 0 else None
        gamma, beta = _al16(gamma.contiguous()), _al16(beta.contiguous())
        if pre is not None:
            y, mean, rstd = pre
        else:
            y = torch.empty((n, H), dtype=torch.float32, device=dev)
            mean = torch.empty((max(n, 1) * (H // seg),), dtype=torch.float32, device=dev)
            rstd = torch.empty((max(n, 1) * (H // seg),), dtype=torch.float32, device=dev)
            _lib.check(lib.agnn_norm_act_fwd_f32(x.data_ptr(), x.stride(0), gamma.data_ptr(), beta.data_ptr(), seg, n, H, float(eps), float(p),
                                                 int(flags), _lib.ptr(rng), int(call_id), y.data_ptr(), y.stride(0), mean.data_ptr(),
                                                 rstd.data_ptr(), _lib.ptr(used), _lib.stream_ptr(dev)), "agnn_norm_act_fwd_f32")
        ctx.save_for_backward(x, gamma, beta, mean, rstd, *([used] if used is not None else []))
        ctx.cfg = (float(eps), float(p), int(flags), int(call_id), seg)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, rstd, *used = ctx.saved_tensors
        eps, p, flags, call_id, seg = ctx.cfg
        dev = dy.device
        lib = _lib.load()
        dy = dy if (dy.stride(1) == 1 and dy.stride(0) % 4 == 0 and dy.data_ptr() % 16 == 0) else dy.contiguous()
        n, H = x.shape
        dx = torch.empty_like(x)
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(beta)
        nws = int(lib.agnn_norm_act_workspace_bytes(H))
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        rng = used[0] if p > 0 else None                     # the forward call's own (seed, step), not the live counter
        # dgamma / dbeta only feed the optimizer: with deferred weight gradients (linear.defer_weight_grads) their column-sum
        # launch leaves the chain too (the closure works on aliases: see linear._LinearFn.backward)
        later = ctx.defer_ok and deferring(dy) and all_steal(ctx.steal_refs)
        _lib.check(lib.agnn_norm_act_bwd_f32(x.data_ptr(), x.stride(0), gamma.data_ptr(), beta.data_ptr(), seg, n, H, eps, p, flags,
                                             _lib.ptr(rng), call_id, dy.data_ptr(), dy.stride(0), mean.data_ptr(), rstd.data_ptr(),
                                             dx.data_ptr(), dx.stride(0), None if later else dgamma.data_ptr(),
                                             None if later else dbeta.data_ptr(), ws.data_ptr(), nws, _lib.stream_ptr(dev)),
                   "agnn_norm_act_bwd_f32")
        if later:
            dg_k, db_k = dgamma.detach(), dbeta.detach()
            defer(ColsumItem(ws, n, H, dg_k, db_k), dev)       # pending column sums of a flush go out in one launch
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, None


_MIX_WS: Dict[str, torch.Tensor] = {}


def _ok16(t: torch.Tensor) -> bool:
    return t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0


class _SkipAct(torch.autograd.Function):
    """z = dropout_p(relu?(x + sigmoid(skip) * (o - x)))  (agnn_skip_act_*): the HGT layer's epilogue in one launch each way."""

    @staticmethod
    def forward(ctx, o, x, skip, relu: bool, p: float, call_id: int):
        dev = _lib.require_gpu(o)
        lib = _lib.load()
        o = o if _ok16(o) else o.float().contiguous()
        if x is not None:
            x = x if _ok16(x) else x.float().contiguous()
        n, H = o.shape
        z = torch.empty((n, H), dtype=torch.float32, device=dev)
        rng = rng_state(dev) if p > 0 else None
        used = torch.empty(2, dtype=torch.int64, device=dev) if p > 0 else None
        sk = skip.detach().reshape(-1)[:1].contiguous() if skip is not None else None
        flags = 1 if relu else 0
        _lib.check(lib.agnn_skip_act_fwd_f32(_lib.ptr(x), x.stride(0) if x is not None else 0, o.data_ptr(), o.stride(0), _lib.ptr(sk), n, H,
                                             float(p), flags, _lib.ptr(rng), int(call_id), z.data_ptr(), z.stride(0), _lib.ptr(used),
                                             _lib.stream_ptr(dev)), "agnn_skip_act_fwd_f32")
        ctx.save_for_backward(o, *([x, sk] if x is not None else []), *([used] if used is not None else []))
        ctx.cfg = (bool(relu), float(p), int(call_id), x is not None, tuple(skip.shape) if skip is not None else None)
        ctx.set_materialize_grads(False)          # an undefined output gradient (a structurally dead node type) stays undefined upstream
        return z

    @staticmethod
    def backward(ctx, dz):
        if dz is None:
            return None, None, None, None, None, None
        relu, p, call_id, has_x, skip_shape = ctx.cfg
        saved = list(ctx.saved_tensors)
        o = saved.pop(0)
        x, sk = (saved.pop(0), saved.pop(0)) if has_x else (None, None)
        used = saved.pop(0) if p > 0 else None
        dev = dz.device
        lib = _lib.load()
        dz = dz if _ok16(dz) else dz.float().contiguous()
        n, H = o.shape
        do = torch.empty_like(o)
        want_dx = has_x and ctx.needs_input_grad[1]
        want_ds = has_x and ctx.needs_input_grad[2]
        dx = torch.empty((n, H), dtype=torch.float32, device=dev) if want_dx else None
        ds = torch.empty(1, dtype=torch.float32, device=dev) if want_ds else None
        ws = None
        if want_ds:
            ws = _MIX_WS.get(str(dev))                    # one per device, zero-filled once; every call leaves its ticket at zero
            if ws is None:
                ws = _MIX_WS[str(dev)] = torch.zeros(int(lib.agnn_skip_act_workspace_bytes()) + 256, dtype=torch.uint8, device=dev)
        wsp = ((ws.data_ptr() + 255) & ~255) if ws is not None else None
        _lib.check(lib.agnn_skip_act_bwd_f32(_lib.ptr(x), x.stride(0) if has_x else 0, o.data_ptr(), o.stride(0), _lib.ptr(sk), n, H, p,
                                             1 if relu else 0, _lib.ptr(used), call_id, dz.data_ptr(), dz.stride(0), _lib.ptr(dx),
                                             dx.stride(0) if dx is not None else 0, do.data_ptr(), do.stride(0), _lib.ptr(ds), wsp,
                                             int(lib.agnn_skip_act_workspace_bytes()) if ws is not None else 0, _lib.stream_ptr(dev)),
                   "agnn_skip_act_bwd_f32")
        return do, dx, (ds.reshape(skip_shape) if ds is not None else None), None, None, None


def skip_act(o: torch.Tensor, x: Optional[torch.Tensor], skip: Optional[torch.Tensor], relu: bool, p: float, training: bool) -> torch.Tensor:
    """dropout_p(relu?(lerp(x, o, sigmoid(skip))))  — `x` / `skip` None: no skip connection.  H % 4 == 0 rows on the kernel, anything
    else on torch ops."""
    p = float(p) if training else 0.0
    if not (ENABLED and o.is_cuda and o.dim() == 2 and o.shape[1] % 4 == 0 and o.shape[1] <= 2048 and o.shape[0] > 0 and
            (x is None or tuple(x.shape) == tuple(o.shape))):
        y = o if x is None else torch.lerp(x, o, torch.sigmoid(skip))
        y = F.relu(y) if relu else y
        return F.dropout(y, p, True) if p > 0 else y
    if x is None and not relu and p == 0.0:
        return o
    return _SkipAct.apply(o, x, skip if x is not None else None, relu, p, next(_CALL_IDS) & 0xFFFFFFFF)


def norm_act(x: torch.Tensor, ln: nn.LayerNorm, pre_relu: bool = False, post_relu: bool = False, p: float = 0.0,
             training: bool = False) -> torch.Tensor:
    """dropout_p( relu?( LayerNorm( relu?(x) ) ) ) on the last dimension of x (2-D or 3-D, fp32, H % 4 == 0)."""
    H = x.shape[-1]
    usable = (ENABLED and x.is_cuda and x.dtype == torch.float32 and H % 4 == 0 and H <= 1024 and ln.elementwise_affine
              and tuple(ln.normalized_shape) == (H,))
    if not usable:
        y = F.relu(x) if pre_relu else x
        y = ln(y)
        y = F.relu(y) if post_relu else y
        return F.dropout(y, p, training)
    x2 = x.reshape(-1, H)
    flags = (PRE_RELU if pre_relu else 0) | (POST_RELU if post_relu else 0)
    y = _NormAct.apply(x2, ln.weight, ln.bias, ln.eps, p if training else 0.0, flags, next(_CALL_IDS) & 0xFFFFFFFF)
    return y.view(x.shape)


def grouped_norm_act(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, pre_relu: bool = False, pre=None) -> torch.Tensor:
    """x [N, G, H] -> LayerNorm over H with per-group affine gamma/beta [G, H] (optionally ReLU first): the G task heads'
    `ReLU -> LayerNorm` of one note in ONE pass over the [N, G*H] row (segmented statistics, models/analysis.py:488-493)."""
    N, G, H = x.shape
    W = G * H
    gl = H // 4
    ok = (ENABLED and x.is_cuda and x.dtype == torch.float32 and H % 4 == 0 and 256 % H == 0 and (gl & (gl - 1)) == 0 and W <= 2048)
    if not ok:
        if pre is not None:
            raise _lib.AgnnError("grouped_norm_act: precomputed results need the kernel path")
        y = F.relu(x) if pre_relu else x
        return F.layer_norm(y, (H,), None, None, eps) * gamma + beta
    view_only = all(t.is_leaf or getattr(t, "_agnn_wgrad_deferrable", False) for t in (gamma, beta))
    y = _NormAct.apply(x.reshape(N, W), gamma.reshape(W), beta.reshape(W), eps, 0.0, PRE_RELU if pre_relu else 0,
                       next(_CALL_IDS) & 0xFFFFFFFF, H, view_only, leaf_refs(gamma, beta), pre)
    return y.view(N, G, H)


class FusedSequential(nn.Sequential):
    """`nn.Sequential` whose forward runs `ReLU -> LayerNorm [-> Dropout]`, `LayerNorm [-> ReLU] [-> Dropout]` chains
    through `norm_act`; every other module runs as usual.  Module indices (and so `state_dict` keys) are untouched."""

    def forward(self, x):
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            nn2 = mods[i + 2] if i + 2 < len(mods) else None
            if isinstance(m, nn.ReLU) and isinstance(nxt, nn.LayerNorm):
                p, used = (nn2.p, 3) if isinstance(nn2, nn.Dropout) else (0.0, 2)
                x = norm_act(x, nxt, pre_relu=True, p=p, training=self.training)
                i += used
            elif isinstance(m, nn.LayerNorm):
                post = isinstance(nxt, nn.ReLU)
                d = mods[i + 1 + int(post)] if i + 1 + int(post) < len(mods) else None
                p = d.p if isinstance(d, nn.Dropout) else 0.0
                x = norm_act(x, m, post_relu=post, p=p, training=self.training)
                i += 1 + int(post) + int(isinstance(d, nn.Dropout))
            else:
                x = m(x)
                i += 1
        return x
