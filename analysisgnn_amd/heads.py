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
from .linear import all_steal, defer, deferring, leaf_refs, linear, mark_wgrad_async, wgrad_stream
from .params import cat_rows, stack_rows


def fused_head_logits(clf_dict: nn.ModuleDict, x: torch.Tensor, tasks: Sequence[str]) -> Tuple[torch.Tensor, List[int]]:
    """Concatenated logits [N, sum_t C_t] of the given tasks and the segment offsets (len T+1)."""
    mods = [clf_dict[t] for t in tasks]
    T = len(mods)
    h2 = mods[0][0].out_features
    # cat / stack of leaf parameters: their backward is narrow / unbind (views), so these gradients may arrive late;
    # the cats themselves are views when the parameters are adjacent in memory (dp.plan_parameters), launches otherwise
    W1 = _cat_params([m[0].weight for m in mods])                         # [T*h2, o]; the cat's backward only takes views
    b1 = _cat_params([m[0].bias for m in mods])
    gamma = _cat_params([m[2].weight for m in mods], stack=True)          # [T, h2]; backward = views
    beta = _cat_params([m[2].bias for m in mods], stack=True)
    offs = [0]
    for m in mods:
        offs.append(offs[-1] + m[3].out_features)
    b2 = _cat_params([m[3].bias for m in mods])
    eps = mods[0][2].eps
    if (HEADS_FUSED and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[1] == HEADS_IN and h2 == HEADS_HIDDEN
            and T <= _lib.MAX_SEG and x.shape[0] > 0 and x.shape[0] * max(T * h2, offs[-1]) < 2 ** 30 - 2 ** 20):
        # one launch computes everything (agnn_heads_fwd_f32); the three autograd nodes below are created around its results
        # and only carry the backward pass
        W2 = _cat_params([m[3].weight for m in mods])
        z, y, mean, rstd, logits = heads_forward(x, W1, b1, gamma, beta, eps, W2, b2, offs, h2)
        a = linear(x, W1, b1, pre=[z])
        a = grouped_norm_act(a.view(-1, T, h2), gamma, beta, eps, pre_relu=True, pre=[y, mean, rstd])
        return grouped_projection(a.reshape(-1, T * h2), W2, b2, offs, h2, pre=[logits]), offs
    a = linear(x, W1, b1)                                                 # [N, T*h2]
    a = grouped_norm_act(a.view(-1, T, h2), gamma, beta, eps, pre_relu=True)   # ReLU + per-task LayerNorm, one launch
    a = a.reshape(-1, T * h2)
    if a.is_cuda and h2 in GPROJ_K and T <= _lib.MAX_SEG and GPROJ_ENABLED:
        W2 = _cat_params([m[3].weight for m in mods])                     # [sum C, h2]
        logits = grouped_projection(a, W2, b2, offs, h2)
    else:                                                                 # widths the kernel is not built for
        W2 = torch.block_diag(*[m[3].weight for m in mods])               # [sum C, T*h2]
        logits = linear(a, W2, b2)
    return logits, offs


def _cat_params(ps, stack: bool = False) -> torch.Tensor:
    """cat / stack of LEAF parameters as one operand whose gradient may arrive late (linear.mark_wgrad_async): the
    backward of either only hands out views, and the consumers check the leaves' `.grad` state before they defer."""
    ps = list(ps)
    t = stack_rows(ps) if stack else cat_rows(ps)
    return mark_wgrad_async(t, deferrable=True, leaves=ps) if all(p.is_leaf for p in ps) else t


HEADS_FUSED = False      # one launch for the whole head block (csrc/heads.hip).  Correct, but 148 us against 141 us for the three launches
                         # at C2 (profiles/r03_heads.md: why) — off until it wins; bench.py --fused-heads switches it on for A/B runs
HEADS_IN, HEADS_HIDDEN = 128, 64     # what agnn_heads_fwd_f32 is built for (the reference's out_channels = 128 models)
_OFFS_HOST: dict = {}


def heads_forward(x, W1, b1, gamma, beta, eps, W2, b2, offs, h2):
    """(z, y, mean, rstd, logits) of the whole head block in one launch (csrc/heads.hip); operands are the stacked parameters."""
    dev = _lib.require_gpu(x, W1, W2)
    with torch.no_grad():
        xc = x.detach()
        xc = xc if (xc.stride(1) == 1 and xc.stride(0) % 4 == 0 and xc.data_ptr() % 16 == 0) else xc.contiguous()
        al = lambda t: t if (t.is_contiguous() and t.data_ptr() % 16 == 0) else t.contiguous().clone()     # noqa: E731
        w1, w2 = al(W1.detach()), al(W2.detach())
        c = lambda t: t.detach().reshape(-1).contiguous()                                                  # noqa: E731
        T = len(offs) - 1
        N = xc.shape[0]
        key = tuple(int(o) for o in offs)
        oh = _OFFS_HOST.get(key)
        if oh is None:
            oh = _OFFS_HOST[key] = (_lib.C.c_int32 * len(key))(*key)
        z = torch.empty((N, T * h2), dtype=torch.float32, device=dev)
        y = torch.empty((N, T * h2), dtype=torch.float32, device=dev)
        mean = torch.empty((N * T,), dtype=torch.float32, device=dev)
        rstd = torch.empty((N * T,), dtype=torch.float32, device=dev)
        logits = torch.empty((N, key[-1]), dtype=torch.float32, device=dev)
        b2c = c(b2) if b2 is not None else None
        _lib.check(_lib.load().agnn_heads_fwd_f32(xc.data_ptr(), xc.stride(0), N, xc.shape[1], h2, T, w1.data_ptr(), c(b1).data_ptr(),
                                                  c(gamma).data_ptr(), c(beta).data_ptr(), float(eps), w2.data_ptr(), _lib.ptr(b2c),
                                                  _offs_tensor(offs, dev).data_ptr(), oh, z.data_ptr(), y.data_ptr(), z.stride(0),
                                                  mean.data_ptr(), rstd.data_ptr(), logits.data_ptr(), logits.stride(0),
                                                  _lib.stream_ptr(dev)), "agnn_heads_fwd_f32")
    return z, y, mean, rstd, logits


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
    def forward(ctx, a, w, b, offs_t, offs, K, pre=None):
        dev = _lib.require_gpu(a, w, offs_t)
        ctx._wg_async_in = all(t is None or t.is_leaf or getattr(t, "_agnn_wgrad_async", False) for t in (w, b))
        ctx._wg_defer_in = all(t is None or t.is_leaf or getattr(t, "_agnn_wgrad_deferrable", False) for t in (w, b))
        ctx.steal_refs = leaf_refs(w, b)
        a = _lib.f32c(a)
        w = _lib.f32c(w)
        if w.data_ptr() % 16:
            w = w.clone()
        G = len(offs) - 1
        sum_c = offs[-1]
        tiles = sum((offs[i + 1] - offs[i] + 31) // 32 for i in range(G))
        N = a.shape[0]
        if pre is not None:                      # computed by the fused head kernel: this node only carries the backward pass
            out = pre[0]
        else:
            out = torch.empty((N, sum_c), dtype=torch.float32, device=dev)
            lib = _lib.load()
            bb = b.float().contiguous() if b is not None else None
            _lib.check(lib.agnn_gproj_fwd_f32(a.data_ptr(), a.stride(0), w.data_ptr(), _lib.ptr(bb), offs_t.data_ptr(), G, K, tiles, N,
                                              out.data_ptr(), out.stride(0), _lib.stream_ptr(dev)), "agnn_gproj_fwd_f32")
        ctx.save_for_backward(a, w, offs_t)
        ctx.meta = (G, K, tiles, sum_c, b is not None)
        ctx.wg_async = ctx._wg_async_in
        ctx.wg_defer = ctx._wg_defer_in
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
        if need_w:                                   # optimizer-only outputs: deferred, or on the weight-gradient stream (linear.py)
            dw = torch.empty_like(w)
            db = torch.empty((sum_c,), dtype=torch.float32, device=dev) if has_b else None
            dw_k, db_k = dw.detach(), (db.detach() if db is not None else None)     # aliases: see linear._LinearFn.backward

            def weight_grads():
                nws = int(lib.agnn_gproj_workspace_bytes(N, sum_c, K, tiles))
                ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=dev)
                _lib.check(lib.agnn_gproj_bwd_f32(dout.data_ptr(), dout.stride(0), a.data_ptr(), a.stride(0), w.data_ptr(),
                                                  offs_t.data_ptr(), G, K, tiles, sum_c, N, None, 0, dw_k.data_ptr(), _lib.ptr(db_k),
                                                  ws.data_ptr(), nws, _lib.stream_ptr(dev)), "agnn_gproj_bwd_f32")
            off_chain = ctx.wg_async and all_steal(ctx.steal_refs)      # else: AccumulateGrad adds on THIS stream, right now
            if off_chain and ctx.wg_defer and deferring(dout):
                defer(weight_grads, dev)
            else:
                with wgrad_stream(dev, dout, a, dw_k, db_k, active=off_chain):
                    weight_grads()
        da = None
        if ctx.needs_input_grad[0]:
            da = torch.empty_like(a)
            _lib.check(lib.agnn_gproj_bwd_f32(dout.data_ptr(), dout.stride(0), a.data_ptr(), a.stride(0), w.data_ptr(),
                                              offs_t.data_ptr(), G, K, tiles, sum_c, N, da.data_ptr(), da.stride(0), None, None,
                                              None, 0, _lib.stream_ptr(dev)), "agnn_gproj_bwd_f32")
        return da, dw, db, None, None, None, None


def grouped_projection(a: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], offs: Sequence[int], K: int, pre=None) -> torch.Tensor:
    """out[:, offs[g]:offs[g+1]] = a[:, g*K:(g+1)*K] @ w[offs[g]:offs[g+1]].T + b[offs[g]:offs[g+1]]  for every group g."""
    return _GroupedProj.apply(a, w, b, _offs_tensor(offs, a.device), tuple(int(o) for o in offs), int(K), pre)


class _GroupedInProj(torch.autograd.Function):
    """The mirror image of the grouped projection: group g reads ITS OWN C_g input columns and writes K output columns,
        out[:, g*K:(g+1)*K] = x[:, offs[g]:offs[g+1]] @ w[offs[g]:offs[g+1]]            x [N, sum C], w [sum C, K]
    (the 21 `clf_proj_layers[task][0] = Linear(C_t, o/2)` of the logit-fusion path, models/analysis.py:499-505, :552).
    Same three kernels with the roles swapped: forward = k_gproj_dx, input gradient = k_gproj_fwd, weight gradient =
    k_gproj_dw."""

    @staticmethod
    def forward(ctx, x, w, offs_t, offs, K):
        dev = _lib.require_gpu(x, w, offs_t)
        x = _lib.f32c(x)
        w = _lib.f32c(w)
        if w.data_ptr() % 16:
            w = w.clone()
        G = len(offs) - 1
        sum_c = offs[-1]
        tiles = sum((offs[i + 1] - offs[i] + 31) // 32 for i in range(G))
        N = x.shape[0]
        out = torch.empty((N, G * K), dtype=torch.float32, device=dev)
        lib = _lib.load()
        _lib.check(lib.agnn_gproj_bwd_f32(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), w.data_ptr(), offs_t.data_ptr(), G, K,
                                          tiles, sum_c, N, out.data_ptr(), out.stride(0), None, None, None, 0, _lib.stream_ptr(dev)),
                   "agnn_gproj_bwd_f32(dx as forward)")
        ctx.save_for_backward(x, w, offs_t)
        ctx.meta = (G, K, tiles, sum_c)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w, offs_t = ctx.saved_tensors
        G, K, tiles, sum_c = ctx.meta
        dev = x.device
        dout = _lib.f32c(dout)
        N = x.shape[0]
        lib = _lib.load()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.check(lib.agnn_gproj_fwd_f32(dout.data_ptr(), dout.stride(0), w.data_ptr(), None, offs_t.data_ptr(), G, K, tiles, N,
                                              dx.data_ptr(), dx.stride(0), _lib.stream_ptr(dev)), "agnn_gproj_fwd_f32(as input gradient)")
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            nws = int(lib.agnn_gproj_workspace_bytes(N, sum_c, K, tiles))
            ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=dev)
            _lib.check(lib.agnn_gproj_bwd_f32(x.data_ptr(), x.stride(0), dout.data_ptr(), dout.stride(0), w.data_ptr(), offs_t.data_ptr(),
                                              G, K, tiles, sum_c, N, None, 0, dw.data_ptr(), None, ws.data_ptr(), nws,
                                              _lib.stream_ptr(dev)), "agnn_gproj_bwd_f32(dw)")
        return dx, dw, None, None, None


def grouped_in_projection(x: torch.Tensor, w: torch.Tensor, offs: Sequence[int], K: int) -> torch.Tensor:
    """out[:, g*K:(g+1)*K] = x[:, offs[g]:offs[g+1]] @ w[offs[g]:offs[g+1]] for every group g (w = the groups' [C_g, K]
    matrices stacked along rows)."""
    return _GroupedInProj.apply(x, w, _offs_tensor(offs, x.device), tuple(int(o) for o in offs), int(K))


class CrossTaskTransformer(nn.Module):
    """models/analysis.py:408-418: `LayerNorm(x + MultiheadAttention(x, x, x))` over the T task tokens of every note
    (batch_first, `num_heads` heads, dropout on the attention probabilities).  Same parameters and names
    (`multihead_attn.in_proj_weight / in_proj_bias / out_proj.*`, `norm.*`); the schedule differs: the packed in-projection
    and the out-projection are single [N*T, E] GEMMs on `linear` (weight gradients on the split-N MFMA kernel), the
    T x T attention of all notes and heads is one scaled-dot-product call, and residual + LayerNorm is one fused launch."""

    def __init__(self, proj_dim, num_heads=4, dropout=0.1):
        super().__init__()
        self.multihead_attn = nn.MultiheadAttention(proj_dim, num_heads, dropout=dropout, batch_first=True)
        self.norm = nn.LayerNorm(proj_dim)

    def forward(self, task_projections):
        from .fused import norm_act
        mha = self.multihead_attn
        N, T, E = task_projections.shape
        h = mha.num_heads
        x2 = task_projections.reshape(N * T, E)
        qkv = linear(x2, mha.in_proj_weight, mha.in_proj_bias).view(N, T, 3, h, E // h)
        q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))                      # [N, h, T, E/h]
        att = F.scaled_dot_product_attention(q, k, v, dropout_p=mha.dropout if self.training else 0.0)
        att = att.transpose(1, 2).reshape(N * T, E)
        out = linear(att, mha.out_proj.weight, mha.out_proj.bias, acc=x2)                # residual as the GEMM's beta = 1 epilogue
        return norm_act(out, self.norm).view(N, T, E)


def fused_logit_fusion(proj_layers: nn.ModuleDict, transformer: nn.Module, fusion_layers: nn.ModuleDict, logits: torch.Tensor,
                       offs: Sequence[int], tasks: Sequence[str], training: bool) -> torch.Tensor:
    """The reference's logit-fusion epilogue (models/analysis.py:550-565) on the side-by-side logits [N, sum C]:
      proj_t   = LayerNorm(ReLU(Linear_t(raw_t)))             21 x clf_proj_layers  -> ONE grouped launch + one fused norm
      enhanced = LayerNorm(proj + MHA(proj, proj, proj))      CrossTaskTransformer over the T task tokens of every note
      refined_t = Linear_t(enhanced[:, t])                    21 x fusion_layers    -> ONE grouped launch
    Returns the refined logits in the same [N, sum C] layout.  Parameters stay in the reference's modules
    (`clf_proj_layers.<task>.{0,2}`, `cross_task_transformer.multihead_attn.*`, `.norm`, `fusion_layers.<task>`)."""
    T = len(tasks)
    pm = [proj_layers[t] for t in tasks]
    K = pm[0][0].out_features
    N = logits.shape[0]
    _lib.require_gpu(logits)
    if not (K in GPROJ_K and T <= _lib.MAX_SEG):
        # Widths the grouped kernels are not built for (out_channels // 2 outside {32, 64, 128}; more than MAX_SEG tasks): the
        # same epilogue as per-task launches on the device — the reference's own loop (:552-565).  Correct, not tuned: the
        # tuned shape is the reference's default (out_channels = 128, 21 tasks).
        proj = [pm[i](logits[:, offs[i]:offs[i + 1]]) for i in range(T)]
        enh = transformer(torch.stack(proj, dim=1))
        return torch.cat([fusion_layers[t](enh[:, i]) for i, t in enumerate(tasks)], dim=1)
    Wp = torch.cat([m[0].weight.t() for m in pm], dim=0)                                 # [sum C, K]
    bp = cat_rows([m[0].bias for m in pm])                                                # [T*K]
    a = grouped_in_projection(logits, Wp, offs, K) + bp
    gamma = _cat_params([m[2].weight for m in pm], stack=True)
    beta = _cat_params([m[2].bias for m in pm], stack=True)
    a = grouped_norm_act(a.view(N, T, K), gamma, beta, pm[0][2].eps, pre_relu=True)       # [N, T, K]
    enh = transformer(a)                                                                  # [N, T, K]
    Wf = cat_rows([fusion_layers[t].weight for t in tasks])                               # [sum C, K]
    bf = cat_rows([fusion_layers[t].bias for t in tasks])
    return grouped_projection(enh.reshape(N, T * K), Wf, bf, offs, K)


class _MultiTaskCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, offs_t, eps: float, ignore_index: int, full_cover: bool = False):
        dev = _lib.require_gpu(logits, labels, offs_t)
        if logits.dtype != torch.float32 or logits.stride(1) != 1:
            logits = logits.float().contiguous()
        N = logits.shape[0]
        T = offs_t.numel() - 1
        labels = _check_labels(labels, T, N)
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


def _check_labels(labels: torch.Tensor, T: int, N: int) -> torch.Tensor:
    """The kernels read labels as int64 [T, N]; anything else would be reinterpreted, not converted."""
    if labels.dtype != torch.int64:
        raise _lib.AgnnError(f"labels must be int64 (torch.long), got {labels.dtype}")
    if tuple(labels.shape) != (T, N):
        raise _lib.AgnnError(f"labels must have shape [T={T}, N={N}] (one row per task), got {tuple(labels.shape)}")
    return labels.contiguous()


_UNIT_GRAD: dict = {}
FINAL_GRADIENTS = True    # A/B switch: False = the round-2 flow (unscaled gradient forward, agnn_train_loss_bwd_f32 backward)


def unit_gradient(device) -> torch.Tensor:
    """THE resident scalar 1.0 of a device: `loss.backward(gradient=heads.unit_gradient(dev))` tells the objective's backward —
    by identity, without reading the value — that the incoming gradient is one, so the gradients finished in the forward
    launches are handed on as they are (no launch).  Any other gradient tensor (the fresh ones `loss.backward()` fills, a
    loss scaled for gradient accumulation) takes the general path: three element-wise multiplies.  Never written to."""
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else (torch.cuda.current_device() if dev.type == "cuda" else 0))
    t = _UNIT_GRAD.get(key)
    if t is None:
        t = _UNIT_GRAD[key] = torch.ones((), dtype=torch.float32, device=dev)
    return t


class _TrainLoss(torch.autograd.Function):
    """total = ce_scale * sum_t (w_t CE_t + reg_t) + lam * mean(feat^2) and its gradients w.r.t. the logits, feat and the task
    weights p, all finished by the forward's three launches (agnn_train_loss_final_f32) for an incoming gradient of one; the
    backward hands them on (`unit_gradient`) or multiplies them by the incoming scalar."""

    @staticmethod
    def forward(ctx, logits, labels, offs_t, feat, task_param, eps: float, ignore_index: int, lam: float, ce_scale: float):
        dev = _lib.require_gpu(logits, labels, offs_t, feat)
        if logits.dtype != torch.float32 or logits.stride(1) != 1:
            logits = logits.float().contiguous()
        if feat.dtype != torch.float32 or feat.stride(1) != 1:
            feat = feat.float().contiguous()
        N = logits.shape[0]
        T = offs_t.numel() - 1
        labels = _check_labels(labels, T, N)
        tp = None
        if task_param is not None:
            tp = task_param.detach().to(torch.float32).contiguous()
            if tp.numel() != T:
                raise _lib.AgnnError(f"task_param must have {T} entries, got {tp.numel()}")
        lib = _lib.load()
        ws = _LOSS_WS.get(str(dev))                           # one per device: calls on one device are assumed not to overlap
        if ws is None:                                        # zero-filled once; every call leaves it zero-filled
            ws = _LOSS_WS[str(dev)] = torch.zeros(int(lib.agnn_train_loss_workspace_bytes()) + 256, dtype=torch.uint8, device=dev)
        wsp = (ws.data_ptr() + 255) & ~255
        row_loss = torch.empty((T, N), dtype=torch.float32, device=dev)
        out = torch.empty((4 * T + 1,), dtype=torch.float32, device=dev)      # loss[T] | inv_cnt[T] | total | wscale[T] | dparam[T]
        dlogits = torch.empty_like(logits)
        ctx.final = FINAL_GRADIENTS
        dfeat = None
        if ctx.final:
            dfeat = torch.empty_like(feat) if ctx.needs_input_grad[3] else None
            _lib.check(lib.agnn_train_loss_final_f32(logits.data_ptr(), logits.stride(0), offs_t.data_ptr(), T, labels.data_ptr(), N, float(eps),
                                                     int(ignore_index), feat.data_ptr(), feat.stride(0), feat.shape[1], float(lam),
                                                     _lib.ptr(tp), float(ce_scale), row_loss.data_ptr(), dlogits.data_ptr(), out.data_ptr(),
                                                     out[T:].data_ptr(), out[2 * T:].data_ptr(), out[2 * T + 1:].data_ptr(),
                                                     out[3 * T + 1:].data_ptr(), _lib.ptr(dfeat), dfeat.stride(0) if dfeat is not None else 0,
                                                     wsp, int(lib.agnn_train_loss_workspace_bytes()), _lib.stream_ptr(dev)),
                       "agnn_train_loss_final_f32")
        else:
            _lib.check(lib.agnn_train_loss_f32(logits.data_ptr(), logits.stride(0), offs_t.data_ptr(), T, labels.data_ptr(), N, float(eps),
                                               int(ignore_index), feat.data_ptr(), feat.stride(0), feat.shape[1], float(lam),
                                               _lib.ptr(tp), float(ce_scale), row_loss.data_ptr(), dlogits.data_ptr(), out.data_ptr(),
                                               out[T:].data_ptr(), out[2 * T:].data_ptr(), out[2 * T + 1:].data_ptr(),
                                               out[3 * T + 1:].data_ptr(), wsp, int(lib.agnn_train_loss_workspace_bytes()),
                                               _lib.stream_ptr(dev)), "agnn_train_loss_f32")
        ctx.save_for_backward(dlogits, offs_t, out, feat, *([dfeat] if dfeat is not None else []))
        ctx.lam = float(lam)
        ctx.T = T
        ctx.has_param = task_param is not None
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)                      # no zero-filled gradient for the logging output
        return out[2 * T], out

    @staticmethod
    def backward(ctx, g, _g_parts):
        dlogits, offs_t, out, feat, *rest = ctx.saved_tensors
        dev = dlogits.device
        T = ctx.T
        want_param = ctx.has_param and ctx.needs_input_grad[4]
        if ctx.final:
            dfeat = rest[0] if rest else None
            if dfeat is None and ctx.needs_input_grad[3]:     # (feat did not require a gradient when the forward ran)
                dfeat = (2.0 * ctx.lam / feat.numel()) * feat
            dparam = out[3 * T + 1:4 * T + 1] if want_param else None
            if g.data_ptr() == unit_gradient(dev).data_ptr():           # one, known by identity: the gradients are finished
                return dlogits, None, None, dfeat, dparam, None, None, None, None
            g = g.to(torch.float32)
            return (dlogits * g, None, None, dfeat * g if dfeat is not None else None, dparam * g if dparam is not None else None,
                    None, None, None, None)
        g = g.to(torch.float32).contiguous()
        dl = torch.empty_like(dlogits)
        dfeat = torch.empty_like(feat) if ctx.needs_input_grad[3] else None
        lib = _lib.load()
        _lib.check(lib.agnn_train_loss_bwd_f32(dlogits.data_ptr(), dlogits.stride(0), offs_t.data_ptr(), T, dlogits.shape[0],
                                               dlogits.shape[1], out[2 * T + 1:].data_ptr(), g.data_ptr(), dl.data_ptr(), dl.stride(0),
                                               feat.data_ptr(), feat.stride(0), feat.shape[1], ctx.lam, _lib.ptr(dfeat),
                                               dfeat.stride(0) if dfeat is not None else 0, _lib.stream_ptr(dev)),
                   "agnn_train_loss_bwd_f32")
        dparam = out[3 * T + 1:4 * T + 1] * g if want_param else None
        return dl, None, None, dfeat, dparam, None, None, None, None


class MultiTaskLoss(nn.Module):
    """The reference's task-weighting module (models/chord.py:16-49): one learned uncertainty weight per task,
    `params` initialised to ones (same parameter name, so a reference state_dict / optimizer group maps by name).
    `requires_grad=False` is the plain sum (mt_strategy other than 'wloss', models/analysis.py:904-908).  The weighting itself
    runs inside `agnn_train_loss_f32` (heads.training_loss); `forward` accepts the reference's dict arguments."""

    def __init__(self, tasks: Sequence[str], loss_ft=None, loss_weights=None, requires_grad: bool = True):
        super().__init__()
        self.tasks = list(tasks)
        self.requires_grad = bool(requires_grad)
        if self.requires_grad:
            self.params = nn.Parameter(torch.ones(len(self.tasks)))
        else:
            self.register_buffer("params", torch.ones(len(self.tasks)), persistent=False)

    def weights(self) -> Optional[torch.Tensor]:
        return self.params if self.requires_grad else None

    def forward(self, pred: Dict[str, torch.Tensor], gt: Dict[str, torch.Tensor], label_smoothing: float = 0.1,
                ignore_index: int = -1) -> Dict[str, torch.Tensor]:
        """pred / gt dicts as at models/analysis.py:1034: per-task losses plus "total" (the weighted sum, NOT yet divided
        by the number of tasks — the caller does that, :1036).  Tasks are weighted by their position in `gt`
        (models/chord.py:41-44 enumerates `out.values()`)."""
        tasks = list(gt.keys())
        logits = torch.cat([pred[t] for t in tasks], dim=1)
        offs = [0]
        for t in tasks:
            offs.append(offs[-1] + pred[t].shape[1])
        labels = torch.stack([gt[t] for t in tasks])
        per = multitask_cross_entropy(logits, offs, labels, label_smoothing, ignore_index)
        out = {t: per[i] for i, t in enumerate(tasks)}
        if self.requires_grad:
            p = self.params[:len(tasks)]
            out["total"] = (0.5 / p ** 2 * per + torch.log(1 + p ** 2)).sum()
        else:
            out["total"] = per.sum()
        return out


def training_loss(logits: torch.Tensor, offs: Sequence[int], labels: torch.Tensor, feat: torch.Tensor, lambda_feat: float = 0.1,
                  label_smoothing: float = 0.1, ignore_index: int = -1, task_params: Optional[torch.Tensor] = None,
                  ce_scale: Optional[float] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(total, per_task): the reference's training objective without its optional (continual-learning, edge) terms,
        total = ce_scale * sum_t (w_t CE_t + reg_t) + lambda_feat * feat.pow(2).mean(),     ce_scale = 1 / T by default
    (models/analysis.py:1034-1036 `loss_dict.pop("total") / len(labels_dict)`, :984, :1072); with `task_params` (the `params`
    of `MultiTaskLoss`, --mt_strategy wloss) w_t = 0.5 / p_t^2 and reg_t = log(1 + p_t^2) (models/chord.py:39-49), otherwise
    w_t = 1, reg_t = 0 — in two launches forward and one backward.  per_task [T] are the mean cross entropies per task (for
    logging; not differentiable).  A task whose labels are all `ignore_index` contributes 0 (torch: NaN); a label outside
    [0, C_t) that is not `ignore_index` makes total and gradients NaN (torch: device assert).  Requires segments that cover
    the logits' columns side by side (what fused_head_logits produces)."""
    T = len(offs) - 1
    scale = (1.0 / T) if ce_scale is None else float(ce_scale)
    full = len(offs) > 1 and offs[0] == 0 and offs[-1] == logits.shape[1] and all(offs[i] < offs[i + 1] for i in range(len(offs) - 1))
    if not full:
        per_task = multitask_cross_entropy(logits, offs, labels, label_smoothing, ignore_index)
        if task_params is not None:
            ce = (0.5 / task_params ** 2 * per_task + torch.log(1 + task_params ** 2)).sum()
        else:
            ce = per_task.sum()
        return scale * ce + lambda_feat * feat.pow(2).mean(), per_task.detach()
    total, out = _TrainLoss.apply(logits, labels, _offs_tensor(offs, logits.device), feat, task_params, label_smoothing,
                                  ignore_index, lambda_feat, scale)
    return total, out[:T]
