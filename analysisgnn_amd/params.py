"""Per-relation parameters presented as ONE GEMM operand, and its gradient handed back per parameter, with one
`agnn_pack_f32` launch per direction (instead of a cat, six adds and nine copies per fused HeteroConv layer).

PyG `HeteroConv({et: SAGEConv})` (ref: analysisgnn/models/cadence.py:147-159,174) sums, per destination type,
`lin_l_r(mean_r) + lin_r_r(x_dst)` over the relations r.  With the per-relation means side by side in A [N, R*in]:
    sum_r lin_l_r(mean_r) = A @ cat_r(W_l_r, dim=1)^T + sum_r b_r ,      sum_r lin_r_r(x) = x @ (sum_r W_r_r)^T
The parameters stay separate `nn.Parameter`s (state_dict unchanged); this module only builds the operands."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch

from . import _lib
from .linear import all_steal, defer, deferring, mark_wgrad_async, wgrad_stream


def pack(items: Sequence[Tuple[torch.Tensor, Sequence[torch.Tensor]]], device) -> None:
    """items: (dst 2-D view, [src 2-D views of the same shape and row stride]);  dst = sum(srcs); no sources: dst = 0."""
    arr = (_lib.PackItem * len(items))()
    for i, (dst, srcs) in enumerate(items):
        if len(srcs) > _lib.PACK_MAX_SRC:
            raise _lib.AgnnError(f"pack: {len(srcs)} sources (max {_lib.PACK_MAX_SRC})")
        rows, cols = dst.shape
        ld_src = (srcs[0].stride(0) if rows > 1 else max(cols, 1)) if srcs else max(cols, 1)
        for t in (dst, *srcs):
            if t.dtype != torch.float32 or t.dim() != 2 or (cols > 1 and t.stride(1) != 1) or tuple(t.shape) != (rows, cols):
                raise _lib.AgnnError("pack: fp32 2-D pieces of one shape with unit inner stride expected")
        for k, t in enumerate(srcs):
            if rows > 1 and t.stride(0) != ld_src:
                raise _lib.AgnnError("pack: the sources of an item must share their row stride")
            arr[i].src[k] = t.data_ptr()
        arr[i].dst = dst.data_ptr()
        arr[i].ld_dst = dst.stride(0) if rows > 1 else max(cols, 1)
        arr[i].ld_src = ld_src
        arr[i].rows, arr[i].cols, arr[i].n_src = rows, cols, len(srcs)
    lib = _lib.load()
    _lib.check(lib.agnn_pack_f32(len(items), arr, _lib.stream_ptr(device)), "agnn_pack_f32")


class _SageOperands(torch.autograd.Function):
    """(W_l [out, R*in], b [out], W_r [out, in]) from R x (lin_l.weight, lin_l.bias, lin_r.weight)."""

    @staticmethod
    def forward(ctx, R: int, *params):
        w_l, b_l, w_r = params[:R], params[R:2 * R], params[2 * R:]
        dev = _lib.require_gpu(*params)
        out_f, in_f = w_l[0].shape
        W_l = torch.empty((out_f, R * in_f), dtype=torch.float32, device=dev)
        b = torch.empty((out_f,), dtype=torch.float32, device=dev)
        W_r = torch.empty((out_f, in_f), dtype=torch.float32, device=dev)
        items = [(W_l[:, r * in_f:(r + 1) * in_f], [w_l[r].detach()]) for r in range(R)]
        items.append((b.view(1, -1), [t.detach().view(1, -1) for t in b_l]))
        items.append((W_r, [t.detach() for t in w_r]))
        pack(items, dev)
        ctx.R = R
        ctx.shape = (out_f, in_f)
        ctx.leaves = all(t.is_leaf for t in params)
        ctx.param_refs = tuple(params)
        return W_l, b, W_r

    @staticmethod
    def backward(ctx, dW_l, db, dW_r):
        R = ctx.R
        out_f, in_f = ctx.shape
        dev = dW_l.device
        G_l = torch.empty((R, out_f, in_f), dtype=torch.float32, device=dev)
        G_b = torch.empty((R, out_f), dtype=torch.float32, device=dev)
        G_r = torch.empty((R, out_f, in_f), dtype=torch.float32, device=dev)
        def fan_out():
            dW_l_, db_, dW_r_ = _lib.f32c(dW_l), db.contiguous(), _lib.f32c(dW_r)
            items = []
            for r in range(R):
                items.append((G_l[r], [dW_l_[:, r * in_f:(r + 1) * in_f]]))
                items.append((G_b[r].view(1, -1), [db_.view(1, -1)]))
                items.append((G_r[r], [dW_r_]))
            pack(items, dev)
        # the incoming gradients may be deferred (linear.defer_weight_grads: they do not exist yet — fan them out behind
        # them) or produced on the weight-gradient stream (fan them out there as well)
        # ... and only while the parameters take these gradients over without a kernel (linear.all_steal); the producers of
        # dW_l / db / dW_r looked at the same parameters, so both sides take the same decision
        off_chain = ctx.leaves and all_steal(ctx.param_refs)
        if off_chain and deferring(dW_l):
            defer(fan_out, dev)
        else:
            with wgrad_stream(dev, dW_l, db, dW_r, active=off_chain):
                fan_out()
        # unbind: one contiguous tensor per parameter (distinct memory, so each .grad can be taken over as is)
        return (None, *G_l.unbind(0), *G_b.unbind(0), *G_r.unbind(0))


class _SageOperandsCat(torch.autograd.Function):
    """(W [out, (R+1)*in] = [W_l_1 .. W_l_R | sum_r W_r_r], b [out]): the operand of the ONE GEMM of a SAGE layer whose root
    operand rides along with the aggregation (ops.AggSpec.root).  `with_l=False`: only (sum_r W_r_r, b) — the layer that
    keeps no edge uses nothing else (lin_l.weight then gets a zero gradient from one fill)."""

    @staticmethod
    def forward(ctx, R: int, with_l: bool, *params):
        w_l, b_l, w_r = params[:R], params[R:2 * R], params[2 * R:]
        dev = _lib.require_gpu(*params)
        out_f, in_f = w_r[0].shape
        nl = R if with_l else 0
        W = torch.empty((out_f, (nl + 1) * in_f), dtype=torch.float32, device=dev)
        b = torch.empty((out_f,), dtype=torch.float32, device=dev)
        items = [(W[:, r * in_f:(r + 1) * in_f], [w_l[r].detach()]) for r in range(nl)]
        items.append((b.view(1, -1), [t.detach().view(1, -1) for t in b_l]))
        items.append((W[:, nl * in_f:], [t.detach() for t in w_r]))
        pack(items, dev)
        ctx.R, ctx.nl = R, nl
        ctx.shape = (out_f, in_f)
        ctx.leaves = all(t.is_leaf for t in params)
        ctx.param_refs = tuple(params)
        return W, b

    @staticmethod
    def backward(ctx, dW, db):
        R, nl = ctx.R, ctx.nl
        out_f, in_f = ctx.shape
        dev = dW.device
        G_l = torch.empty((nl, out_f, in_f), dtype=torch.float32, device=dev)
        G_b = torch.empty((R, out_f), dtype=torch.float32, device=dev)
        G_r = torch.empty((R, out_f, in_f), dtype=torch.float32, device=dev)
        if not nl:    # unused this step, yet still a parameter of the step: a zero gradient (as the reference's zeros @ W gives), not None
            G_l = torch.zeros((R, out_f, in_f), dtype=torch.float32, device=dev)

        def fan_out():
            dW_, db_ = _lib.f32c(dW), db.contiguous()
            items = []
            for r in range(R):
                if r < nl:
                    items.append((G_l[r], [dW_[:, r * in_f:(r + 1) * in_f]]))
                items.append((G_b[r].view(1, -1), [db_.view(1, -1)]))
                items.append((G_r[r], [dW_[:, nl * in_f:]]))
            pack(items, dev)
        off_chain = ctx.leaves and all_steal(ctx.param_refs)
        if off_chain and deferring(dW):      # the incoming gradients are deferred too: fan them out behind them (see above)
            defer(fan_out, dev)
        else:
            with wgrad_stream(dev, dW, db, active=off_chain):
                fan_out()
        return (None, None, *G_l.unbind(0), *G_b.unbind(0), *G_r.unbind(0))


def sage_operands_cat(w_l: List[torch.Tensor], b_l: List[torch.Tensor], w_r: List[torch.Tensor], with_l: bool = True):
    """([W_l_1 .. W_l_R | sum W_r], sum b) — or (sum W_r, sum b) with `with_l=False` — in one launch; None when the
    parameters do not fit the pack kernel (the caller then takes `sage_operands`)."""
    R = len(w_l)
    ok = (w_l[0].is_cuda and R <= _lib.PACK_MAX_SRC and all(t is not None for t in b_l)
          and all(t.shape == w_l[0].shape and t.is_contiguous() for t in (*w_l, *w_r)))
    if not ok:
        return None
    ops_ = _SageOperandsCat.apply(R, with_l, *w_l, *b_l, *w_r)
    if all(t.is_leaf for t in (*w_l, *b_l, *w_r)):
        for t in ops_:
            mark_wgrad_async(t, deferrable=True, leaves=(*w_l, *b_l, *w_r))   # the fan-out defers itself behind a deferred gradient
    return ops_


def sage_operands(w_l: List[torch.Tensor], b_l: List[torch.Tensor], w_r: List[torch.Tensor]):
    R = len(w_l)
    ok = (w_l[0].is_cuda and R <= _lib.PACK_MAX_SRC and all(t is not None for t in b_l)
          and all(t.shape == w_l[0].shape and t.is_contiguous() for t in (*w_l, *w_r)))
    if not ok:
        return torch.cat(w_l, dim=1), (sum(b_l) if all(t is not None for t in b_l) else None), sum(w_r)
    ops_ = _SageOperands.apply(R, *w_l, *b_l, *w_r)
    if all(t.is_leaf for t in (*w_l, *b_l, *w_r)):
        for t in ops_:
            mark_wgrad_async(t, deferrable=True, leaves=(*w_l, *b_l, *w_r))
    return ops_


# ------------------------------------------------------------------------------------------------------------
# Parameters that sit side by side in memory (dp.FlatAdamW lays the model out in the order given by
# dp.plan_parameters) are presented as ONE operand by a view — no launch at all; anywhere else `torch.cat`.
# ------------------------------------------------------------------------------------------------------------
def adjacent(ts: Sequence[torch.Tensor]) -> bool:
    """True when the tensors are contiguous pieces of one storage, one right behind the other."""
    t0 = ts[0]
    if not all(t.is_contiguous() and t.dtype == t0.dtype and t.device == t0.device for t in ts):
        return False
    base = t0.untyped_storage().data_ptr()
    off = t0.storage_offset()
    for t in ts:
        if t.untyped_storage().data_ptr() != base or t.storage_offset() != off:
            return False
        off += t.numel()
    return True


class _AdjacentCat(torch.autograd.Function):
    """cat(params, dim=0) of adjacent parameters as a view of their common storage; backward = row slices (views)."""

    @staticmethod
    def forward(ctx, *params):
        p0 = params[0]
        rows = [int(p.shape[0]) for p in params]
        ctx.rows = rows
        return p0.detach().as_strided((sum(rows),) + tuple(p0.shape[1:]), p0.stride(), p0.storage_offset())

    @staticmethod
    def backward(ctx, g):
        return tuple(g.split(ctx.rows, dim=0))


def cat_rows(params: Sequence[torch.Tensor]) -> torch.Tensor:
    """torch.cat(params, dim=0); free when the parameters are adjacent in memory."""
    if len(params) > 1 and all(p.shape[1:] == params[0].shape[1:] for p in params) and adjacent(params):
        return _AdjacentCat.apply(*params)
    return torch.cat(list(params), dim=0)


def stack_rows(params: Sequence[torch.Tensor]) -> torch.Tensor:
    """torch.stack(params); free when the parameters are adjacent in memory."""
    if len(params) > 1 and all(p.shape == params[0].shape for p in params) and adjacent(params):
        return _AdjacentCat.apply(*[p.unsqueeze(0) for p in params])
    return torch.stack(list(params))
