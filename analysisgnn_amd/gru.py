"""Host side of the persistent GRU kernels (include/agnn.h: agnn_gru_fwd_f32 / agnn_gru_bwd_f32).

Takes the parameters of a plain `torch.nn.GRU(batch_first=True, bidirectional=True)` — the module
the reference's hybrid branch builds (analysisgnn/models/cadence.py:249-251) — so `state_dict`s stay
identical; the input projections and all weight gradients are library GEMMs, the T-step
recurrence (forward and backward) is one kernel launch per layer."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .linear import (WgItem, all_steal, defer, defer_home, deferring, leaf_refs, mark_wgrad_async, weight_grad, weight_grad_batch,
                     wgrad_stream)

KERNEL_HIDDEN = (64, 128)      # hidden sizes per direction the kernels are instantiated for (H = 128 / 256 hybrid models)


YIELD_TO_PROJECTIONS = True      # A/B switch (bench.py --no-yield-gemm)
PROJECTION_EVENTS: list = []


def wait_for_projections(device) -> None:
    """Called by the graph stacks before their second layer: the current stream waits for the sequence branch's inner input
    projections issued so far (a no-op when there is no sequence branch, or it runs on this very stream)."""
    device = torch.device(device)
    if device.type != "cuda":
        return
    idx = device.index if device.index is not None else torch.cuda.current_device()
    mine = [e for e in PROJECTION_EVENTS if e[0] == idx]
    PROJECTION_EVENTS[:] = [e for e in PROJECTION_EVENTS if e[0] != idx]
    for _, ev in mine:
        torch.cuda.current_stream(device).wait_event(ev)


class _GRULayer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, last_in_backward=False, drop_scale=None):
        """`drop_scale` [B, T, 2*H] (0 or 1 / (1 - p) per element): the layer's OUTPUT is y * drop_scale — nn.GRU's inter-layer
        dropout — written by the recurrence kernel itself; the backward kernel applies the same factor to the incoming
        gradient.  (As separate launches the dropout and its backward sat on the recurrence chain in both directions.)"""
        ctx.last_in_backward = bool(last_in_backward)
        # x [B,T,I]; w_ih [2,3H,I]; w_hh [2,3H,H]; b_ih, b_hh [2,3H]
        dev = _lib.require_gpu(x, w_ih, w_hh, b_ih, b_hh)
        B, T, I = x.shape
        Hh = w_hh.shape[2]
        x2 = x.reshape(B * T, I)
        # 1-D bias: the library adds it in the GEMM epilogue (a [1, 6H] operand is first broadcast-copied into the 49 MB result)
        gi = torch.addmm(b_ih.reshape(-1), x2, w_ih.reshape(6 * Hh, I).t())         # [B*T, 2*3H]
        dev_idx = dev.index if dev.index is not None else torch.cuda.current_device()
        if last_in_backward:                     # first layer of a new sequence pass: nobody collected the previous pass's events
            PROJECTION_EVENTS[:] = [e for e in PROJECTION_EVENTS if e[0] != dev_idx]
        elif YIELD_TO_PROJECTIONS:
            # an inner layer's input projection sits between two recurrence kernels on the step's longest chain: whoever runs
            # GEMMs beside it (the GNN stack on the main stream, which has ~0.2 ms of slack) waits for this event instead of
            # halving its speed (wait_for_projections; C2 step 3.153 -> 3.104 ms).  The same on the way back — the main chain's
            # weight-gradient flush waiting for the inner dX projection — cost 0.09 ms and is not done.
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            PROJECTION_EVENTS.append((dev_idx, ev))
        w_hh_c = w_hh.contiguous()
        b_hh_c = b_hh.contiguous()
        y = torch.empty((B, T, 2 * Hh), dtype=torch.float32, device=dev)
        saved = torch.empty((B, T, 2, 4, Hh), dtype=torch.float32, device=dev)
        lib = _lib.load()
        out = y
        if drop_scale is not None:
            if tuple(drop_scale.shape) != (B, T, 2 * Hh) or drop_scale.dtype != torch.float32 or not drop_scale.is_contiguous():
                raise _lib.AgnnError("gru: drop_scale must be a contiguous fp32 [B, T, 2*hidden] tensor")
            out = torch.empty_like(y)
        _lib.check(lib.agnn_gru_fwd_f32(gi.data_ptr(), w_hh_c.data_ptr(), b_hh_c.data_ptr(), B, T, Hh,
                                        y.data_ptr(), saved.data_ptr(), _lib.ptr(drop_scale), out.data_ptr() if drop_scale is not None else None,
                                        _lib.stream_ptr(dev)), "agnn_gru_fwd_f32")
        ctx.save_for_backward(x2, w_ih, w_hh_c, y, saved, *([drop_scale] if drop_scale is not None else []))
        ctx.dims = (B, T, I, Hh)
        ctx.wg_async = all(getattr(t, "_agnn_wgrad_async", False) or t.is_leaf for t in (w_ih, w_hh, b_ih, b_hh))
        ctx.steal_refs = leaf_refs(w_ih, w_hh, b_ih, b_hh)
        return out

    @staticmethod
    def backward(ctx, dy):
        x2, w_ih, w_hh, y, saved, *drop = ctx.saved_tensors
        drop_scale = drop[0] if drop else None
        B, T, I, Hh = ctx.dims
        dev = dy.device
        dy = dy.contiguous()
        dgi = torch.empty((B, T, 2, 3 * Hh), dtype=torch.float32, device=dev)
        dgh = torch.empty((B, T, 2, 3 * Hh), dtype=torch.float32, device=dev)      # (d r_pre, d z_pre, d n_pre * r)
        hp = torch.empty((B, T, 2, Hh), dtype=torch.float32, device=dev)                 # h_{t-1} per direction: left by the walk itself
        lib = _lib.load()
        _lib.check(lib.agnn_gru_bwd_f32(dy.data_ptr(), y.data_ptr(), saved.data_ptr(), w_hh.data_ptr(), B, T, Hh,
                                        dgi.data_ptr(), dgh.data_ptr(), _lib.ptr(drop_scale), hp.data_ptr(), _lib.stream_ptr(dev)), "agnn_gru_bwd_f32")
        dgi2 = dgi.view(B * T, 6 * Hh)
        wf = w_ih.reshape(6 * Hh, I)
        dx = (dgi2 @ wf).view(B, T, I) if ctx.needs_input_grad[0] else None
        # Everything that only feeds the optimizer leaves the recurrence chain (layer l-1's kernel waits for dx alone).
        # Forked AFTER dx is queued: the weight-gradient GEMMs then run beside the next layer's recurrence kernel
        # (64 of 256 CUs) instead of halving the speed of the dx GEMM the chain is waiting for.
        dw_ih = torch.empty((6 * Hh, I), dtype=torch.float32, device=dev)
        db_ih = torch.empty((6 * Hh,), dtype=torch.float32, device=dev)
        dw_hh = torch.empty((2, 3 * Hh, Hh), dtype=torch.float32, device=dev)            # both directions land in the stacked
        db_hh = torch.empty((2, 3 * Hh), dtype=torch.float32, device=dev)                # gradients directly (no torch.stack)

        def weight_grads():
            items = []
            if I % 2 == 0:
                items.append(WgItem(dgi2, x2, True, dw_ih, db_ih))
            else:                               # odd input width: the kernel pads a column, results are copied into place
                dw, db = weight_grad(dgi2, x2, True)
                dw_ih.copy_(dw)
                db_ih.copy_(db)
            dgh2, hp2 = dgh.view(B * T, 6 * Hh), hp.view(B * T, 2 * Hh)
            for d in range(2):
                items.append(WgItem(dgh2[:, d * 3 * Hh:(d + 1) * 3 * Hh], hp2[:, d * Hh:(d + 1) * Hh], True, dw_hh[d], db_hh[d]))
            weight_grad_batch(items)            # the layer's three products in one launch pair

        # off the chain only while every parameter behind the stacked operands takes its gradient over without a kernel
        # (no .grad yet, no hook): otherwise AccumulateGrad adds on this stream at once and must find the values there
        off_chain = ctx.wg_async and all_steal(ctx.steal_refs)

        def forked():
            with wgrad_stream(dev, dgi, dgh, y, x2, hp, dw_ih, db_ih, dw_hh, db_hh, active=off_chain, kind="sequence"):
                weight_grads()
        # The layer whose backward runs last (layer 0) ends the branch: with deferred weight gradients its fork is issued by
        # the branch's join (encoders._ForkInput), AFTER the join's own kernel — the join is then the first-captured dependent
        # of this layer's dX GEMM and is not held back by the replayed graph (profiles/r02_step_timeline.md).
        # Any other layer, with deferred weight gradients on a branch stream: no fork either (a fork inside the chain makes the
        # chain's own continuation a later-captured dependent) — the work goes to the main chain's flush, behind an event.
        if ctx.last_in_backward and off_chain and deferring(dy):
            defer(forked, dev)
        elif not (off_chain and deferring(dy) and defer_home(weight_grads, dev, (dgi, dgh, y, x2, hp, dw_ih, db_ih, dw_hh, db_hh))):
            forked()
        return dx, dw_ih.view(2, 3 * Hh, I), dw_hh, db_ih.view(2, 3 * Hh), db_hh, None, None


def kernel_applicable(rnn: nn.GRU) -> bool:
    return (isinstance(rnn, nn.GRU) and rnn.hidden_size in KERNEL_HIDDEN and rnn.bidirectional and rnn.batch_first
            and rnn.bias and getattr(rnn, "proj_size", 0) == 0)


def gru_forward(rnn: nn.GRU, x: torch.Tensor, training: bool) -> torch.Tensor:
    """y = rnn(x)[0] with zero initial state.  Persistent HIP kernels when hidden_size is 64 or 128 (H = 128 / 256 models);
    any other size runs the library RNN (MIOpen) — still on the GPU (hidden 256 = H 512: W_hh does not fit one CU's registers
    + LDS)."""
    _lib.require_gpu(x)
    if not kernel_applicable(rnn):
        return rnn(x)[0]
    y = x
    stacked = _stack_gru_params(rnn)
    for layer in range(rnn.num_layers):
        w_ih, w_hh, b_ih, b_hh = stacked[4 * layer:4 * layer + 4]
        drop = None
        if layer < rnn.num_layers - 1 and rnn.dropout > 0 and training:
            B, T = y.shape[0], y.shape[1]
            drop = F.dropout(_ones((B, T, 2 * rnn.hidden_size), y.device), rnn.dropout, True)     # 0 or 1 / (1 - p): one launch
        y = _GRULayer.apply(y.contiguous(), w_ih, w_hh, b_ih, b_hh, layer == 0, drop)
    return y


_ONES: dict = {}


def _ones(shape, device) -> torch.Tensor:
    key = (tuple(shape), str(device))
    t = _ONES.get(key)
    if t is None:
        if len(_ONES) >= 8:
            _ONES.pop(next(iter(_ONES)))
        t = _ONES[key] = torch.ones(shape, dtype=torch.float32, device=device)
    return t


class _StackPairs(torch.autograd.Function):
    """stack((forward, reverse)) of every GRU parameter of every layer in ONE `agnn_pack_f32` launch (torch.stack is a
    launch per parameter, 8 per step, on the sequence branch's chain); backward hands out views of the incoming
    gradients (no kernel), so the weight gradients may be produced late (linear.mark_wgrad_async)."""

    @staticmethod
    def forward(ctx, *params):
        from .params import pack
        dev = params[0].device
        outs, items = [], []
        for i in range(0, len(params), 2):
            a, b = params[i], params[i + 1]
            o = torch.empty((2,) + tuple(a.shape), dtype=torch.float32, device=dev)
            outs.append(o)
            for k, src in enumerate((a, b)):
                src = src.detach()
                items.append((o[k].reshape(1, -1) if src.dim() == 1 else o[k], [src.reshape(1, -1) if src.dim() == 1 else src]))
        pack(items, dev)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        out = []
        for g in grads:
            out.extend(g.unbind(0))
        return tuple(out)


def _stack_gru_params(rnn: nn.GRU):
    names = ("weight_ih", "weight_hh", "bias_ih", "bias_hh")
    flat = []
    for layer in range(rnn.num_layers):
        for n in names:
            flat.append(getattr(rnn, f"{n}_l{layer}"))
            flat.append(getattr(rnn, f"{n}_l{layer}_reverse"))
    from .params import adjacent, stack_rows
    if all(adjacent(flat[i:i + 2]) for i in range(0, len(flat), 2)):      # laid out by dp.plan_parameters: views, no launch
        outs = tuple(stack_rows(flat[i:i + 2]) for i in range(0, len(flat), 2))
        leaves = all(t.is_leaf for t in flat)
    elif flat[0].is_cuda and all(t.is_contiguous() and t.dtype == torch.float32 for t in flat):
        outs = _StackPairs.apply(*flat)
        leaves = all(t.is_leaf for t in flat)
    else:
        outs = tuple(torch.stack((flat[i], flat[i + 1])) for i in range(0, len(flat), 2))
        leaves = all(t.is_leaf for t in flat)
    if leaves:
        for i, o in enumerate(outs):
            mark_wgrad_async(o, leaves=flat[2 * i:2 * i + 2])
    return outs
