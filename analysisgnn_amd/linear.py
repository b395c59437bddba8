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


# ------------------------------------------------------------------------------------------------------------
# Weight gradients off the back-propagation chain.  dW / db of a projection are needed only by the optimizer, yet in
# stream order they sit between a layer's dX and the next layer's backward.  With `enable_wgrad_overlap()` they are
# issued on a second HIP stream (forked where dY is ready) and joined once, before the gradients are gathered
# (`join_wgrad()`, called by dp.FlatGradBuffer.pack).  Only for weights whose gradient is consumed WITHOUT a kernel on
# the launching stream: leaf parameters whose .grad is None when backward starts (FlatGradBuffer(views=False)) and
# operands marked by `mark_wgrad_async` (built by kernel-free views or by ops that handle the stream themselves).
# ------------------------------------------------------------------------------------------------------------
_WG = {"enabled": False, "scope": "all", "streams": {}, "dirty": set()}


def enable_wgrad_overlap(flag: bool = True, scope="all") -> None:
    """scope "all": every projection; otherwise the kinds of work that go to the weight-gradient stream, one name or several
    ("sequence": the GRU layers' weight gradients — the recurrence chain; "embed": the embedding tables' gradient, the last node
    of the backward pass, which then runs beside the input layers' deferred weight gradients instead of in front of them)."""
    _WG["enabled"] = bool(flag)
    _WG["scope"] = scope if scope == "all" else frozenset([scope] if isinstance(scope, str) else scope)


def wgrad_overlap_enabled() -> bool:
    return _WG["enabled"]


def mark_wgrad_async(t: torch.Tensor, deferrable: bool = False, leaves=()) -> torch.Tensor:
    """Declare that the gradient of this (non-leaf) operand is consumed without kernels on the launching stream.
    `deferrable`: it may also be produced LATER than the backward pass reaches its consumer (defer_weight_grads) — true when
    the consumer only takes views of it, or defers its own kernels behind it.
    `leaves`: the leaf parameters that (views of) this operand's gradient end up in.  "Without kernels" only holds while
    every one of them has no `.grad` yet and no hook (`_steals`): with gradient accumulation, `zero_grad(set_to_none=False)`
    or `FlatGradBuffer(views=True)` AccumulateGrad ADDS on the main stream at once — before a deferred closure or the
    weight-gradient stream has produced the addend.  The consumers check the leaves at backward time (`all_steal`)."""
    t._agnn_wgrad_async = True
    t._agnn_wgrad_deferrable = bool(deferrable)
    t._agnn_leaves = tuple(leaves)
    return t


def _deferrable(t: Optional[torch.Tensor]) -> bool:
    return t is None or t.is_leaf or getattr(t, "_agnn_wgrad_deferrable", False)


def _async_ok(t: Optional[torch.Tensor]) -> bool:
    return t is None or t.is_leaf or getattr(t, "_agnn_wgrad_async", False)


def _steals(p: Optional[torch.Tensor]) -> bool:
    """At backward time: will autograd take this operand's gradient WITHOUT launching a kernel on the node's (main)
    stream?  A leaf's AccumulateGrad steals a fresh contiguous tensor only when .grad is None and nothing hooks it;
    otherwise it adds / clones on the main stream, which knows nothing about the weight-gradient stream.  A non-leaf
    operand (a cat / stack / pack of parameters, `mark_wgrad_async`) is as good as the leaves behind it."""
    if p is None:
        return True
    if not p.is_leaf:
        return all(_steals(q) for q in getattr(p, "_agnn_leaves", ()))
    return p.grad is None and not p._backward_hooks and not getattr(p, "_post_accumulate_grad_hooks", None)


def leaf_refs(*ts) -> tuple:
    """Collected in an op's FORWARD (where the operands still are the caller's Python objects): the leaf parameters whose
    AccumulateGrad nodes will receive (views of) these operands' gradients."""
    out = []
    for t in ts:
        if t is None:
            continue
        out.extend((t,) if t.is_leaf else getattr(t, "_agnn_leaves", ()))
    return tuple(out)


def all_steal(refs) -> bool:
    """Checked in an op's BACKWARD, right before it decides to produce a gradient late or on another stream.  (A parameter
    used TWICE in one graph cannot be seen from here — autograd's input buffer then adds the two gradients on the main
    stream; such models must not enable the overlap / deferral.)
    A parameter that already HAS a gradient may have got it from an earlier backward pass of the same accumulation window,
    whose late work (deferred closures, the weight-gradient stream) has not been joined yet: AccumulateGrad would add onto a
    tensor that is still being written.  So the first "no" of a backward pass joins everything outstanding — closures run,
    the current stream waits for the side streams; no host synchronisation."""
    if all(_steals(p) for p in refs):
        return True
    if _DEFER["pending"] or _WG["dirty"] or _WG.get("join"):
        join_wgrad()
    return False


class wgrad_stream:
    """`with wgrad_stream(dev, *inputs):` runs the body on the weight-gradient stream (after everything queued so far on
    the current stream) and keeps `inputs` alive for it; a no-op context when the overlap is disabled."""

    def __init__(self, dev, *inputs, active: bool = True, kind: str = "linear"):
        self.on = bool(_WG["enabled"] and active and torch.device(dev).type == "cuda" and (_WG["scope"] == "all" or kind in _WG["scope"]))
        self.dev, self.inputs = torch.device(dev), inputs

    def __enter__(self):
        if not self.on:
            return self
        idx = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
        ws = _WG["streams"].get(idx)
        if ws is None:
            ws = _WG["streams"][idx] = torch.cuda.Stream(device=self.dev)
        cur = torch.cuda.current_stream(self.dev)
        if cur.cuda_stream != ws.cuda_stream:           # already on it (nested use): a stream must not wait for itself in a capture
            ws.wait_stream(cur)
        self.ws = ws
        self.ctx = torch.cuda.stream(ws)
        self.ctx.__enter__()
        _WG["dirty"].add(idx)
        return self

    def __exit__(self, *exc):
        if self.on:
            self.ctx.__exit__(*exc)
            for t in self.inputs:
                if t is not None:
                    t.record_stream(self.ws)
        return False


# ------------------------------------------------------------------------------------------------------------
# Deferred weight gradients.  On the stream that carries the backward pass, dW / db of a projection sit between its dX
# and the next layer's backward although only the optimizer reads them; in the hybrid encoders that stream later IDLES
# (~0.7 ms at C2) until the sequence branch's backward has finished (profiles/r02_step_timeline.md).  With
# `defer_weight_grads(True)` a projection's backward leaves a closure instead of launching dW / db; the closures run, in
# order, on the same stream at `flush_deferred()` — the hybrid encoders call it where the GNN stack's backward ends, i.e.
# at the start of that idle window — and, for whatever is still pending, in `join_wgrad()` before the gradients are
# gathered.  No second stream, no fork: inside a replayed hipGraph a fork delays whichever chain is captured second.
# Only for gradients that autograd takes over without a kernel (`_steals`).  Closures are kept per stream and run on the
# stream that deferred them (their operands were produced there).
# ------------------------------------------------------------------------------------------------------------
_DEFER = {"on": False, "pending": {}}        # pending: (device index, stream handle) -> (stream, [closures])


def defer_weight_grads(flag: bool = True) -> None:
    _DEFER["on"] = bool(flag)
    if not flag:
        flush_all_deferred()


def deferring(t: torch.Tensor) -> bool:
    return bool(_DEFER["on"] and t.is_cuda)


def defer(fn, dev, here: bool = False) -> None:
    """Queue optimizer-only work on the CURRENT stream's list: it runs at that stream's flush.  (Sending a branch stream's
    projections to the main chain's flush instead — `defer_home` — measured slower, 3.46 vs 3.39 ms at C2; only the inner
    recurrent layers' weight gradients go that way, because they must run beside the next layer's recurrence.)"""
    s = torch.cuda.current_stream(dev)
    home = _DEFER.get("home")
    if ITEMS_HOME and not here and isinstance(fn, (WgItem, ColsumItem)) and home is not None and (home.device.index, home.cuda_stream) != (s.device.index, s.cuda_stream):
        # a branch stream's batchable product joins the main chain's next flush (one launch with that chain's own products)
        fn.event = torch.cuda.Event()
        fn.event.record(s)
        defer_on(home, fn)
        return
    _DEFER["pending"].setdefault((s.device.index, s.cuda_stream), (s, []))[1].append(fn)


ITEMS_HOME = True        # A/B switch (bench.py --no-items-home): 3.22 -> 3.19 ms at C2 (two alternating pairs on one box)


def defer_on(stream, fn) -> None:
    """Queue `fn` for the flush of ANOTHER stream (`fn` must order itself behind its operands, e.g. wait for an event)."""
    _DEFER["pending"].setdefault((stream.device.index, stream.cuda_stream), (stream, []))[1].append(fn)


def set_home_stream(stream) -> None:
    """The stream that carries the backward pass's main chain: where work handed over with `defer_home` is run."""
    _DEFER["home"] = stream


def defer_home(fn, dev, tensors=()) -> bool:
    """From a branch stream: hand optimizer-only work to the main chain's flush (it runs there behind an event recorded
    now on the current stream).  False when there is no home stream, or the current stream is it.  `tensors`: what `fn`
    reads and writes — allocated on the branch's stream, used on the home stream: recorded there, so that the allocator does
    not hand their blocks back to the branch's pool while the home stream's kernels are only queued."""
    home = _DEFER.get("home")
    cur = torch.cuda.current_stream(dev)
    if home is None or (home.device.index, home.cuda_stream) == (cur.device.index, cur.cuda_stream):
        return False
    ev = torch.cuda.Event()
    ev.record(cur)

    def run():
        here = torch.cuda.current_stream(dev)
        here.wait_event(ev)
        fn()
        for t in tensors:
            if t is not None:
                t.record_stream(here)
    run.handed_over = True               # (flush_deferred's spill must not send it back to the stream it waits for)
    defer_on(home, run)
    return True


def _run_deferred(fns) -> None:
    """The weight-gradient products of the list first, TOGETHER (`weight_grad_batch`: one launch pair per 16 of them), then the
    other closures in their order — those only consume weight gradients (the SAGE layers' per-relation fan-out), never feed one."""
    items = [f for f in fns if isinstance(f, WgItem)]
    sums = [f for f in fns if isinstance(f, ColsumItem)]
    foreign = [f for f in items + sums if f.event is not None]
    if foreign:                          # handed over from another stream: wait for their operands, keep those alive for this stream
        here = torch.cuda.current_stream(foreign[0].tensors()[0].device)
        for f in foreign:
            here.wait_event(f.event)
            for t in f.tensors():
                if t is not None:
                    t.record_stream(here)
    with torch.no_grad():                # a flush may run outside a backward pass (FlatGradBuffer.pack): gradient math, never recorded
        if items:
            weight_grad_batch(items)
        if sums:
            colsum_batch(sums)
        for fn in fns:
            if not isinstance(fn, (WgItem, ColsumItem)):
                fn()


FLUSH_KEEP = 1.0         # A/B switch (bench.py --flush-keep): share (by FLOPs) of a flush point's products that run there


def _spill(fns, keep: float, other) -> list:
    """Hand the LAST weight-gradient products of the list — each with the closures that follow it: its consumers — to `other`'s
    flush until only `keep` of the list's FLOPs remain; returns what stays.  The operands are complete on the current stream now:
    an event recorded here orders `other` behind them.  Work that was itself handed over from another stream (`event` set, or a
    `defer_home` wrapper) stays where it was sent: `other` may be the stream it waits for."""
    cost = lambda f: 2.0 * f.dy.shape[0] * f.dy.shape[1] * f.x.shape[1]                 # noqa: E731
    units, cur = [], []                  # [leading closures], [product, its closures ...], ...
    for f in fns:
        if isinstance(f, WgItem):
            units.append(cur)
            cur = [f]
        else:
            cur.append(f)
    units.append(cur)
    movable = lambda u: (u and isinstance(u[0], WgItem) and u[0].event is None                                  # noqa: E731
                         and not any(getattr(f, "event", None) is not None or getattr(f, "handed_over", False) for f in u[1:]))
    total = sum(cost(u[0]) for u in units if u and isinstance(u[0], WgItem))
    if total <= 0:
        return fns
    moved, out = 0.0, set()
    for i in range(len(units) - 1, 0, -1):
        u = units[i]
        if not movable(u):
            continue
        if moved + cost(u[0]) > (1.0 - keep) * total + 1e-9:
            break
        moved += cost(u[0])
        out.add(i)
    if not out:
        return fns
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(other.device))
    for i in sorted(out):
        for f in units[i]:
            if isinstance(f, (WgItem, ColsumItem)):
                f.event = ev
            defer_on(other, f)
    return [f for i, u in enumerate(units) if i not in out for f in u]


def flush_deferred(dev=None, spill_to=None) -> None:
    """Run, on the current stream, the closures that were deferred on it.  `spill_to`: another stream whose own flush comes
    later and has room (the sequence branch's, which idles once the recurrence is through): with FLUSH_KEEP < 1 the last
    products of the list go there."""
    s = torch.cuda.current_stream(dev)
    entry = _DEFER["pending"].pop((s.device.index, s.cuda_stream), None)
    if entry is not None:
        fns = entry[1]
        if spill_to is not None and FLUSH_KEEP < 1.0:
            fns = _spill(fns, FLUSH_KEEP, spill_to)
        _run_deferred(fns)
        join_later(s)                    # whoever gathers the gradients waits for this stream (a no-op for its own)


def flush_all_deferred() -> None:
    """Every stream's pending closures, each list on its own stream; the current stream then waits for the others."""
    cur = torch.cuda.current_stream() if torch.cuda.is_available() else None
    for key in list(_DEFER["pending"]):
        s, fns = _DEFER["pending"].pop(key)
        with torch.cuda.stream(s):
            _run_deferred(fns)
        if cur is not None and (s.device.index, s.cuda_stream) != (cur.device.index, cur.cuda_stream):
            torch.cuda.current_stream(s.device).wait_stream(s)


def join_later(stream) -> None:
    """`stream` has been given optimizer-only work that nothing waits for yet: `join_wgrad` will."""
    _WG.setdefault("join", {})[(stream.device.index, stream.cuda_stream)] = stream


def join_wgrad() -> None:
    """Make the current stream wait for all weight-gradient work issued so far (no host sync)."""
    flush_all_deferred()
    for idx in list(_WG["dirty"]):
        torch.cuda.current_stream(idx).wait_stream(_WG["streams"][idx])
    _WG["dirty"].clear()
    for s in list(_WG.get("join", {}).values()):
        cur = torch.cuda.current_stream(s.device)
        if (cur.device.index, cur.cuda_stream) != (s.device.index, s.cuda_stream):
            cur.wait_stream(s)
    _WG.get("join", {}).clear()


def _ok(t: torch.Tensor) -> bool:
    return (t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 2 == 0
            and t.shape[1] % 2 == 0 and t.data_ptr() % 8 == 0)


def weight_grad(dy: torch.Tensor, x: torch.Tensor, want_bias: bool, dw_out: Optional[torch.Tensor] = None,
                db_out: Optional[torch.Tensor] = None):
    """(dW [out, in], db [out] or None) for dy [N, out], x [N, in] on the HIP kernel; library GEMM when the
    shape / alignment does not fit the kernel.  An odd `in` is served when x has a spare (zero) column behind its last
    one — rows padded to an even stride, as models.encode lays out the 153-wide note input: the kernel then runs on
    in + 1 columns and the extra gradient column is dropped.  `dw_out` [out, in] / `db_out` [out] (contiguous, even `in`):
    write the results there (a slot of a stacked gradient) instead of into fresh tensors."""
    n, out_f = dy.shape
    in_f = x.shape[1]
    in_k = in_f
    if in_f & 1 and x.dim() == 2 and x.stride(1) == 1 and x.stride(0) > in_f:
        in_k = in_f + 1
        x = x.as_strided((n, in_k), (x.stride(0), 1), x.storage_offset())
    if not (ENABLED and dy.is_cuda and n >= MIN_ROWS and out_f * in_k <= MAX_OUT_IN and _ok(dy) and _ok(x)):
        if in_k != in_f:
            x = x[:, :in_f]
        if dw_out is not None:
            torch.mm(dy.t(), x, out=dw_out)
            if want_bias:
                torch.sum(dy, dim=0, out=db_out)
            return dw_out, (db_out if want_bias else None)
        return dy.t() @ x, (dy.sum(dim=0) if want_bias else None)
    lib = _lib.load()
    dev = dy.device
    if dw_out is not None:
        if in_k != in_f or tuple(dw_out.shape) != (out_f, in_f) or not dw_out.is_contiguous() or (want_bias and (db_out is None or not db_out.is_contiguous())):
            raise _lib.AgnnError("weight_grad: dw_out / db_out must be contiguous [out, in] / [out] with an even `in`")
        dw, db = dw_out, (db_out if want_bias else None)
    else:
        dw = torch.empty((out_f, in_k), dtype=torch.float32, device=dev)
        db = torch.empty((out_f,), dtype=torch.float32, device=dev) if want_bias else None
    nws = int(lib.agnn_wgrad_workspace_bytes(n, out_f, in_k))
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    _lib.check(lib.agnn_wgrad_f32(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), n, out_f, in_k, dw.data_ptr(),
                                  dw.stride(0), _lib.ptr(db), ws.data_ptr(), nws, _lib.stream_ptr(dev)), "agnn_wgrad_f32")
    # a column-sliced view would break the layout contract of the (contiguous) parameter: AccumulateGrad would clone it on
    # the MAIN stream without waiting for this (possibly side) stream — hand out a contiguous tensor made right here
    return (dw if in_k == in_f else dw[:, :in_f].contiguous()), db


class WgItem:
    """One pending weight-gradient product dW = dY^T X (+ db) with its destinations: what a projection's backward leaves behind
    under dp.defer_weight_grads instead of a closure, so that the flush can issue all of them in one launch pair."""
    __slots__ = ("dy", "x", "want_bias", "dw_out", "db_out", "event")

    def __init__(self, dy, x, want_bias, dw_out, db_out):
        self.dy, self.x, self.want_bias, self.dw_out, self.db_out, self.event = dy, x, bool(want_bias), dw_out, db_out, None

    def tensors(self):
        return (self.dy, self.x, self.dw_out, self.db_out)

    def __call__(self):
        weight_grad(self.dy, self.x, self.want_bias, dw_out=self.dw_out, db_out=self.db_out)


class ColsumItem:
    """One pending LayerNorm dgamma / dbeta column sum (fused._NormAct.backward under dp.defer_weight_grads)."""
    __slots__ = ("ws", "n", "H", "dgamma", "dbeta", "event")

    def __init__(self, ws, n, H, dgamma, dbeta):
        self.ws, self.n, self.H, self.dgamma, self.dbeta, self.event = ws, int(n), int(H), dgamma, dbeta, None

    def tensors(self):
        return (self.ws, self.dgamma, self.dbeta)

    def __call__(self):
        _lib.check(_lib.load().agnn_norm_act_colsum_f32(self.ws.data_ptr(), self.ws.numel(), self.n, self.H, self.dgamma.data_ptr(),
                                                        self.dbeta.data_ptr(), _lib.stream_ptr(self.ws.device)), "agnn_norm_act_colsum_f32")


def colsum_batch(items) -> None:
    if not BATCH or len(items) == 1:
        for it in items:
            it()
        return
    lib = _lib.load()
    for i in range(0, len(items), _lib.WGRAD_BATCH_MAX):
        grp = items[i:i + _lib.WGRAD_BATCH_MAX]
        arr = (_lib.ColsumItem * len(grp))()
        for a, it in zip(arr, grp):
            a.workspace, a.workspace_bytes, a.n, a.H = it.ws.data_ptr(), it.ws.numel(), it.n, it.H
            a.dgamma, a.dbeta = it.dgamma.data_ptr(), it.dbeta.data_ptr()
        _lib.check(lib.agnn_norm_act_colsum_batch_f32(len(grp), arr, _lib.stream_ptr(grp[0].ws.device)), "agnn_norm_act_colsum_batch_f32")


BATCH = True             # A/B switch for benchmarking: False = one launch pair per product, as in round 2


def weight_grad_batch(items) -> None:
    """`WgItem`s in as few launches as possible: those the MFMA kernel takes (fp32, even widths, >= MIN_ROWS rows, output up to
    MAX_OUT_IN, results written in place) go to `agnn_wgrad_batch_f32` in groups of up to 16, the rest one by one."""
    lib = None
    group = []

    def fits(it):
        n, out_f = it.dy.shape
        in_f = it.x.shape[1]
        return (BATCH and ENABLED and it.dy.is_cuda and it.dy.dim() == 2 and it.x.dim() == 2 and n >= MIN_ROWS and out_f * in_f <= MAX_OUT_IN and in_f % 2 == 0
                and _ok(it.dy) and _ok(it.x) and it.dw_out is not None and tuple(it.dw_out.shape) == (out_f, in_f) and it.dw_out.is_contiguous()
                and it.dw_out.data_ptr() % 8 == 0 and (not it.want_bias or (it.db_out is not None and it.db_out.is_contiguous() and it.db_out.data_ptr() % 8 == 0)))

    def run(group):
        if len(group) == 1:
            group[0]()
            return
        nonlocal lib
        lib = lib or _lib.load()
        dev = group[0].dy.device
        arr = (_lib.WgradItem * len(group))()
        for a, it in zip(arr, group):
            a.dy, a.x, a.dw = it.dy.data_ptr(), it.x.data_ptr(), it.dw_out.data_ptr()
            a.db = it.db_out.data_ptr() if it.want_bias else None
            a.ld_dy, a.ld_x, a.ld_dw, a.n = it.dy.stride(0), it.x.stride(0), it.dw_out.stride(0), it.dy.shape[0]
            a.out_f, a.in_f = it.dy.shape[1], it.x.shape[1]
        nws = int(lib.agnn_wgrad_batch_workspace_bytes(len(group), arr))
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        _lib.check(lib.agnn_wgrad_batch_f32(len(group), arr, ws.data_ptr(), nws, _lib.stream_ptr(dev)), "agnn_wgrad_batch_f32")

    for it in items:
        if fits(it):
            group.append(it)
            if len(group) == _lib.WGRAD_BATCH_MAX:
                run(group)
                group = []
        else:
            it()
    if group:
        run(group)


HAND_GEMM = True         # the forward projections on the hand-written fp32-MFMA kernel (csrc/gemm.hip, agnn_gemm_nt_f32); False (bench.py --set linear.HAND_GEMM=False): the library
HAND_GEMM_DX = False     # ... and their input gradients dX = dY W (agnn_gemm_nn_f32: the weight K-major, as it lies).  Off: in the backward
                         # pass's contended window the TunableOp-chosen library kernels are faster (C2 step 2.87 vs 2.96 ms); bench.py --set linear.HAND_GEMM_DX=True
HAND_GEMM_MIN_ROWS = 4096
HAND_GEMM_MIN_K, HAND_GEMM_MIN_N = 256, 256    # shorter K (8 steps: the tile's prologue / epilogue weigh as much as its MFMAs) or 128 columns: the
                                                # library keeps 15 - 25 % (project_enc's last layers, the heads' first layer: profiles/r03_gemm.md)
HAND_GEMM_MAX_K = 1536    # beyond it (C5's [16 000, 2 048] x [2 048, 512] layers) the library's 256-wide tiles keep 8 % on this kernel: C5 step 6.24 vs 6.17 ms


def _hand_gemm_dx_ok(dy, w) -> bool:
    return (dy.is_cuda and dy.dim() == 2 and dy.dtype == torch.float32 and w.dtype == torch.float32 and dy.shape[0] >= HAND_GEMM_MIN_ROWS
            and w.shape[1] % 64 == 0 and w.shape[0] % 16 == 0 and dy.stride(1) == 1 and w.stride(1) == 1 and dy.stride(0) % 4 == 0
            and w.stride(0) % 4 == 0 and dy.stride(0) >= dy.shape[1] and w.stride(0) >= w.shape[1] and dy.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0)


def _hand_gemm_ok(x, w, b) -> bool:
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and w.dtype == torch.float32 and x.shape[0] >= HAND_GEMM_MIN_ROWS
            and HAND_GEMM_MIN_K <= x.shape[1] <= HAND_GEMM_MAX_K and w.shape[0] >= HAND_GEMM_MIN_N and w.shape[0] % 64 == 0 and x.shape[1] % 16 == 0 and x.stride(1) == 1 and w.stride(1) == 1 and x.stride(0) % 4 == 0
            and w.stride(0) % 4 == 0 and x.stride(0) >= x.shape[1] and w.stride(0) >= w.shape[1] and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0 and (b is None or (b.dtype == torch.float32 and b.is_contiguous())))


def hand_gemm(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    """x W^T (+ b) on the hand-written fp32-MFMA kernel (csrc/gemm.hip); operands as `_hand_gemm_ok` checks."""
    y = torch.empty((x.shape[0], w.shape[0]), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().agnn_gemm_nt_f32(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), _lib.ptr(b), x.shape[0], w.shape[0], x.shape[1],
                                            y.data_ptr(), y.stride(0), _lib.stream_ptr(x.device)), "agnn_gemm_nt_f32")
    return y


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, acc, pre=None):
        """`pre` = [y]: the product was computed by a fused producer (heads.fused_head_logits) — the node only carries the backward pass."""
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        ctx.wg_async = _async_ok(w) and _async_ok(b)
        ctx.wg_defer = _deferrable(w) and _deferrable(b)
        ctx.steal_refs = leaf_refs(w, b)
        ctx.grad_cols = getattr(x, "_agnn_grad_cols", None) if x.dim() == 2 else None     # (embedding.embed_cat: only these columns of dX are read)
        ctx.set_materialize_grads(False)          # an undefined output gradient (a structurally dead branch) stays undefined upstream
        if pre is not None:
            return pre[0]
        if acc is not None:                       # y = acc + x W^T (+ b): the GEMM's beta = 1 epilogue, no separate add
            y = torch.addmm(acc, x, w.t())
            return y + b if b is not None else y
        if HAND_GEMM and _hand_gemm_ok(x, w, b):
            return hand_gemm(x, w, b)
        return torch.addmm(b, x, w.t()) if b is not None else x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return None, None, None, None, None
        x, w = ctx.saved_tensors
        dw = db = None
        want_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        steals = ctx.wg_async and all_steal(ctx.steal_refs)     # gradients taken over without a kernel on this stream
        in_f = x.shape[1]
        padded = bool(in_f & 1) and x.dim() == 2 and x.stride(1) == 1 and x.stride(0) > in_f     # a spare zero column behind the last one
        if want_w and steals and ctx.wg_defer and (in_f % 2 == 0 or padded) and deferring(dy):
            # only dX stays here.  The closure fills aliases: a second reference to `dw` itself would make AccumulateGrad
            # CLONE it (now, before it is computed) instead of taking it over.
            dw = torch.empty((dy.shape[1], in_f), dtype=torch.float32, device=dy.device)
            db = torch.empty((dy.shape[1],), dtype=torch.float32, device=dy.device) if want_b else None
            dw_k, db_k = dw.detach(), (db.detach() if db is not None else None)
            if padded:                   # the product runs on in + 1 columns (models.encode's 281-wide note input); its first `in` are copied out
                dw_wide = torch.empty((dy.shape[1], in_f + 1), dtype=torch.float32, device=dy.device)
                defer(WgItem(dy, x.as_strided((x.shape[0], in_f + 1), (x.stride(0), 1), x.storage_offset()), want_b, dw_wide, db_k), dy.device, here=True)       # stays with its copy
                defer(lambda: dw_k.copy_(dw_wide[:, :in_f]), dy.device)
            else:
                defer(WgItem(dy, x, want_b, dw_k, db_k), dy.device)
        elif want_w:
            # forked before dX is queued: both start at once — only when the gradients will be STOLEN (no kernel on the main stream)
            with wgrad_stream(dy.device, dy, x, active=steals):
                dw, db = weight_grad(dy, x, want_b)
        dx = None
        if ctx.needs_input_grad[0]:
            gc = ctx.grad_cols
            if gc is not None and x.stride(1) == 1 and x.stride(0) >= x.shape[1] and w.stride(1) == 1 and dy.is_cuda:
                # the producer only reads columns [c0, c1) of dX (the other inputs are data): the product over those columns of W,
                # written in place into a matrix with the producer's row stride; the rest of it stays unwritten
                c0, c1 = gc
                full = torch.empty((dy.shape[0], x.stride(0)), dtype=dy.dtype, device=dy.device)
                torch.mm(dy, w[:, c0:c1], out=full[:, c0:c1])
                dx = full[:, :x.shape[1]]
            elif HAND_GEMM and HAND_GEMM_DX and _hand_gemm_dx_ok(dy, w):
                dx = torch.empty((dy.shape[0], w.shape[1]), dtype=torch.float32, device=dy.device)
                _lib.check(_lib.load().agnn_gemm_nn_f32(dy.data_ptr(), dy.stride(0), w.data_ptr(), w.stride(0), None, dy.shape[0], w.shape[1], w.shape[0],
                                                        dx.data_ptr(), dx.stride(0), _lib.stream_ptr(dy.device)), "agnn_gemm_nn_f32")
            else:
                dx = dy @ w
        return dx, dw, db, (dy if ctx.needs_input_grad[3] else None), None


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None, acc: Optional[torch.Tensor] = None, pre=None) -> torch.Tensor:
    """x W^T (+ b) (+ acc).  `pre` = [y]: already computed elsewhere (2-D x only) — only the autograd node is created."""
    if pre is not None:
        if torch.is_grad_enabled() and (x.requires_grad or w.requires_grad or (b is not None and b.requires_grad)):
            return _LinearFn.apply(x, w, b, None, pre)
        return pre[0]
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
