"""HGT path: `HGTConv` with PyG (>= 2.3) parameter names and semantics, the
layer stack, and the `HybridHGT` encoder the reference constructs at
analysisgnn/models/analysis.py:445-453 (`heads=4`).  The per-edge work (scores, edge softmax over all
incoming relations, weighted sum, and their gradients) runs on the C-ABI kernels
`agnn_hgt_attn_*`; the per-type K/Q/V and output projections and the per-(relation, head) D x D
transforms are library GEMMs.  Build-spec notes (parity unpinned vs graphmuse): encoders.py header."""
from __future__ import annotations

import contextlib
import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .encoders import TrimPlan, _HybridMixin
from .graph import Csr, HeteroIndex, hetero_index
from .fused import skip_act
from .linear import all_steal, defer, deferring, leaf_refs, linear
from .params import cat_rows, pack

EdgeType = Tuple[str, str, str]


class _AttnSpec:
    def __init__(self, fwd: List[Csr], bwd: List[Csr], n_rows: int, heads: int, n_edges: List[int],
                 e_limit: Optional[List[Optional[int]]], src_rows: List[int]):
        self.fwd, self.bwd, self.n_rows, self.heads = fwd, bwd, n_rows, heads
        self.n_edges, self.e_limit, self.src_rows = n_edges, e_limit, src_rows

    def limit(self, r):
        return None if self.e_limit is None else self.e_limit[r]


def _mat(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32 or t.dim() != 2:
        raise _lib.AgnnError("fp32 2-D matrix expected")
    if t.stride(1) != 1 or t.stride(0) % 4 or t.data_ptr() % 16:
        t = t.contiguous()
    return t


class _HGTAttention(torch.autograd.Function):
    """M = edge_softmax_attention(q; {k'_r, v'_r}); inputs: q [N_d,H], pscale [R,heads], then k'_0, v'_0, k'_1, ..."""

    @staticmethod
    def forward(ctx, spec: _AttnSpec, q, pscale, *kv):
        dev = _lib.require_gpu(q, pscale, *kv)
        lib = _lib.load()
        R = len(spec.fwd)
        n, H, heads = spec.n_rows, q.shape[1], spec.heads
        q = _mat(q)
        pscale = pscale.contiguous()
        ks = [_mat(kv[2 * r]) for r in range(R)]
        vs = [_mat(kv[2 * r + 1]) for r in range(R)]
        out = torch.empty((n, H), dtype=torch.float32, device=dev)
        m = torch.empty((max(n, 1), heads), dtype=torch.float32, device=dev)
        linv = torch.empty((max(n, 1), heads), dtype=torch.float32, device=dev)
        rels = (_lib.HgtRel * max(R, 1))()
        keep = []
        for r in range(R):
            c = spec.fwd[r]
            re = c.rowend(spec.limit(r))
            keep.append(re)
            if ks[r].stride(0) != vs[r].stride(0):
                vs[r] = vs[r].contiguous()
                ks[r] = ks[r].contiguous()
            rels[r].k, rels[r].v = ks[r].data_ptr(), vs[r].data_ptr()
            rels[r].rowptr, rels[r].rowend = c.rowptr.data_ptr(), _lib.ptr(re)
            rels[r].col, rels[r].perm = c.col.data_ptr(), c.perm.data_ptr()
            rels[r].pscale = pscale[r].data_ptr()
            rels[r].ld = ks[r].stride(0)
        if n > 0:
            _lib.check(lib.agnn_hgt_attn_fwd_f32(R, rels, q.data_ptr(), q.stride(0), n, H, heads, out.data_ptr(),
                                                 out.stride(0), m.data_ptr(), linv.data_ptr(), _lib.stream_ptr(dev)),
                       "agnn_hgt_attn_fwd_f32")
        ctx.spec = spec
        ctx.save_for_backward(q, pscale, out, m, linv, *ks, *vs)
        return out

    @staticmethod
    def backward(ctx, dm):
        spec: _AttnSpec = ctx.spec
        q, pscale, out, m, linv, *rest = ctx.saved_tensors
        R = len(spec.fwd)
        ks, vs = rest[:R], rest[R:]
        dev = dm.device
        lib = _lib.load()
        n, H, heads = spec.n_rows, q.shape[1], spec.heads
        dm = _mat(dm)
        dq = torch.empty_like(q) if n == q.shape[0] else torch.zeros_like(q)
        # per-edge arrays of all relations in three allocations; only tdot is summed over ALL edges (trimmed ones
        # included), so only it needs zeros — alpha / gs are read back solely at positions the kernel wrote
        ne = [max(spec.n_edges[r], 1) for r in range(R)]
        offs = [0]
        for k_ in ne:
            offs.append(offs[-1] + k_)
        a_all = torch.empty((offs[-1], heads), dtype=torch.float32, device=dev)
        g_all = torch.empty((offs[-1], heads), dtype=torch.float32, device=dev)
        alpha = [a_all[offs[r]:offs[r + 1]] for r in range(R)]
        gs = [g_all[offs[r]:offs[r + 1]] for r in range(R)]
        # tdot as [R, max E_r, heads] (zero padded): the per-relation sums over edges are ONE reduction instead of R
        t3 = torch.zeros((max(R, 1), max(ne) if ne else 1, heads), dtype=torch.float32, device=dev)
        tdot = [t3[r, :ne[r]] for r in range(R)]
        rels = (_lib.HgtRel * max(R, 1))()
        keep = []
        for r in range(R):
            c = spec.fwd[r]
            re = c.rowend(spec.limit(r))
            keep.append(re)
            rels[r].k, rels[r].v = ks[r].data_ptr(), vs[r].data_ptr()
            rels[r].rowptr, rels[r].rowend = c.rowptr.data_ptr(), _lib.ptr(re)
            rels[r].col, rels[r].perm = c.col.data_ptr(), c.perm.data_ptr()
            rels[r].pscale = pscale[r].data_ptr()
            rels[r].ld = ks[r].stride(0)
            rels[r].alpha, rels[r].gs, rels[r].tdot = alpha[r].data_ptr(), gs[r].data_ptr(), tdot[r].data_ptr()
        if n > 0:
            _lib.check(lib.agnn_hgt_attn_bwd_dst_f32(R, rels, q.data_ptr(), q.stride(0), dm.data_ptr(), dm.stride(0),
                                                     out.data_ptr(), out.stride(0), m.data_ptr(), linv.data_ptr(), n, H,
                                                     heads, dq.data_ptr(), dq.stride(0), _lib.stream_ptr(dev)),
                       "agnn_hgt_attn_bwd_dst_f32")
        dps = t3.sum(dim=1) if R else torch.zeros_like(pscale)
        grads = []
        for r in range(R):
            c = spec.bwd[r]
            n_src = spec.src_rows[r]
            dk = torch.empty((n_src, H), dtype=torch.float32, device=dev)
            dv = torch.empty((n_src, H), dtype=torch.float32, device=dev)
            re = c.rowend(spec.limit(r))
            lim = n if spec.fwd[r].n_rows > n else _lib.INT32_MAX
            if n_src > 0:
                _lib.check(lib.agnn_hgt_attn_bwd_src_f32(c.rowptr.data_ptr(), _lib.ptr(re), c.col.data_ptr(),
                                                         c.perm.data_ptr(), alpha[r].data_ptr(), gs[r].data_ptr(),
                                                         q.data_ptr(), q.stride(0), dm.data_ptr(), dm.stride(0), n_src, lim,
                                                         H, heads, dk.data_ptr(), dv.data_ptr(), dk.stride(0),
                                                         _lib.stream_ptr(dev)), "agnn_hgt_attn_bwd_src_f32")
            grads += [dk, dv]
        return (None, dq, dps, *grads)


class _ColSplit(torch.autograd.Function):
    """G column blocks of a [N, G*H] matrix as views (row stride G*H, no copies).  Plain slicing would make autograd
    materialise one zero-filled [N, G*H] gradient per block and add them up; here the backward gathers the blocks'
    gradients into one [N, G*H] buffer with a single `agnn_pack_f32` launch."""

    @staticmethod
    def forward(ctx, big, G: int):
        N, W = big.shape
        ctx.meta = (N, W, G)
        ctx.set_materialize_grads(False)           # unused blocks: zero-filled in place below, not as separate tensors first
        H = W // G
        return tuple(big[:, g * H:(g + 1) * H] for g in range(G))

    @staticmethod
    def backward(ctx, *grads):
        N, W, G = ctx.meta
        H = W // G
        ref = next((g for g in grads if g is not None), None)
        if ref is None:
            return None, None
        dbig = torch.empty((N, W), dtype=torch.float32, device=ref.device)
        items = []
        for g_i, g in enumerate(grads):
            blk = dbig[:, g_i * H:(g_i + 1) * H]
            if g is None:
                blk.zero_()
            elif N > 0:
                items.append((blk, [_mat(g)]))
        if items:
            pack(items, ref.device)
        return dbig, None


def col_split(big: torch.Tensor, G: int):
    return _ColSplit.apply(big, G)


class _DictLinear(nn.Module):
    """`HeteroDictLinear`: one Linear per node type, parameters under `lins.<type>`."""

    def __init__(self, in_channels, out_channels, types: Sequence[str]):
        super().__init__()
        self.lins = nn.ModuleDict({t: nn.Linear(in_channels, out_channels) for t in types})

    def forward(self, x_dict):
        return {k: self.lins[k](v) for k, v in x_dict.items() if k in self.lins}


class _RelWeight(nn.Module):
    """`HeteroLinear(D, D, num_types=heads*edge_types, bias=False)`: weight [T, D, D], y = x @ W[type]."""

    def __init__(self, num_types, dim):
        super().__init__()
        bound = 1.0 / math.sqrt(dim)
        self.weight = nn.Parameter(torch.empty(num_types, dim, dim).uniform_(-bound, bound))


_SEL_CACHE: Dict[tuple, torch.Tensor] = {}


def _index_tensor(ids: tuple, device) -> torch.Tensor:
    key = (ids, str(device))
    if key not in _SEL_CACHE:
        _SEL_CACHE[key] = torch.tensor(list(ids), device=device)
    return _SEL_CACHE[key]


class _BlockDiagWeight(torch.autograd.Function):
    """The operand of x W^T for the relations `rel_ids` leaving one source type: row block r holds (blockdiag_h A_{r,h})^T,
    i.e. out[r*H + h*D + j, h*D + i] = weight[rel_ids[r]*heads + h, i, j] and zeros elsewhere, so that
    (x W^T)[:, r*H:(r+1)*H] = concat_h(x_h A_{r,h}).  Three launches forward (zeros, gather, block scatter), three
    backward (gather of the diagonal blocks, zeros, scatter) — plain indexing costs ~15 and its autograd chain as many."""

    @staticmethod
    def forward(ctx, weight, rel_ids: tuple, n_rel: int, heads: int, D: int):
        dev = weight.device
        R, H = len(rel_ids), heads * D
        sel = _index_tensor(tuple(r * heads + h for r in rel_ids for h in range(heads)), dev)      # [R*heads]
        blocks = weight.detach().index_select(0, sel).view(R, heads, D, D)                         # A_{r,h}[i, j]
        out = weight.new_zeros((R, heads, D, heads, D))                                             # [r, h, j, h', i]
        ar = _arange(heads, dev)
        out[:, ar, :, ar, :] = blocks.permute(1, 0, 3, 2)                                           # (h, r, j, i) -> out[r, h, j, h, i]
        ctx.sel, ctx.meta = sel, (R, heads, D, weight.shape[0])
        return out.view(R * H, H)

    @staticmethod
    def backward(ctx, g):
        R, heads, D, T = ctx.meta
        ar = _arange(heads, g.device)
        gb = g.reshape(R, heads, D, heads, D)[:, ar, :, ar, :]                                      # [h, r, j, i]
        gw = g.new_zeros((T, D, D))
        gw.index_copy_(0, ctx.sel, gb.permute(1, 0, 3, 2).reshape(R * heads, D, D))                # back to [r*heads + h, i, j]
        return gw, None, None, None, None


RELT_D = 64           # the head width the relation-transform kernels are built for (csrc/relt.hip)
RELT_ENABLED = True   # A/B switch: False = round 1's dense GEMM against a block-diagonal weight
CORE_ENABLED = True   # A/B switch: False = one autograd node per relation / destination type (round 1's graph)
ATTN_ONE_LAUNCH = True  # A/B switch (bench.py --set hgt.ATTN_ONE_LAUNCH=False): the destination types' forward attention in one launch
TYPE_STREAMS = True   # A/B switch (bench.py --set hgt.TYPE_STREAMS=False): the small node types' projections on a second stream
_TYPE_STREAM: dict = {}


def _type_stream(dev) -> "torch.cuda.Stream":
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    s = _TYPE_STREAM.get(idx)
    if s is None:
        s = _TYPE_STREAM[idx] = torch.cuda.Stream(device=dev)
    return s


class _RelTransform(torch.autograd.Function):
    """(k', v') for the relations `rel_ids` leaving one source type: k'[:, r*H + h*D + j] = sum_i k[:, h*D + i] *
    k_rel.weight[rel_ids[r]*heads + h, i, j] (same for v) — `agnn_relt_*`: R*heads independent D x D products per operand on
    the fp32 MFMA, K and V in one launch, instead of one dense [N, H] x [H, R*H] GEMM against a block-diagonal weight
    (heads x the useful FLOPs) and the launches that assemble it."""

    @staticmethod
    def forward(ctx, k, v, wk, wv, rel_ids: tuple, heads: int, D: int):
        dev = _lib.require_gpu(k, v, wk, wv)
        lib = _lib.load()
        k, v = _mat(k), _mat(v)
        R, H, N = len(rel_ids), heads * D, k.shape[0]
        sel = _index_tensor(tuple(r * heads + h for r in rel_ids for h in range(heads)), dev)
        Wk = wk.detach().index_select(0, sel)                     # [R*heads, D, D], contiguous
        Wv = wv.detach().index_select(0, sel)
        yk = torch.empty((N, R * H), dtype=torch.float32, device=dev)
        yv = torch.empty((N, R * H), dtype=torch.float32, device=dev)
        items = (_lib.ReltItem * 2)()
        for it, (x, w, y) in zip(items, ((k, Wk, yk), (v, Wv, yv))):
            it.x, it.w, it.y, it.ld_x, it.ld_y = x.data_ptr(), w.data_ptr(), y.data_ptr(), x.stride(0), y.stride(0)
        if N > 0:
            _lib.check(lib.agnn_relt_fwd_f32(2, items, R, heads, D, N, _lib.stream_ptr(dev)), "agnn_relt_fwd_f32")
        ctx.save_for_backward(k, v, Wk, Wv, sel)
        ctx.meta = (R, heads, D, tuple(wk.shape), tuple(wv.shape))
        return yk, yv

    @staticmethod
    def backward(ctx, dyk, dyv):
        k, v, Wk, Wv, sel = ctx.saved_tensors
        R, heads, D, wk_shape, wv_shape = ctx.meta
        dev = k.device
        lib = _lib.load()
        N, H = k.shape[0], heads * D
        zk = dyk is None
        dyk = _mat(dyk) if dyk is not None else torch.zeros((N, R * H), dtype=torch.float32, device=dev)
        dyv = _mat(dyv) if dyv is not None else torch.zeros((N, R * H), dtype=torch.float32, device=dev)
        dk = dv = gwk = gwv = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dk = torch.empty((N, H), dtype=torch.float32, device=dev)
            dv = torch.empty((N, H), dtype=torch.float32, device=dev)
            Wkt, Wvt = Wk.transpose(1, 2).contiguous(), Wv.transpose(1, 2).contiguous()
            items = (_lib.ReltItem * 2)()
            for it, (x, w, y) in zip(items, ((dyk, Wkt, dk), (dyv, Wvt, dv))):
                it.x, it.w, it.y, it.ld_x, it.ld_y = x.data_ptr(), w.data_ptr(), y.data_ptr(), x.stride(0), y.stride(0)
            if N > 0:
                _lib.check(lib.agnn_relt_bwd_f32(2, items, R, heads, D, N, _lib.stream_ptr(dev)), "agnn_relt_bwd_f32")
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            dWk, dWv = torch.empty_like(Wk), torch.empty_like(Wv)
            nws = int(lib.agnn_relt_dw_workspace_bytes(2, R, heads, D, N))
            ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=dev)
            items = (_lib.ReltItem * 2)()
            for it, (x, dy, y) in zip(items, ((k, dyk, dWk), (v, dyv, dWv))):
                it.x, it.w, it.y, it.ld_x, it.ld_y = x.data_ptr(), dy.data_ptr(), y.data_ptr(), x.stride(0), dy.stride(0)
            _lib.check(lib.agnn_relt_dw_f32(2, items, R, heads, D, N, ws.data_ptr(), nws, _lib.stream_ptr(dev)), "agnn_relt_dw_f32")
            gwk = torch.zeros(wk_shape, dtype=torch.float32, device=dev).index_copy_(0, sel, dWk)
            gwv = torch.zeros(wv_shape, dtype=torch.float32, device=dev).index_copy_(0, sel, dWv)
        return dk, dv, gwk, gwv, None, None, None


class _CorePlan:
    """Static layout of one HGTConv layer on one batch (nothing that carries a gradient)."""

    def __init__(self, types, n_of, heads, D, src_rels, dst_rels, block_of, index, e_keep, n_edge_types):
        self.types, self.n_of, self.heads, self.D = types, n_of, heads, D
        self.src_rels, self.dst_rels, self.block_of = src_rels, dst_rels, block_of
        self.index, self.e_keep, self.n_edge_types = index, e_keep, n_edge_types

    def limit(self, et):
        return None if self.e_keep is None else self.e_keep[et]


class _HGTCore(torch.autograd.Function):
    """Everything of an HGTConv layer between the K|Q|V projections and the output projections, for ALL node types at
    once:  kqv[t] [n_t, 3H]  ->  m[t] [n_t, H]  (relation transforms, scores, edge softmax over all incoming relations,
    weighted sum).  One autograd node instead of ~15 per layer, so that every gradient is WRITTEN IN PLACE by a kernel:
      * the attention's source-side pass writes dK' / dV' of a relation straight into its column block of the source type's
        [N_s, R_s*H] gradient (no per-relation tensors, no packing launch, no zero-filled placeholders);
      * the input-gradient relation transform writes dk / dv, and the destination-side pass writes dq, straight into the
        three column blocks of d kqv[t] (no [N, 3H] re-assembly);
      * the 1/sqrt(D) scale of p_rel, the per-relation stacking and the gradient fan-out are one small launch each way."""

    @staticmethod
    def forward(ctx, plan: _CorePlan, wk, wv, p_all, *kqv):
        dev = _lib.require_gpu(wk, wv, p_all, *kqv)
        lib = _lib.load()
        heads, D = plan.heads, plan.D
        H = heads * D
        X = {t: _mat(x) for t, x in zip(plan.types, kqv)}
        for t, x in X.items():
            if x.shape[1] != 3 * H:
                raise _lib.AgnnError("HGT core: kqv must be [n, 3H]")
        # relation transforms, K and V of one source type in one launch
        kp, vp, Wk, Wv = {}, {}, {}, {}
        for s_t, e_ids in plan.src_rels.items():
            x = X[s_t]
            N, R = x.shape[0], len(e_ids)
            ids = tuple(e * heads + h for e in e_ids for h in range(heads))
            if ids == tuple(range(wk.shape[0])) and wk.is_contiguous() and wv.is_contiguous() and wk.data_ptr() % 16 == 0 and wv.data_ptr() % 16 == 0:
                Wk[s_t], Wv[s_t] = wk.detach(), wv.detach()          # every relation leaves this type, in order: no gather
            else:
                sel = _index_tensor(ids, dev)
                Wk[s_t], Wv[s_t] = wk.detach().index_select(0, sel), wv.detach().index_select(0, sel)
            kp[s_t] = torch.empty((N, R * H), dtype=torch.float32, device=dev)
            vp[s_t] = torch.empty((N, R * H), dtype=torch.float32, device=dev)
            if N > 0:
                items = (_lib.ReltItem * 2)()
                for it, (off, w, y) in zip(items, ((0, Wk[s_t], kp[s_t]), (2 * H, Wv[s_t], vp[s_t]))):
                    it.x, it.w, it.y, it.ld_x, it.ld_y = x.data_ptr() + 4 * off, w.data_ptr(), y.data_ptr(), x.stride(0), y.stride(0)
                _lib.check(lib.agnn_relt_fwd_f32(2, items, R, heads, D, N, _lib.stream_ptr(dev)), "agnn_relt_fwd_f32")
        ps = (p_all.detach() * (1.0 / math.sqrt(D))).contiguous()                   # [n_edge_types, heads]
        outs, stats = [], {}
        live = [t for t in plan.types if plan.dst_rels.get(t) and plan.n_of[t] > 0]
        one_launch = (ATTN_ONE_LAUNCH and 1 < len(live) <= _lib.HGT_MAX_DST
                      and sum(len(plan.dst_rels[t]) for t in live) <= _lib.MAX_SEG)
        items = (_lib.HgtDstItem * max(len(live), 1))()
        hold = []                                            # relation tables and row-end vectors: alive until the launch is issued
        for t in plan.types:
            n = plan.n_of[t]
            rels = plan.dst_rels.get(t, [])
            if not rels or n == 0:
                outs.append(torch.zeros((n, H), dtype=torch.float32, device=dev))
                continue
            out = torch.empty((n, H), dtype=torch.float32, device=dev)
            m = torch.empty((n, heads), dtype=torch.float32, device=dev)
            linv = torch.empty((n, heads), dtype=torch.float32, device=dev)
            arr, keep = _HGTCore._rel_table(plan, rels, kp, vp, ps, H)
            q = X[t]
            if one_launch:                               # every destination type's attention in ONE launch (agnn_hgt_attn_fwd_multi_f32)
                it = items[len(hold)]
                it.rels, it.n_rel = _lib.C.cast(arr, _lib.C.c_void_p), len(rels)
                it.q, it.ld_q, it.n_rows = q.data_ptr() + 4 * H, q.stride(0), n
                it.out, it.ld_out, it.m_out, it.linv_out = out.data_ptr(), out.stride(0), m.data_ptr(), linv.data_ptr()
                hold.append((arr, keep))
            else:
                _lib.check(lib.agnn_hgt_attn_fwd_f32(len(rels), arr, q.data_ptr() + 4 * H, q.stride(0), n, H, heads, out.data_ptr(),
                                                     out.stride(0), m.data_ptr(), linv.data_ptr(), _lib.stream_ptr(dev)), "agnn_hgt_attn_fwd_f32")
            stats[t] = (m, linv)
            outs.append(out)
        if one_launch:
            _lib.check(lib.agnn_hgt_attn_fwd_multi_f32(len(hold), items, H, heads, _lib.stream_ptr(dev)), "agnn_hgt_attn_fwd_multi_f32")
        ctx.plan = plan
        ctx.steal_refs = leaf_refs(wk, wv)             # (their gradients may be produced late: see backward)
        ctx.wg_defer = all(t.is_leaf or getattr(t, "_agnn_wgrad_deferrable", False) for t in (wk, wv))
        ctx.srcs = list(plan.src_rels.keys())
        ctx.dsts = list(stats.keys())
        saved = list(X.values()) + [ps]
        for s_t in ctx.srcs:
            saved += [Wk[s_t], Wv[s_t], kp[s_t], vp[s_t]]
        for t in ctx.dsts:
            saved += [stats[t][0], stats[t][1], outs[plan.types.index(t)]]
        ctx.save_for_backward(*saved)
        ctx.shapes = (tuple(wk.shape), tuple(wv.shape), tuple(p_all.shape))
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def _rel_table(plan, rels, kp, vp, ps, H, extra=None):
        arr = (_lib.HgtRel * max(len(rels), 1))()
        keep = []
        for r, (e_idx, et) in enumerate(rels):
            s_t, blk = plan.block_of[e_idx]
            c = plan.index.fwd[et]
            re = c.rowend(plan.limit(et))
            keep.append(re)
            arr[r].k, arr[r].v = kp[s_t].data_ptr() + 4 * blk * H, vp[s_t].data_ptr() + 4 * blk * H
            arr[r].ld = kp[s_t].stride(0)
            arr[r].rowptr, arr[r].rowend = c.rowptr.data_ptr(), _lib.ptr(re)
            arr[r].col, arr[r].perm = c.col.data_ptr(), c.perm.data_ptr()
            arr[r].pscale = ps[e_idx].data_ptr()
            if extra is not None:
                arr[r].alpha, arr[r].gs, arr[r].tdot = (x.data_ptr() for x in extra[r])
        return arr, keep

    @staticmethod
    def backward(ctx, *dms):
        plan: _CorePlan = ctx.plan
        lib = _lib.load()
        heads, D = plan.heads, plan.D
        H = heads * D
        saved = list(ctx.saved_tensors)
        nt = len(plan.types)
        X = dict(zip(plan.types, saved[:nt]))
        ps = saved[nt]
        pos = nt + 1
        Wk, Wv, kp, vp = {}, {}, {}, {}
        for s_t in ctx.srcs:
            Wk[s_t], Wv[s_t], kp[s_t], vp[s_t] = saved[pos:pos + 4]
            pos += 4
        stats = {}
        for t in ctx.dsts:
            stats[t] = tuple(saved[pos:pos + 3])
            pos += 3
        dev = ps.device
        st = _lib.stream_ptr(dev)
        dkqv = {t: torch.empty_like(X[t]) for t in plan.types}
        dkp = {s_t: torch.empty_like(kp[s_t]) for s_t in ctx.srcs}
        dvp = {s_t: torch.empty_like(vp[s_t]) for s_t in ctx.srcs}
        written = set()                                              # relations whose dK' / dV' block a kernel has written
        zero_q = set()                                               # types whose dq is structurally zero
        dps_rows, dps_ids = [], []
        keep_alive = []                      # row-end tensors referenced by a launch table until the launch is issued
        for ti, t in enumerate(plan.types):
            dm = dms[ti]
            n = plan.n_of[t]
            rels = plan.dst_rels.get(t, [])
            if t not in stats or dm is None:                         # no incoming relation / output not used downstream: dq = 0
                zero_q.add(t)
                continue
            dm = _mat(dm)
            m, linv, out = stats[t]
            q = X[t]
            R = len(rels)
            ne = [max(plan.index.num_edges[et], 1) for _, et in rels]
            offs = [0]
            for k_ in ne:
                offs.append(offs[-1] + k_)
            a_all = torch.empty((offs[-1], heads), dtype=torch.float32, device=dev)
            g_all = torch.empty((offs[-1], heads), dtype=torch.float32, device=dev)
            # tdot head-major as [R, heads, max E_r] (zero padded): the sums over the edges are ONE contiguous reduction
            t3 = torch.zeros((R, heads, max(ne)), dtype=torch.float32, device=dev)
            extra = [(a_all[offs[r]:offs[r + 1]], g_all[offs[r]:offs[r + 1]], t3[r]) for r in range(R)]
            arr, keep = _HGTCore._rel_table(plan, rels, kp, vp, ps, H, extra)
            for r in range(R):
                arr[r].ld_tdot = max(ne)
            dq_ptr = dkqv[t].data_ptr() + 4 * H
            _lib.check(lib.agnn_hgt_attn_bwd_dst_f32(R, arr, q.data_ptr() + 4 * H, q.stride(0), dm.data_ptr(), dm.stride(0), out.data_ptr(),
                                                     out.stride(0), m.data_ptr(), linv.data_ptr(), n, H, heads, dq_ptr, dkqv[t].stride(0), st),
                       "agnn_hgt_attn_bwd_dst_f32")
            dps_rows.append(t3.sum(dim=2))
            dps_ids += [e_idx for e_idx, _ in rels]
            src_items = (_lib.HgtSrcItem * len(rels))()
            n_items = 0
            for r, (e_idx, et) in enumerate(rels):
                s_t, blk = plan.block_of[e_idx]
                c = plan.index.bwd[et]
                n_src = X[s_t].shape[0]
                re = c.rowend(plan.limit(et))
                lim = n if plan.index.fwd[et].n_rows > n else _lib.INT32_MAX
                if n_src > 0:
                    it = src_items[n_items]
                    n_items += 1
                    it.rowptr, it.rowend, it.col, it.perm = c.rowptr.data_ptr(), _lib.ptr(re), c.col.data_ptr(), c.perm.data_ptr()
                    it.alpha, it.gs = extra[r][0].data_ptr(), extra[r][1].data_ptr()
                    it.dk, it.dv = dkp[s_t].data_ptr() + 4 * blk * H, dvp[s_t].data_ptr() + 4 * blk * H
                    it.ld_o, it.n_src_rows, it.col_limit = dkp[s_t].stride(0), n_src, lim
                    keep_alive.append(re)
                written.add(e_idx)
            if n_items:                                   # every relation that ends in this type: one launch
                _lib.check(lib.agnn_hgt_attn_bwd_src_batch_f32(n_items, src_items, q.data_ptr() + 4 * H, q.stride(0), dm.data_ptr(), dm.stride(0),
                                                               H, heads, st), "agnn_hgt_attn_bwd_src_batch_f32")
        gwk = gwv = None
        wk_shape, wv_shape, p_shape = ctx.shapes
        later = ctx.wg_defer and dev.type == "cuda" and deferring(next(iter(X.values()))) and all_steal(ctx.steal_refs)
        # column blocks that no kernel writes (a type without live outgoing relations, a relation trimmed to nothing, a type whose
        # own output is unused) are cleared in ONE launch (pack with no sources); as `.zero_()` calls they were six ~6 us launches
        # per layer on the stack's backward chain
        clears = []
        for s_t in plan.types:
            live_src = s_t in kp and any(e_idx in written for e_idx in plan.src_rels[s_t])
            if not live_src:
                if s_t not in zero_q:
                    clears += [dkqv[s_t][:, :H], dkqv[s_t][:, 2 * H:]]          # no (live) outgoing relation: dk = dv = 0
                continue
            if s_t in zero_q:
                clears.append(dkqv[s_t][:, H:2 * H])
            for blk, e_idx in enumerate(plan.src_rels[s_t]):
                if e_idx not in written:
                    clears += [dkp[s_t][:, blk * H:(blk + 1) * H], dvp[s_t][:, blk * H:(blk + 1) * H]]
        clears = [c for c in clears if c.numel() > 0]
        if clears:
            pack([(c, []) for c in clears], dev)
        dead = set()                                                 # types whose whole d kqv is structurally zero: hand back None,
        for s_t in plan.types:                                       # so that autograd prunes everything upstream of it
            x = X[s_t]
            N = x.shape[0]
            live_src = s_t in kp and any(e_idx in written for e_idx in plan.src_rels[s_t])
            if not live_src:
                if s_t in zero_q:
                    dead.add(s_t)
                continue
            e_ids = plan.src_rels[s_t]
            R = len(e_ids)
            dx = dkqv[s_t]
            if N > 0:
                Wkt, Wvt = Wk[s_t].transpose(1, 2).contiguous(), Wv[s_t].transpose(1, 2).contiguous()
                items = (_lib.ReltItem * 2)()
                for it, (dy, w, off) in zip(items, ((dkp[s_t], Wkt, 0), (dvp[s_t], Wvt, 2 * H))):
                    it.x, it.w, it.y, it.ld_x, it.ld_y = dy.data_ptr(), w.data_ptr(), dx.data_ptr() + 4 * off, dy.stride(0), dx.stride(0)
                _lib.check(lib.agnn_relt_bwd_f32(2, items, R, heads, D, N, st), "agnn_relt_bwd_f32")
            # The relation weights' gradients only feed the optimizer: with dp.defer_weight_grads they leave the stack's backward
            # chain (95 us per layer at C3, between the input-gradient transform and the next layer's backward) and run at the
            # stream's flush.  The closure works on aliases (a second reference to the returned gradient would make
            # AccumulateGrad clone it before it is computed: see linear._LinearFn.backward).
            dWk, dWv = torch.empty_like(Wk[s_t]), torch.empty_like(Wv[s_t])
            ids = tuple(e * heads + h for e in e_ids for h in range(heads))
            whole = gwk is None and ids == tuple(range(wk_shape[0])) and len(ctx.srcs) == 1
            if whole:
                gwk, gwv = dWk, dWv                                  # the one source type covers the whole parameter
            elif gwk is None:
                gwk = torch.zeros(wk_shape, dtype=torch.float32, device=dev)
                gwv = torch.zeros(wv_shape, dtype=torch.float32, device=dev)

            def weight_grads(x=x, dk_=dkp[s_t], dv_=dvp[s_t], dWk=dWk.detach(), dWv=dWv.detach(), R=R, N=N, ids=ids, whole=whole,
                             gk=gwk.detach(), gv=gwv.detach()):
                nws = int(lib.agnn_relt_dw_workspace_bytes(2, R, heads, D, N))
                ws = torch.empty(max(nws, 1), dtype=torch.uint8, device=dev)
                items = (_lib.ReltItem * 2)()
                for it, (off, dy, y) in zip(items, ((0, dk_, dWk), (2 * H, dv_, dWv))):
                    it.x, it.w, it.y, it.ld_x, it.ld_y = x.data_ptr() + 4 * off, dy.data_ptr(), y.data_ptr(), x.stride(0), dy.stride(0)
                _lib.check(lib.agnn_relt_dw_f32(2, items, R, heads, D, N, ws.data_ptr(), nws, _lib.stream_ptr(dev)), "agnn_relt_dw_f32")
                if not whole:
                    sel = _index_tensor(ids, dev)
                    gk.index_copy_(0, sel, dWk)
                    gv.index_copy_(0, sel, dWv)

            if later:
                defer(weight_grads, dev)
            else:
                weight_grads()
        if dps_rows and tuple(dps_ids) == tuple(range(p_shape[0])):
            dp_all = (torch.cat(dps_rows, dim=0) * (1.0 / math.sqrt(D))).view(p_shape)
        else:
            dp_all = torch.zeros(p_shape, dtype=torch.float32, device=dev)
            if dps_rows:
                dp_all.index_copy_(0, _index_tensor(tuple(dps_ids), dev), torch.cat(dps_rows, dim=0) * (1.0 / math.sqrt(D)))
        if gwk is None:      # no live source type this step: still parameters of the step — zero gradients, as the SAGE path hands out
            gwk = torch.zeros(wk_shape, dtype=torch.float32, device=dev)
            gwv = torch.zeros(wv_shape, dtype=torch.float32, device=dev)
        # A structurally dead node type (no live outgoing relation, its own output unused) gets None: autograd prunes its
        # whole upstream.  PyG's HGTConv routes those parameters through empty index ops and they receive ZERO gradients;
        # dp.FlatGradBuffer.pack substitutes zeros, a stock torch optimizer needs dp.fill_missing_grads(model) first.
        return (None, gwk, gwv, dp_all, *[None if t in dead else dkqv[t] for t in plan.types])


def _arange(n: int, device) -> torch.Tensor:
    return _index_tensor(tuple(range(n)), device)


def block_diag_weight(weight: torch.Tensor, rel_ids: tuple, n_rel: int, heads: int, D: int) -> torch.Tensor:
    return _BlockDiagWeight.apply(weight, rel_ids, n_rel, heads, D)


class HGTConv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, metadata, heads: int = 1):
        super().__init__()
        if out_channels % heads:
            raise ValueError("out_channels must be divisible by heads")
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.node_types = list(metadata[0])
        self.edge_types = [tuple(e) for e in metadata[1]]
        D = out_channels // heads
        self.kqv_lin = _DictLinear(in_channels, out_channels * 3, self.node_types)
        self.out_lin = _DictLinear(out_channels, out_channels, self.node_types)
        self.k_rel = _RelWeight(heads * len(self.edge_types), D)
        self.v_rel = _RelWeight(heads * len(self.edge_types), D)
        self.skip = nn.ParameterDict({t: nn.Parameter(torch.ones(1)) for t in self.node_types})
        self.p_rel = nn.ParameterDict({"__".join(e): nn.Parameter(torch.ones(1, heads)) for e in self.edge_types})

    def forward(self, x_dict, edge_index_dict, index: Optional[HeteroIndex] = None,
                n_keep: Optional[Dict[str, int]] = None, e_keep: Optional[Dict[EdgeType, Optional[int]]] = None, post=None):
        """`post` = (relu, dropout p, training) or None: the activation the caller would apply to every output right after this
        layer (HeteroHGTStack between layers) — folded into the layer's epilogue launch; None: the plain HGTConv output."""
        _lib.require_gpu(*x_dict.values())
        if index is None:
            index = hetero_index(edge_index_dict, {k: int(v.shape[0]) for k, v in x_dict.items()})
        heads, H = self.heads, self.out_channels
        D = H // heads
        n_of = {t: (n_keep[t] if n_keep is not None else int(x.shape[0])) for t, x in x_dict.items()}
        if CORE_ENABLED and RELT_ENABLED and D == RELT_D and all(x.shape[1] == self.in_channels for x in x_dict.values()):
            return self._forward_core(x_dict, index, n_of, e_keep, post)
        # k | q | v in ONE projection per node type; the three H-wide column blocks are handed out as views (col_split)
        k, q, v = {}, {}, {}
        for t, x in x_dict.items():
            lin = self.kqv_lin.lins[t]
            xt = x if n_of[t] >= x.shape[0] else x[:n_of[t]]
            k[t], q[t], v[t] = col_split(linear(xt, lin.weight, lin.bias), 3)
        by_dst: Dict[str, List[Tuple[int, EdgeType]]] = {}
        for e_idx, et in enumerate(self.edge_types):
            s, _, d = et
            if et in index.fwd and s in x_dict and d in x_dict:
                by_dst.setdefault(d, []).append((e_idx, et))
        # Relation transforms k' = k A_r^k, v' = v A_r^v: the heads' D x D matrices of one relation are laid out as one
        # block-diagonal [H, H] weight (one scatter for all relations), so each relation is a plain [N_s, H] x [H, H] GEMM
        # with a well-shaped weight gradient, instead of batched 64 x 64 GEMMs whose weight gradients have K = N.
        used = sorted({e_idx for rels in by_dst.values() for e_idx, _ in rels})
        kv_of: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}
        if used:
            # all relations leaving one source type share their input: ONE GEMM [N_s, H] x [H, R_s*H] for the keys and one
            # for the values (weight gradient [R_s*H, H] in one piece), the relations' blocks handed out as column views
            by_src: Dict[str, List[int]] = {}
            for e_idx in used:
                by_src.setdefault(self.edge_types[e_idx][0], []).append(e_idx)
            for s_t, e_ids in by_src.items():
                if RELT_ENABLED and D == RELT_D and k[s_t].is_cuda:
                    kp, vp = _RelTransform.apply(k[s_t], v[s_t], self.k_rel.weight, self.v_rel.weight, tuple(e_ids), heads, D)
                    ks, vs = col_split(kp, len(e_ids)), col_split(vp, len(e_ids))
                else:
                    Wk_s = block_diag_weight(self.k_rel.weight, tuple(e_ids), len(self.edge_types), heads, D)     # [R_s*H, H]
                    Wv_s = block_diag_weight(self.v_rel.weight, tuple(e_ids), len(self.edge_types), heads, D)
                    ks = col_split(linear(k[s_t], Wk_s), len(e_ids))
                    vs = col_split(linear(v[s_t], Wv_s), len(e_ids))
                for j, e_idx in enumerate(e_ids):
                    kv_of[e_idx] = (ks[j], vs[j])
        out = {}
        for t, x in x_dict.items():
            n = n_of[t]
            rels = by_dst.get(t, [])
            if rels and n > 0:
                kv, ps = [], []
                for e_idx, et in rels:
                    kv += list(kv_of[e_idx])
                    ps.append(self.p_rel["__".join(et)].reshape(heads) / math.sqrt(D))
                spec = _AttnSpec(fwd=[index.fwd[et] for _, et in rels], bwd=[index.bwd[et] for _, et in rels], n_rows=n,
                                 heads=heads, n_edges=[index.num_edges[et] for _, et in rels],
                                 e_limit=[e_keep[et] for _, et in rels] if e_keep is not None else None,
                                 src_rows=[kv[2 * i].shape[0] for i in range(len(rels))])
                m = _HGTAttention.apply(spec, q[t], torch.stack(ps), *kv)
            else:
                m = x.new_zeros((n, H))
            o = self.out_lin.lins[t](F.gelu(m))
            relu, p, training = post if post is not None else (False, 0.0, False)
            xs = (x if n >= x.shape[0] else x[:n]) if o.shape[-1] == x.shape[-1] else None
            out[t] = skip_act(o, xs, self.skip[t] if xs is not None else None, relu, p, training)
        return out


    def adjacent_parameter_groups(self):
        """The per-relation priors are consumed stacked (one [n_edge_types, heads] operand): adjacent in the flat buffer
        (dp.plan_parameters) the stack is a view."""
        return [[self.p_rel["__".join(e)] for e in self.edge_types]]

    def _forward_core(self, x_dict, index: HeteroIndex, n_of, e_keep, post=None):
        """One K|Q|V GEMM per node type -> `_HGTCore` (one autograd node for the whole message passing) -> one output GEMM
        per node type (+ skip)."""
        heads, H = self.heads, self.out_channels
        D = H // heads
        types = list(x_dict.keys())
        dev = x_dict[types[0]].device
        # The node types' projections are independent of each other, and every type but the largest is small (C3: 16 000 notes,
        # ~2 600 beats, ~700 measures): their kernels are launch-latency bound and, in one stream, sat between the note type's
        # — 130 us per layer of the stack's serial chain.  They run on a second stream beside the note type's kernels.
        big = max(types, key=lambda t: n_of[t])
        two = TYPE_STREAMS and dev.type == "cuda" and len(types) > 1
        main = torch.cuda.current_stream(dev) if two else None
        side = _type_stream(dev) if two else None

        def where(t):
            return torch.cuda.stream(side) if (two and t != big) else contextlib.nullcontext()
        if two:
            side.wait_stream(main)
        kqv = []
        for t in types:
            lin = self.kqv_lin.lins[t]
            x = x_dict[t]
            with where(t):
                if two and t != big:
                    x.record_stream(side)
                kqv.append(linear(x if n_of[t] >= x.shape[0] else x[:n_of[t]], lin.weight, lin.bias))
        if two:
            main.wait_stream(side)
            for t, y in zip(types, kqv):
                if t != big:
                    y.record_stream(main)
        src_rels: Dict[str, List[int]] = {}
        dst_rels: Dict[str, List[Tuple[int, EdgeType]]] = {}
        for e_idx, et in enumerate(self.edge_types):
            s_t, _, d_t = et
            if et in index.fwd and s_t in x_dict and d_t in x_dict:
                src_rels.setdefault(s_t, []).append(e_idx)
                dst_rels.setdefault(d_t, []).append((e_idx, et))
        block_of = {e_idx: (s_t, blk) for s_t, ids in src_rels.items() for blk, e_idx in enumerate(ids)}
        plan = _CorePlan(types, n_of, heads, D, src_rels, dst_rels, block_of, index, e_keep, len(self.edge_types))
        p_all = cat_rows([self.p_rel["__".join(e)] for e in self.edge_types])       # [n_edge_types, heads]
        ms = _HGTCore.apply(plan, self.k_rel.weight, self.v_rel.weight, p_all, *kqv)
        out = {}
        if two:
            side.wait_stream(main)
        for t, m in zip(types, ms):
            x, n = x_dict[t], n_of[t]
            lo = self.out_lin.lins[t]
            with where(t):
                if two and t != big:
                    m.record_stream(side)
                o = linear(F.gelu(m), lo.weight, lo.bias)
                # skip connection (+ the ReLU / dropout the stack puts between layers, when it hands them in): one launch each way
                relu, p, training = post if post is not None else (False, 0.0, False)
                xs = (x if n >= x.shape[0] else x[:n]) if o.shape[-1] == x.shape[-1] else None
                out[t] = skip_act(o, xs, self.skip[t] if xs is not None else None, relu, p, training)
        if two:
            main.wait_stream(side)
            for t in types:
                if t != big:
                    out[t].record_stream(main)
        return out


class HeteroHGTStack(nn.Module):
    def __init__(self, metadata, input_channels, hidden_channels, num_layers, heads=4, dropout=0.5):
        super().__init__()
        self.num_layers, self.dropout = num_layers, dropout
        self.convs = nn.ModuleList([HGTConv(input_channels if i == 0 else hidden_channels, hidden_channels, metadata, heads)
                                    for i in range(num_layers)])

    def forward(self, x_dict, edge_index_dict, plan: TrimPlan, collect: Optional[list] = None):
        index = hetero_index(edge_index_dict, {k: int(v.shape[0]) for k, v in x_dict.items()})
        index.prepare_trim(plan.e_keep)
        self.last_index = index
        for i, conv in enumerate(self.convs):
            post = (True, self.dropout, self.training) if i < self.num_layers - 1 else None
            x_dict = conv(x_dict, edge_index_dict, index, plan.n_keep[i], plan.e_keep[i], post)
            if collect is not None:
                collect.append(x_dict["note"])
        return x_dict


class HybridHGT(nn.Module, _HybridMixin):
    """analysis.py:445-453 constructor (`heads=4`); forward keywords analysis.py:576-579."""

    def __init__(self, metadata, input_channels, hidden_channels, num_layers, heads=4, dropout=0.5, use_jk=False, **kwargs):
        super().__init__()
        self.metadata = (list(metadata[0]), [tuple(e) for e in metadata[1]])
        self.num_layers = num_layers
        self.gnn = HeteroHGTStack(self.metadata, input_channels, hidden_channels, num_layers, heads, dropout)
        self._init_hybrid(input_channels, hidden_channels, num_layers, dropout, use_jk)

    def forward(self, x_dict, edge_index_dict, batch_dict=None, batch_size=None, neighbor_mask_node=None,
                neighbor_mask_edge=None, return_edge_index=False, edge_attr_dict=None):
        _lib.require_gpu(*x_dict.values())
        if batch_size is None:
            batch_size = int(x_dict["note"].shape[0])
        plan = TrimPlan(self.num_layers, x_dict, edge_index_dict, neighbor_mask_node, neighbor_mask_edge)
        outs: list = []
        z, side, gnn_note = self._start_branch(x_dict["note"], batch_dict, batch_size)
        h = self.gnn(self._gnn_input(x_dict, gnn_note), edge_index_dict, plan, outs)
        out = self._finish(h["note"], outs, z, side, batch_size)
        return (out, edge_index_dict) if return_edge_index else out


def attention_roofline_case(g, I, hid: int, dev, timed, hbm_peak: float, heads: int = 4) -> dict:
    """bench.py's `roofline` object for the HGT workload (C3): the forward edge-softmax attention launch of the note
    destination type (the relations that end in `note`), same CSR and shapes as in the step, timed by `timed`.
    ALGORITHMIC bytes per launch: per relation the CSR (4 (N_dst + 1) + 4 E_r) and every distinct K', V' row once
    (2 * 4 H * N_src_unique_r: K' and V' are relation-specific), plus q and the output once (2 * 4 H * N_dst) and the
    softmax statistics kept for backward (2 * 4 * heads * N_dst)."""
    import numpy as np
    n_nodes = {k: int(v.shape[0]) for k, v in I["x_dict"].items()}
    hix = HeteroIndex(I["edge_index_dict"], n_nodes)
    ets = [et for et in hix.edge_types if et[2] == "note"]
    n = n_nodes["note"]
    gen = torch.Generator(device="cpu").manual_seed(0)
    q = torch.randn(n, hid, generator=gen).to(dev)
    kv = []
    for et in ets:
        kv += [torch.randn(n_nodes[et[0]], hid, generator=gen).to(dev), torch.randn(n_nodes[et[0]], hid, generator=gen).to(dev)]
    ps = torch.full((len(ets), heads), 1.0 / math.sqrt(hid // heads), device=dev)
    spec = _AttnSpec(fwd=[hix.fwd[et] for et in ets], bwd=[hix.bwd[et] for et in ets], n_rows=n, heads=heads,
                     n_edges=[hix.num_edges[et] for et in ets], e_limit=None, src_rows=[n_nodes[et[0]] for et in ets])
    with torch.no_grad():
        t_k, launches = timed(lambda: _HGTAttention.apply(spec, q, ps, *kv))
    b_alg, e_tot = 0, 0
    for et in ets:
        ei = g.edge_index[et]
        ei = ei[:, ei[0] >= 0]                                   # (-1, -1): padding slots of a device-sampled batch
        e_tot += int(ei.shape[1])
        b_alg += 4 * (n + 1) + 4 * int(ei.shape[1]) + 2 * 4 * hid * int(np.unique(ei[0]).size)
    b_alg += 2 * 4 * hid * n + 2 * 4 * heads * n
    return {"bound": "hbm",
            "kernel": f"k_hgt_fwd forward edge-softmax attention of the note type (R={len(ets)}, N_dst={n}, E={e_tot}, H={hid}, heads={heads})",
            "achieved": b_alg / t_k / 1e9, "peak": hbm_peak / 1e9, "unit": "GB/s", "frac": (b_alg / t_k) / hbm_peak,
            "traffic": None, "traffic_source": "profiles/r02_hgt_pmc.md (rocprofv3 --pmc passes of this launch)",
            "alg_bytes_per_launch": b_alg, "avg_us": t_k * 1e6, "launches": launches,
            "timing": "HIP events around hipGraph replays of 10 back-to-back launches"}
