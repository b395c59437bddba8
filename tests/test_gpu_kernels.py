"""Kernel-level parity on a real MI355X: HIP C-ABI (libagnn_hip.so) vs the plain-C oracle.
Integer outputs (CSR) must be identical; fp32 sums follow the same operation order as the oracle
(fmaf in CSR order); the generic kernel divides (bitwise equal), the H=256/512 fast path multiplies by
v_rcp_f32(count) (<= 2 ulp).  Tolerance 1e-6 relative."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import c_oracle  # noqa: E402


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _build(segs_np):
    from analysisgnn_amd.graph import SegSpec, build_csr
    dev = _dev()
    specs = []
    for s in segs_np:
        specs.append(SegSpec(row=torch.from_numpy(s["row"]).to(dev), col=torch.from_numpy(s["col"]).to(dev),
                             n_rows=s["n_rows"],
                             etype=(torch.from_numpy(s["etype"]).to(dev) if s.get("etype") is not None else None),
                             code=s.get("code", 0)))
    return build_csr(specs)


def _check_csr(segs_np):
    csrs = _build(segs_np)
    rs, c, p, kept = c_oracle.csr_build(segs_np)
    base = 0
    for s, csr in zip(segs_np, csrs):
        n = s["n_rows"]
        rp = csr.rowptr.cpu().numpy()
        assert np.array_equal(rp, rs[base:base + n + 1])
        base += n
    assert np.array_equal(csrs[0].col.cpu().numpy()[:kept], c[:kept])
    assert np.array_equal(csrs[0].perm.cpu().numpy()[:kept], p[:kept])
    return csrs


@pytest.mark.parametrize("n,e", [(1, 0), (5, 0), (7, 1), (64, 1000), (500, 3305), (16000, 105760)])
def test_csr_build_single(n, e):
    rng = np.random.default_rng(n + e)
    _check_csr([dict(row=rng.integers(0, n, size=e).astype(np.int64),
                     col=rng.integers(0, n, size=e).astype(np.int64), n_rows=n)])


def test_csr_build_many_segments_with_type_mask_and_empty_relation():
    rng = np.random.default_rng(5)
    n, e = 300, 5000
    row = rng.integers(0, n, size=e).astype(np.int64)
    col = rng.integers(0, n, size=e).astype(np.int64)
    et = rng.integers(0, 5, size=e).astype(np.int64)
    segs = [dict(row=row, col=col, n_rows=n, etype=et, code=k) for k in range(7)]      # codes 5,6 empty
    segs += [dict(row=col, col=row, n_rows=n, etype=et, code=k) for k in range(7)]
    _check_csr(segs)


def test_csr_build_skewed_rows_and_out_of_range():
    rng = np.random.default_rng(6)
    n, e = 50, 4000
    row = np.where(rng.random(e) < 0.7, 3, rng.integers(0, n, size=e)).astype(np.int64)   # one heavy row
    row[:5] = n + 10                                                                      # dropped
    col = rng.integers(0, n, size=e).astype(np.int64)
    _check_csr([dict(row=row, col=col, n_rows=n)])


def test_rowend_matches_oracle():
    rng = np.random.default_rng(7)
    n, e = 200, 3000
    seg = dict(row=rng.integers(0, n, size=e).astype(np.int64), col=rng.integers(0, n, size=e).astype(np.int64), n_rows=n)
    csr = _build([seg])[0]
    rs, c, p, _ = c_oracle.csr_build([seg])
    for lim in (0, 1, 1234, e - 1):
        got = csr.rowend(lim).cpu().numpy()[:n]
        assert np.array_equal(got, c_oracle.csr_rowend(rs, p, lim))
    assert csr.rowend(e) is None and csr.rowend(None) is None


def _spmm_case(n_dst, n_src, es, H, mean, shared, with_self, skip_self=False, col_limit=None, trim=None, seed=0,
               colscale=False, root_rows=None):
    from analysisgnn_amd import _lib, ops
    dev = _dev()
    rng = np.random.default_rng(seed)
    R = len(es)
    segs = [dict(row=rng.integers(0, n_dst, size=e).astype(np.int64),
                 col=rng.integers(0, n_src, size=e).astype(np.int64), n_rows=n_dst) for e in es]
    csrs = _build(segs)
    rs, c, p, kept = c_oracle.csr_build(segs)
    srcs = [rng.standard_normal((n_src, H)).astype(np.float32) for _ in range(R)]
    x = rng.standard_normal((n_dst, H)).astype(np.float32) if with_self else None
    cs = [rng.random(n_src).astype(np.float32) + 0.5 for _ in range(R)] if colscale else [None] * R
    n_rows = n_dst
    orels, grels, keep = [], [], []
    base = 0
    for r in range(R):
        rp = rs[base:base + n_dst + 1]
        base += n_dst
        re_o = c_oracle.csr_rowend(rp, p, trim[r]) if trim else None
        orels.append(dict(src=srcs[r], rowptr=rp, col=c, rowend=re_o, colscale=cs[r]))
        st = torch.from_numpy(srcs[r]).to(dev)
        re_g = csrs[r].rowend(trim[r]) if trim else None
        cst = torch.from_numpy(cs[r]).to(dev) if colscale else None
        keep += [st, re_g, cst]
        grels.append(dict(src=st.data_ptr(), rowptr=csrs[r].rowptr.data_ptr(), rowend=_lib.ptr(re_g),
                          col=csrs[r].col.data_ptr(), colscale=_lib.ptr(cst), ld_src=st.stride(0)))
    lim = col_limit if col_limit is not None else 2 ** 31 - 1
    exp, inv_o = c_oracle.spmm(orels, n_rows, H, 0 if shared else H, self_=None if root_rows is not None else x, mean=mean,
                               skip_self=skip_self, col_limit=lim, want_inv_cnt=True)
    if root_rows is not None:
        # agnn_spmm_root_f32: the root operand is an extra slot (forward) / one more addend for the rows it has (backward)
        if shared:
            exp = exp.copy()
            exp[:root_rows] += x[:root_rows]
        else:
            exp = np.concatenate([exp, x[:n_rows]], axis=1)
    out = torch.full((n_rows, H if shared else (R + (root_rows is not None)) * H), float("nan"), device=dev)
    inv = torch.empty((R, n_rows), device=dev)
    xt = torch.from_numpy(x).to(dev) if with_self else None
    if root_rows is not None and shared:
        xt = xt[:root_rows].contiguous()         # the slot really is shorter than the output
    flags = (_lib.SPMM_MEAN if mean else 0) | (_lib.SPMM_SKIP_SELF if skip_self else 0)
    ops._launch(grels, n_rows, H, out, 0 if shared else H, xt, inv, lim, flags, root_rows=root_rows)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert not np.isnan(got).any()
    np.testing.assert_allclose(got, exp, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(inv.cpu().numpy(), inv_o, rtol=2e-7, atol=0)     # fast path: v_rcp_f32 (<= 1 ulp)
    return float(np.abs(got - exp).max())


@pytest.mark.parametrize("H", [4, 8, 64, 256, 260, 512, 1024])
def test_spmm_widths(H):
    _spmm_case(97, 97, [400, 150], H, mean=True, shared=False, with_self=True, seed=H)


@pytest.mark.parametrize("mean", [True, False])
@pytest.mark.parametrize("shared", [True, False])
@pytest.mark.parametrize("with_self", [True, False])
def test_spmm_modes(mean, shared, with_self):
    _spmm_case(300, 211, [1200, 0, 37, 900], 32, mean, shared, with_self, seed=11, colscale=shared)


def test_spmm_heavy_row_and_empty_rows():
    # one destination with > 64 neighbours (several col batches), many rows without any edge
    from analysisgnn_amd import _lib
    _spmm_case(40, 500, [3000], 256, mean=True, shared=False, with_self=False, seed=3)
    # the same through the backward configuration of the fast path (column scales, relations summed into one slot)
    _spmm_case(40, 500, [3000, 70], 256, mean=False, shared=True, with_self=False, seed=4, colscale=True)
    _spmm_case(33, 500, [2500, 0, 9], 512, mean=True, shared=False, with_self=True, seed=5)


def test_spmm_filters_and_trim():
    _spmm_case(120, 120, [900], 16, mean=True, shared=True, with_self=True, skip_self=True, col_limit=70, seed=5)
    _spmm_case(120, 120, [900, 400], 16, mean=True, shared=False, with_self=False, trim=[300, 0], seed=6)


def test_fast_and_generic_kernels_agree():
    """Same launch through the specialised fast path and through the generic kernel (flag AGNN_SPMM_GENERIC); both are
    compared with the C oracle inside _spmm_case."""
    from analysisgnn_amd import ops
    for H in (256, 512):
        for kw in (dict(mean=True, shared=False, with_self=False), dict(mean=True, shared=False, with_self=True),
                   dict(mean=False, shared=True, with_self=False, colscale=True)):
            ops.SPMM_VARIANT = 0
            _spmm_case(300, 300, [1500, 2, 700, 0, 90], H, seed=H, **kw)
            ops.SPMM_VARIANT = 1024
            try:
                _spmm_case(300, 300, [1500, 2, 700, 0, 90], H, seed=H, **kw)
            finally:
                ops.SPMM_VARIANT = 0


@pytest.mark.parametrize("H", [256, 512])
@pytest.mark.parametrize("variant", [0, 1024])
def test_fast_path_trimmed_and_filtered(H, variant):
    """What a training step of the reference launches (models/analysis.py:960-961 always passes the per-hop counts, so
    every layer is trimmed): row ends from `rowend` (forward), `rowend` + column limit + 1/deg column scales (backward),
    and the onset pooling's predicates (models/analysis.py:581-584) — through the fast kernel (k_spmm_fast7, FILT and
    plain) and through the generic one, each against the C oracle."""
    from analysisgnn_amd import ops
    ops.SPMM_VARIANT = variant
    try:
        # forward of a trimmed SAGE layer: per-relation COO prefixes, one relation untrimmed, one trimmed to nothing
        _spmm_case(300, 300, [1500, 40, 700, 0, 90], H, mean=True, shared=False, with_self=False,
                   trim=[900, 40, 1, 0, 0], seed=H + 1)
        # its backward: relations summed into one slot, 1/deg scales, trimmed rows and the column limit
        _spmm_case(300, 300, [1500, 40, 700, 0, 90], H, mean=False, shared=True, with_self=False, colscale=True,
                   trim=[900, 40, 1, 0, 0], col_limit=211, seed=H + 2)
        # onset pooling forward (self numerator, self loops skipped, column limit) and backward
        _spmm_case(300, 300, [1400], H, mean=True, shared=True, with_self=True, skip_self=True, col_limit=170, seed=H + 3)
        _spmm_case(300, 300, [1400], H, mean=False, shared=True, with_self=False, skip_self=True, col_limit=170,
                   colscale=True, seed=H + 4)
        # rows with more than 64 neighbours (ids beyond the registers: scalar-load loop) with both predicates and a trim
        _spmm_case(40, 60, [3000, 70], H, mean=True, shared=False, with_self=True, skip_self=True, col_limit=41,
                   trim=[2500, 70], seed=H + 5)
        _spmm_case(40, 500, [3000], H, mean=True, shared=False, with_self=False, trim=[2999], seed=H + 6)
    finally:
        ops.SPMM_VARIANT = 0


@pytest.mark.parametrize("H", [256, 512])
def test_root_operand_rides_along(H):
    """agnn_spmm_root_f32: forward = the relations' slots plus the root rows as one more slot; backward = the transposed
    aggregation plus the root slot's gradient for the first root_rows rows (trimmed, column-limited, 1/deg scales)."""
    _spmm_case(300, 300, [1500, 40, 700, 0, 90], H, mean=True, shared=False, with_self=True, trim=[900, 40, 1, 0, 0],
               seed=H + 11, root_rows=300)
    _spmm_case(300, 300, [1500, 40, 700, 0, 90], H, mean=False, shared=True, with_self=True, colscale=True,
               trim=[900, 40, 1, 0, 0], col_limit=211, seed=H + 12, root_rows=211)
    _spmm_case(64, 64, [200], H, mean=False, shared=True, with_self=True, colscale=True, seed=H + 13, root_rows=64)
    _spmm_case(40, 500, [3000], H, mean=True, shared=False, with_self=True, seed=H + 14, root_rows=40)


def test_root_operand_rejects_what_the_kernel_does_not_cover():
    from analysisgnn_amd import _lib
    with pytest.raises(_lib.AgnnError):
        _spmm_case(30, 30, [100], 64, mean=True, shared=False, with_self=True, root_rows=30)


def test_spmm_c2_shape_against_oracle():
    """BASELINE config C2 shape for one layer (32 x 500 notes, 4 relations, H=256) vs the C oracle."""
    from analysisgnn_amd.synth import make_batch
    from analysisgnn_amd.graph import HeteroIndex
    from analysisgnn_amd import ops
    dev = _dev()
    b = make_batch(32, 500)
    N = b.num_nodes["note"]
    eid = {et: torch.from_numpy(e).to(dev) for et, e in b.edge_index.items()}
    hix = HeteroIndex(eid, {"note": N})
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, 256, generator=g)
    xd = x.to(dev).requires_grad_(True)
    ets = list(eid.keys())
    spec = ops.AggSpec(fwd=[hix.fwd[et] for et in ets], bwd=[hix.bwd[et] for et in ets], src_id=[0] * 4, n_rows=N,
                       mean=True, shared_slot=False)
    out = ops.aggregate(spec, [xd])
    segs = [dict(row=b.edge_index[et][1], col=b.edge_index[et][0], n_rows=N) for et in ets]
    rs, c, p, _ = c_oracle.csr_build(segs)
    exp = c_oracle.spmm([dict(src=x.numpy(), rowptr=rs[r * N:(r + 1) * N + 1], col=c) for r in range(4)], N, 256, 256)
    np.testing.assert_allclose(out.detach().cpu().numpy(), exp, rtol=1e-6, atol=1e-6)
    # backward = transposed gather with 1/deg(dst) weights; property: <A x, g> == <x, A^T g>
    gout = torch.randn(N, 1024, generator=g).to(dev)
    out.backward(gout)
    lhs = float((out.detach().double() * gout.double()).sum())
    rhs = float((xd.detach().double() * xd.grad.double()).sum())
    assert abs(lhs - rhs) <= 1e-6 * max(1.0, abs(lhs))


def test_pack_items_sum_and_strided_pieces():
    """agnn_pack_f32 through analysisgnn_amd.params.pack: strided destinations, several sources, odd sizes, > 24 items."""
    from analysisgnn_amd.params import pack
    torch.manual_seed(0)
    dev = "cuda:0"
    big = torch.zeros(37, 4 * 20 + 3, device=dev)
    srcs = [torch.randn(37, 20, device=dev) for _ in range(4)]
    items = [(big[:, 20 * r:20 * (r + 1)], [srcs[r]]) for r in range(4)]            # cat along columns (ld_dst > cols)
    acc = torch.empty(37, 20, device=dev)
    items.append((acc, srcs))                                                        # sum of four
    vec = torch.empty(1, 131, device=dev)
    vs = [torch.randn(1, 131, device=dev) for _ in range(3)]
    items.append((vec, vs))                                                          # odd width: scalar path
    many_dst = [torch.empty(5, 8, device=dev) for _ in range(30)]
    many_src = [torch.randn(5, 8, device=dev) for _ in range(30)]
    items += [(d, [s]) for d, s in zip(many_dst, many_src)]                          # more than one launch worth of items
    pack(items, torch.device(dev))
    torch.cuda.synchronize()
    assert torch.equal(big[:, :80], torch.cat(srcs, dim=1)) and float(big[:, 80:].abs().max()) == 0.0
    assert torch.allclose(acc, srcs[0] + srcs[1] + srcs[2] + srcs[3], rtol=0, atol=1e-6)
    assert torch.allclose(vec, vs[0] + vs[1] + vs[2], rtol=0, atol=1e-6)
    assert all(torch.equal(d, s) for d, s in zip(many_dst, many_src))


def test_multitask_ce_uncovered_columns_and_ignored_rows():
    """Segments that do not cover every column (zero-filled gradient there), all-ignored task, C > 192 (strided path)."""
    from analysisgnn_amd.heads import multitask_cross_entropy
    import torch.nn.functional as F
    torch.manual_seed(1)
    dev = "cuda:0"
    N = 203
    offs = [2, 7, 7 + 200, 7 + 200 + 3]                   # columns 0-1 and the tail belong to no task
    width = offs[-1] + 4
    logits = torch.randn(N, width, device=dev, requires_grad=True)
    labels = torch.stack([torch.randint(0, 5, (N,)), torch.randint(0, 200, (N,)), torch.full((N,), -1)]).to(dev)
    labels[0, ::3] = -1
    loss = multitask_cross_entropy(logits, offs, labels, 0.1, -1)
    (loss * torch.tensor([1.0, 2.0, 3.0], device=dev)).sum().backward()
    ref_in = logits.detach().double().cpu().requires_grad_(True)
    ref = []
    for t in range(3):
        seg = ref_in[:, offs[t]:offs[t + 1]]
        lab = labels[t].cpu()
        ref.append(F.cross_entropy(seg, lab, ignore_index=-1, label_smoothing=0.1) if (lab != -1).any() else seg.sum() * 0.0)
    ref = torch.stack(ref)
    (ref * torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64)).sum().backward()
    assert torch.allclose(loss.cpu().double(), ref.detach(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(logits.grad.cpu().double(), ref_in.grad, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("M,N,K,bias", [(300, 128, 16, True), (1000, 256, 160, False), (16335, 256, 1280, True), (129, 384, 48, True)])
def test_gemm_nt_matches_float64(M, N, K, bias):
    """agnn_gemm_nt_f32 (hand-written fp32 MFMA projection GEMM: C = A W^T + b) against float64, incl. a row count that is not
    a multiple of the 128-row tile, strided operands and the C2 SAGE-layer shape.  Exact-fp32 MFMA: ~1e-6 relative."""
    from analysisgnn_amd import _lib
    lib = _lib.load()
    DEV = "cuda:0"
    g = torch.Generator().manual_seed(M + N + K)
    a_full = torch.randn(M, K + 8, generator=g).to(DEV)
    a = a_full[:, :K]                                            # leading dimension > K
    w = (torch.randn(N, K, generator=g) * 0.1).to(DEV)
    b = torch.randn(N, generator=g).to(DEV) if bias else None
    c = torch.full((M, N + 4), 7.0, device=DEV)
    _lib.check(lib.agnn_gemm_nt_f32(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), _lib.ptr(b), M, N, K, c.data_ptr(), c.stride(0),
                                    _lib.stream_ptr(torch.device(DEV))), "agnn_gemm_nt_f32")
    ref = a.double() @ w.double().t() + (b.double() if bias else 0.0)
    err = float((c[:, :N].double() - ref).abs().max() / ref.abs().max())
    assert err < 5e-6, err
    assert float((c[:, N:] - 7.0).abs().max()) == 0.0            # nothing written past N
    assert lib.agnn_gemm_nt_f32(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), None, M, N + 1, K, c.data_ptr(), c.stride(0), None) < 0


@pytest.mark.parametrize("M,N,K", [(300, 128, 16), (1000, 256, 160), (16335, 1280, 256), (129, 384, 48), (5000, 64, 80), (16000, 128, 1344)])
def test_gemm_nn_matches_float64(M, N, K):
    """agnn_gemm_nn_f32: C = A w with the second operand K-major (the input-gradient product dX = dY W on the weight as it lies):
    against float64, strided operands, row counts that are not tile multiples, odd K-step counts, both tile widths."""
    from analysisgnn_amd import _lib
    lib = _lib.load()
    DEV = "cuda:0"
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K + 8, generator=g).to(DEV)[:, :K]
    w = (torch.randn(K, N + 4, generator=g) * 0.1).to(DEV)[:, :N]
    c = torch.full((M, N + 4), 7.0, device=DEV)
    _lib.check(lib.agnn_gemm_nn_f32(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), None, M, N, K, c.data_ptr(), c.stride(0),
                                    _lib.stream_ptr(torch.device(DEV))), "agnn_gemm_nn_f32")
    ref = a.double() @ w.double()
    assert float((c[:, :N].double() - ref).abs().max() / ref.abs().max()) < 5e-6
    assert float((c[:, N:] - 7.0).abs().max()) == 0.0
    assert lib.agnn_gemm_nn_f32(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), None, M, N, K + 1, c.data_ptr(), c.stride(0), None) < 0


@pytest.mark.parametrize("M,N,K,bias", [(5000, 256, 256, True), (4096, 64, 16, False), (16000, 1344, 128, True), (6001, 512, 2048, True), (4500, 192, 80, True)])
def test_linear_forward_on_the_hand_written_gemm(M, N, K, bias):
    """linear.linear() with HAND_GEMM (the default): forward on agnn_gemm_nt_f32 — 64-wide column tiles when 128-wide ones would
    leave fewer than two workgroups per CU, odd K-step counts (K = 80: five steps of 16), a strided input — against float64 and against
    the library path; the input gradient runs on agnn_gemm_nn_f32 (library with the switch off), dW on the MFMA weight-gradient kernel either way."""
    from analysisgnn_amd import linear as L
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(M + N + K)
    x_full = torch.randn(M, K + 4, generator=g).to(dev)
    x = x_full[:, :K].requires_grad_(True)
    w = (torch.randn(N, K, generator=g) * 0.1).to(dev).requires_grad_(True)
    b = torch.randn(N, generator=g).to(dev).requires_grad_(True) if bias else None
    saved = (L.HAND_GEMM, L.HAND_GEMM_DX, L.HAND_GEMM_MAX_K, L.HAND_GEMM_MIN_K, L.HAND_GEMM_MIN_N)
    L.HAND_GEMM_MAX_K, L.HAND_GEMM_MIN_K, L.HAND_GEMM_MIN_N = 1 << 20, 16, 64          # every shape the kernel takes, not only the ones the step sends it
    assert L.HAND_GEMM and L._hand_gemm_ok(x, w, b)
    try:
        outs = []
        for hand in (True, False):
            L.HAND_GEMM = L.HAND_GEMM_DX = hand
            for t in (x, w, b):
                if t is not None:
                    t.grad = None
            y = L.linear(x, w, b)
            y.sum().backward()
            outs.append((y.detach(), x.grad.clone(), w.grad.clone()))
    finally:
        L.HAND_GEMM, L.HAND_GEMM_DX, L.HAND_GEMM_MAX_K, L.HAND_GEMM_MIN_K, L.HAND_GEMM_MIN_N = saved
    ref = x.detach().double() @ w.detach().double().t() + (b.detach().double() if bias else 0.0)
    scale = float(ref.abs().max())
    assert float((outs[0][0].double() - ref).abs().max()) / scale < 5e-6
    assert float((outs[1][0].double() - ref).abs().max()) / scale < 5e-6
    dx_ref = torch.ones(M, N, dtype=torch.float64, device=dev) @ w.detach().double()        # d(sum y) / dx
    for o in outs:
        assert float((o[1].double() - dx_ref).abs().max()) / float(dx_ref.abs().max()) < 5e-6
    assert torch.equal(outs[0][2], outs[1][2])                 # dW: the MFMA weight-gradient kernel either way


def test_linear_falls_back_to_the_library_for_operands_the_hand_written_gemm_refuses():
    """A broadcast (stride-0) input, an unaligned view, K not a multiple of 16: `linear` must take the library path (same result as
    torch), not hand the kernel an operand it rejects."""
    from analysisgnn_amd import linear as L
    dev = torch.device("cuda", 0)
    w = (torch.randn(256, 256, device=dev) * 0.1)
    b = torch.randn(256, device=dev)
    row = torch.randn(1, 256, device=dev)
    for x in (row.expand(5000, 256), torch.randn(5000, 257, device=dev)[:, 1:], torch.randn(5000, 256, device=dev)):
        y = L.linear(x, w, b)
        ref = torch.addmm(b, x, w.t())
        assert float((y - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    x = torch.randn(5000, 250, device=dev)
    assert torch.equal(L.linear(x, w[:, :250].contiguous(), b), torch.addmm(b, x, w[:, :250].contiguous().t()))


def test_pack_zero_fills_pieces_without_sources():
    """agnn_pack_f32: an item with n_src = 0 clears its destination piece (strided column blocks included) and leaves the rest alone."""
    from analysisgnn_amd.params import pack
    dev = torch.device("cuda", 0)
    a = torch.full((37, 48), 7.0, device=dev)
    b = torch.full((5, 12), 3.0, device=dev)
    src = torch.arange(37 * 16, dtype=torch.float32, device=dev).view(37, 16)
    pack([(a[:, 8:24], []), (b[2:4], []), (a[:, 32:48], [src, src])], dev)
    want = torch.full((37, 48), 7.0, device=dev)
    want[:, 8:24] = 0
    want[:, 32:48] = 2 * src
    assert torch.equal(a, want)
    wb = torch.full((5, 12), 3.0, device=dev)
    wb[2:4] = 0
    assert torch.equal(b, wb)
