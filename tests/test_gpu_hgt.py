"""HGT edge-softmax attention on the HIP kernels vs the CPU restatement of PyG HGTConv (oracle/pyg_ref.py,
oracle/encoders_ref.py).  fp32, tolerance 1e-4 relative to max(1,|ref|max) (north-star bound) for outputs,
input gradients and every parameter gradient; dropout = 0, train() mode (MIOpen RNN backward needs it)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import assert_close  # noqa: E402

TOL = 1e-4
DEV = "cuda:0"


def _cpu_params(module):
    return {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in module.state_dict().items()}


def _cmp_grads(module, P):
    n = 0
    for name, p in module.named_parameters():
        if P[name].grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None, f"{name}: no gradient on the HIP path"
        assert_close(p.grad, P[name].grad, TOL, f"grad {name}")
        n += 1
    assert n > 0


def _graph(kind):
    from analysisgnn_amd.synth import make_batch, make_score_graph, sample_hops
    if kind == "notes":
        return make_batch(2, 50, first_seed=4)
    if kind == "c3":      # note + beat + measure, 4 note-note relations + note->beat + note->measure (C3 layout)
        return make_batch(2, 60, first_seed=7, add_beats=True, add_measures=True)
    if kind == "metrical_rev":
        return make_batch(2, 60, first_seed=9, add_beats=True, add_measures=True, reverse_metrical_edges=True)
    if kind == "sampled":
        return sample_hops(make_score_graph(seed=3, n_notes=260), n_targets=48, num_neighbors=[4, 4], seed=2,
                           random_targets=True)
    raise ValueError(kind)


def _c3_edge_types(g):
    """C3 uses 6 relation types: drop beat->measure so only note-sourced relations remain."""
    return [et for et in g.edge_types if et[0] == "note"]


@pytest.mark.parametrize("kind", ["notes", "c3", "metrical_rev"])
@pytest.mark.parametrize("C,heads", [(32, 4), (64, 1), (256, 4)])
def test_hgt_conv_layer(kind, C, heads):
    from analysisgnn_amd.hgt import HGTConv
    from analysisgnn_amd.synth import torch_inputs
    from oracle import pyg_ref as G
    g = _graph(kind)
    md = g.metadata()
    torch.manual_seed(C + heads)
    m = HGTConv(C, C, md, heads)
    with torch.no_grad():
        for p in m.p_rel.values():
            p.uniform_(0.5, 1.5)
        for p in m.skip.values():
            p.uniform_(-1, 1)
    P = _cpu_params(m)
    m = m.to(DEV)
    I = torch_inputs(g, in_channels=C, seed=5)
    xc = {k: v.clone().requires_grad_(True) for k, v in I["x_dict"].items()}
    ref = G.hgt_conv(P, "", md[0], md[1], heads, xc, I["edge_index_dict"])
    xg = {k: v.to(DEV).requires_grad_(True) for k, v in I["x_dict"].items()}
    out = m(xg, {k: v.to(DEV) for k, v in I["edge_index_dict"].items()})
    assert set(out) == set(ref)
    gen = torch.Generator().manual_seed(3)
    lr = lg = 0
    for t in ref:
        assert_close(out[t], ref[t], TOL, f"out[{t}]")
        go = torch.randn(ref[t].shape, generator=gen)
        lr = lr + (ref[t] * go).sum()
        lg = lg + (out[t] * go.to(DEV)).sum()
    lr.backward()
    lg.backward()
    for t in xc:
        assert_close(xg[t].grad, xc[t].grad, TOL, f"grad x[{t}]")
    _cmp_grads(m, P)


def test_attention_weights_sum_to_one():
    """Property at the C3 shape (H=256, heads=4, 32 x 500 notes + beats + measures): with v' = 1 every destination
    row that has an incoming edge aggregates exactly 1 (sum of softmax weights), rows without edges give 0."""
    from analysisgnn_amd.graph import HeteroIndex
    from analysisgnn_amd.hgt import _AttnSpec, _HGTAttention
    from analysisgnn_amd.synth import make_batch
    g = make_batch(32, 500, add_beats=True, add_measures=True)
    ets = [et for et in g.edge_types if et[0] == "note" and et[2] == "beat"] + []
    dev = torch.device(DEV)
    eid = {et: torch.from_numpy(g.edge_index[et]).to(dev) for et in g.edge_types}
    hix = HeteroIndex(eid, g.num_nodes)
    for dst in ("note", "beat", "measure"):
        rels = [et for et in g.edge_types if et[2] == dst]
        n = g.num_nodes[dst]
        gen = torch.Generator().manual_seed(1)
        q = torch.randn(n, 256, generator=gen).to(dev)
        kv = []
        for et in rels:
            kv += [torch.randn(g.num_nodes[et[0]], 256, generator=gen).to(dev), torch.ones(g.num_nodes[et[0]], 256, device=dev)]
        spec = _AttnSpec([hix.fwd[e] for e in rels], [hix.bwd[e] for e in rels], n, 4, [hix.num_edges[e] for e in rels], None,
                         [g.num_nodes[e[0]] for e in rels])
        ps = torch.rand(len(rels), 4, generator=gen).to(dev) + 0.5
        out = _HGTAttention.apply(spec, q, ps, *kv)
        deg = torch.zeros(n, device=dev)
        for et in rels:
            deg.index_add_(0, eid[et][1], torch.ones(eid[et].shape[1], device=dev))
        has = deg > 0
        assert torch.allclose(out[has], torch.ones_like(out[has]), atol=1e-5)
        assert torch.all(out[~has] == 0)


@pytest.mark.parametrize("kind", ["c3", "sampled"])
def test_hybrid_hgt_encoder(kind):
    from analysisgnn_amd.hgt import HybridHGT
    from analysisgnn_amd.synth import torch_inputs
    from oracle import encoders_ref as E
    g = _graph(kind)
    H, L = 32, 3
    torch.manual_seed(0)
    m = HybridHGT(metadata=g.metadata(), input_channels=H, hidden_channels=H, num_layers=L, heads=4, dropout=0.0,
                  use_jk=False, logit_fusion=False).train()
    P = _cpu_params(m)
    m = m.to(DEV)
    I = torch_inputs(g, in_channels=H, seed=1)
    xc = {k: v.clone().requires_grad_(True) for k, v in I["x_dict"].items()}
    ref = E.hybrid_hgt(P, "", g.metadata(), L, 4, xc, I["edge_index_dict"], I["batch_dict"], I["batch_size"],
                       I["neighbor_mask_node"], I["neighbor_mask_edge"])
    xg = {k: v.to(DEV).requires_grad_(True) for k, v in I["x_dict"].items()}
    out = m(x_dict=xg, edge_index_dict={k: v.to(DEV) for k, v in I["edge_index_dict"].items()},
            batch_dict={k: v.to(DEV) for k, v in I["batch_dict"].items()}, batch_size=I["batch_size"],
            neighbor_mask_node=I["neighbor_mask_node"], neighbor_mask_edge=I["neighbor_mask_edge"])
    assert_close(out, ref, TOL, "out")
    gout = torch.randn(ref.shape, generator=torch.Generator().manual_seed(9))
    (ref * gout).sum().backward()
    (out * gout.to(DEV)).sum().backward()
    for k in xc:
        if xc[k].grad is not None:
            assert_close(xg[k].grad, xc[k].grad, TOL, f"grad x[{k}]")
    _cmp_grads(m, P)


def test_analysis_model_hgt_logits():
    from analysisgnn_amd.models import TorchAnalysisGNN
    from analysisgnn_amd.synth import make_batch, torch_inputs
    from oracle import encoders_ref as E
    g = make_batch(2, 64, first_seed=13, add_beats=True, add_measures=True)
    tasks = {"cadence": 4, "localkey": 50, "romanNumeral": 185}
    torch.manual_seed(3)
    m = TorchAnalysisGNN(g.metadata(), in_channels=25, hidden_channels=32, out_channels=16, task_dict=tasks, num_layers=3,
                         dropout=0.0, use_jk=False, logit_fusion=False, encoder_type="hgt").train()
    P = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(DEV)
    I = torch_inputs(g, in_channels=25, seed=4)
    with torch.no_grad():
        x = E.analysis_encode(P, "hgt", g.metadata(), 3, I["pitch_spelling"], I["key_signature"], I["x_dict"],
                              I["edge_index_dict"], I["batch_dict"], I["batch_size"])
        ref = E.analysis_logits(P, x, list(tasks))
        out = m(I["pitch_spelling"].to(DEV), I["key_signature"].to(DEV), {k: v.to(DEV) for k, v in I["x_dict"].items()},
                {k: v.to(DEV) for k, v in I["edge_index_dict"].items()}, {k: v.to(DEV) for k, v in I["batch_dict"].items()},
                I["batch_size"], None, None)
    for t in tasks:
        assert_close(out[t], ref[t], TOL, f"logits[{t}]")


@pytest.mark.parametrize("N,R,heads,T", [(16000, 6, 4, 6), (1003, 2, 4, 7), (130, 1, 1, 3), (5, 3, 2, 3)])
def test_relation_transform_kernels_match_float64(N, R, heads, T):
    """agnn_relt_fwd / _bwd / _dw (the per-head D x D relation transforms of HGTConv, D = 64) against per-block float64
    matmuls: K and V in one launch, relations picked out of a larger parameter (`rel_ids`), rows not a multiple of the
    128-row tile, gradients w.r.t. both operands and both parameters."""
    from analysisgnn_amd.hgt import _RelTransform
    D, H = 64, heads * 64
    g = torch.Generator().manual_seed(N + R)
    rel_ids = tuple(sorted(torch.randperm(T, generator=g)[:R].tolist()))
    kqv = torch.randn(N, 3 * H, generator=g)                       # K | Q | V side by side: the operands are column views
    wk = torch.randn(T * heads, D, D, generator=g) * 0.2
    wv = torch.randn(T * heads, D, D, generator=g) * 0.2
    gk, gv = torch.randn(N, R * H, generator=g), torch.randn(N, R * H, generator=g)
    kqv64, wk64, wv64 = (t.double().requires_grad_(True) for t in (kqv, wk, wv))

    def ref_of(x, w):
        blocks = []
        for r in rel_ids:
            for h in range(heads):
                blocks.append(x[:, h * D:(h + 1) * D] @ w[r * heads + h])
        return torch.cat(blocks, dim=1)
    rk, rv = ref_of(kqv64[:, :H], wk64), ref_of(kqv64[:, 2 * H:], wv64)
    ((rk * gk.double()).sum() + (rv * gv.double()).sum()).backward()
    kd = kqv.to(DEV).requires_grad_(True)
    wkd, wvd = wk.to(DEV).requires_grad_(True), wv.to(DEV).requires_grad_(True)
    yk, yv = _RelTransform.apply(kd[:, :H], kd[:, 2 * H:], wkd, wvd, rel_ids, heads, D)
    ((yk * gk.to(DEV)).sum() + (yv * gv.to(DEV)).sum()).backward()
    assert_close(yk, rk.float(), 1e-5, "k'")
    assert_close(yv, rv.float(), 1e-5, "v'")
    assert_close(kd.grad, kqv64.grad.float(), 1e-5, "d kqv")
    assert_close(wkd.grad, wk64.grad.float(), 1e-5, "d k_rel.weight")
    assert_close(wvd.grad, wv64.grad.float(), 1e-5, "d v_rel.weight")


@pytest.mark.parametrize("n,H,relu,p,with_x", [(16000, 256, True, 0.0, True), (3584, 256, True, 0.3, True), (897, 64, False, 0.0, True),
                                                (50, 256, True, 0.0, False)])
def test_skip_act_epilogue_matches_torch(n, H, relu, p, with_x):
    """agnn_skip_act_*: z = dropout(relu(lerp(x, o, sigmoid(skip)))) in one launch each way against the torch ops it replaces
    (float64): values, d o, d x and the scalar d skip (a deterministic ticket reduction).  With dropout the mask is the library's
    own: checked through its invariants (kept elements = value / (1 - p), dropped = 0, rate ~ p) and by replaying the backward
    through the SAME mask (read off the forward result)."""
    from analysisgnn_amd.fused import skip_act
    from helpers import assert_close_rel
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n + H)
    o = torch.randn(n, H, generator=g)
    x = torch.randn(n, H, generator=g) if with_x else None
    skip = torch.tensor([0.3]) if with_x else None
    od = o.to(dev).requires_grad_(True)
    xd = x.to(dev).requires_grad_(True) if with_x else None
    sd = skip.to(dev).requires_grad_(True) if with_x else None
    z = skip_act(od, xd, sd, relu, p, True)
    gz = torch.randn(n, H, generator=g)
    z.backward(gz.to(dev))
    o64 = o.double().requires_grad_(True)
    x64 = x.double().requires_grad_(True) if with_x else None
    s64 = skip.double().requires_grad_(True) if with_x else None
    y = o64 if not with_x else torch.lerp(x64, o64, torch.sigmoid(s64))
    y = torch.relu(y) if relu else y
    if p > 0:
        zc = z.detach().cpu().double()
        keep = (zc != 0) | (y.detach() == 0)                       # where the forward kept the element (exact zeros of y: either way)
        rate = 1.0 - float(((zc != 0).sum()) / max(int((y.detach() != 0).sum()), 1))
        assert abs(rate - p) < 0.02, rate
        y = y * keep.double() / (1.0 - p)
    assert_close_rel(z, y.detach(), 1e-6, "z")
    y.backward(gz.double())
    assert_close_rel(od.grad, o64.grad, 1e-6, "d o")
    if with_x:
        assert_close_rel(xd.grad, x64.grad, 1e-6, "d x")
        assert_close_rel(sd.grad, s64.grad, 1e-5, "d skip")


def test_destination_types_in_one_attention_launch_give_the_same_layer():
    """agnn_hgt_attn_fwd_multi_f32 (every destination type's edge softmax in one launch) against one launch per type: the layer's
    outputs and input gradients bit for bit (the same rows by the same code), on the graph with note / beat / measure destinations."""
    from analysisgnn_amd import hgt
    from analysisgnn_amd.hgt import HGTConv
    from analysisgnn_amd.synth import torch_inputs
    g = _graph("metrical_rev")
    torch.manual_seed(3)
    m = HGTConv(256, 256, g.metadata(), 4).to(DEV)
    I = torch_inputs(g, in_channels=256, seed=2)
    ei = {k: v.to(DEV) for k, v in I["edge_index_dict"].items()}
    saved = hgt.ATTN_ONE_LAUNCH
    res = []
    try:
        for one in (True, False):
            hgt.ATTN_ONE_LAUNCH = one
            xg = {k: v.to(DEV).requires_grad_(True) for k, v in I["x_dict"].items()}
            out = m(xg, ei)
            sum(o.square().sum() for o in out.values()).backward()
            res.append(({k: o.detach() for k, o in out.items()}, {k: x.grad for k, x in xg.items()}))
    finally:
        hgt.ATTN_ONE_LAUNCH = saved
    assert len(res[0][0]) >= 3
    for k in res[0][0]:
        assert torch.equal(res[0][0][k], res[1][0][k]), k
        assert torch.equal(res[0][1][k], res[1][1][k]), k
