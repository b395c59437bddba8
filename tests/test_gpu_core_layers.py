"""HIP-backed mirrors of the reference's in-tree layers (analysisgnn_amd/core_layers.py) against the
golden vectors produced by RUNNING the reference's own core/gnn.py + core/hgnn.py
(oracle/gen_golden.py).  fp32; tolerance 1e-4 relative to max(1,|ref|max) (north-star bound), the
observed error is ~1e-6."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import assert_close, load_golden  # noqa: E402

TOL = 1e-4


def _load(module, z, dev):
    sd = {k[2:]: torch.from_numpy(np.asarray(z[k])) for k in z.files if k.startswith("w.")}
    module.load_state_dict(sd, strict=True)
    return module.to(dev)


def _inputs(z, dev, grad=("x",)):
    I = {}
    for k in z.files:
        if k.startswith("in."):
            t = torch.from_numpy(np.asarray(z[k])).to(dev)
            if k[3:] in grad:
                t.requires_grad_(True)
            I[k[3:]] = t
    return I


def _check(z, m, out, I, grad_keys=("x",)):
    assert_close(out, z["out"], TOL, "out")
    (out * torch.from_numpy(z["gout"]).to(out.device)).sum().backward()
    for k in grad_keys:
        assert_close(I[k].grad, z[f"grad.{k}"], TOL, f"grad.{k}")
    for n, p in m.named_parameters():
        assert p.grad is not None, n
        assert_close(p.grad, z[f"gw.{n}"], TOL, f"gw.{n}")


@pytest.mark.parametrize("name", ["sage_small", "sage_empty", "sage_small_edgefeat"])
def test_sage_conv_scatter(name):
    from analysisgnn_amd.core_layers import SageConvScatter
    dev = torch.device("cuda:0")
    z = load_golden(name)
    ef = 3 if "in.edge_features" in z.files else None
    m = _load(SageConvScatter(4, 6, in_edge_features=ef), z, dev)
    gk = ("x", "edge_features") if ef else ("x",)
    I = _inputs(z, dev, gk)
    out = m(I["x"], I["edge_index"], I.get("edge_features"))
    _check(z, m, out, I, gk)


def test_sage_big_h256():
    from analysisgnn_amd.core_layers import SageConvScatter
    from analysisgnn_amd.synth import make_score_graph
    from oracle.testing import checksum, seeded_fill_, seeded_randn
    dev = torch.device("cuda:0")
    z = load_golden("sage_big")
    g0 = make_score_graph(seed=int(z["meta.seed_graph"]), n_notes=500)
    m = SageConvScatter(256, 256)
    seeded_fill_(m, int(z["meta.seed_w"]))
    m = m.to(dev)
    x = seeded_randn(int(z["meta.seed_x"]), 500, 256).to(dev).requires_grad_(True)
    out = m(x, torch.from_numpy(g0.edge_index[("note", "onset", "note")]).to(dev))
    assert_close(out[:16], z["out.head"], TOL, "out.head")
    (out * seeded_randn(12345, 500, 256).to(dev)).sum().backward()
    assert_close(x.grad[:16], z["grad.x.head"], TOL, "grad.x.head")
    np.testing.assert_allclose(checksum(out), z["out.sum"], rtol=1e-4, atol=5e-2)
    np.testing.assert_allclose(checksum(x.grad), z["grad.x.sum"], rtol=1e-4, atol=5e-2)
    for n, p in m.named_parameters():
        np.testing.assert_allclose(checksum(p.grad), z[f"gw.{n}.sum"], rtol=2e-4, atol=1e-1)


@pytest.mark.parametrize("red", ["mean", "sum"])
def test_hetero_sage_layer(red):
    from analysisgnn_amd.core_layers import HeteroSageConvLayer
    dev = torch.device("cuda:0")
    z = load_golden(f"hsage_{red}")
    rels = [str(r) for r in z["meta.rels"]]
    m = _load(HeteroSageConvLayer(8, 8, etypes={r: i for i, r in enumerate(rels)}, reduction=red), z, dev)
    I = _inputs(z, dev)
    out = m(I["x"], I["edge_index"], I["edge_type"])
    _check(z, m, out, I)
    if red == "sum":          # dict input form (core/hgnn.py:130-133)
        zd = load_golden("hsage_sum_dictform")
        eid = {r: I["edge_index"][:, I["edge_type"] == c] for c, r in enumerate(rels)}
        assert_close(m(I["x"].detach(), eid), zd["out"], TOL, "dict form")


@pytest.mark.parametrize("jk", [False, True])
def test_hgcn(jk):
    from analysisgnn_amd.core_layers import HGCN
    dev = torch.device("cuda:0")
    z = load_golden("hgcn3_jk" if jk else "hgcn3")
    rels = [str(r) for r in z["meta.rels"]]
    m = _load(HGCN(8, 16, 8, n_layers=2, etypes={r: i for i, r in enumerate(rels)}, dropout=0.0, jk=jk), z, dev)
    I = _inputs(z, dev)
    out = m(I["x"], I["edge_index"], I["edge_type"])
    _check(z, m, out, I)


@pytest.mark.parametrize("name", ["resgated", "resgated_edgefeat"])
def test_res_gated(name):
    from analysisgnn_amd.core_layers import ResGatedGraphConv
    dev = torch.device("cuda:0")
    z = load_golden(name)
    ef = 5 if "in.edge_features" in z.files else None
    m = _load(ResGatedGraphConv(8, 12, in_edge_features=ef), z, dev)
    gk = ("x", "edge_features") if ef else ("x",)
    I = _inputs(z, dev, gk)
    out = m(I["x"], I["edge_index"], I.get("edge_features"))
    _check(z, m, out, I, gk)


def test_hetero_conv_resgated():
    from analysisgnn_amd.core_layers import HeteroConv, ResGatedGraphConv
    dev = torch.device("cuda:0")
    z = load_golden("heteroconv_resgated")
    rels = [str(r) for r in z["meta.rels"]]
    m = _load(HeteroConv(8, 8, etypes={r: i for i, r in enumerate(rels)}, module=ResGatedGraphConv), z, dev)
    I = _inputs(z, dev)
    out = m(I["x"], I["edge_index"], I["edge_type"])
    _check(z, m, out, I)


def test_gat():
    from analysisgnn_amd.core_layers import GATConvLayer
    dev = torch.device("cuda:0")
    z = load_golden("gat")
    m = _load(GATConvLayer(8, 10, num_heads=3, dropout=0.0), z, dev)
    I = _inputs(z, dev)
    out = m(I["x"], I["edge_index"])
    assert_close(out, z["out"], TOL, "out")
    (out * torch.from_numpy(z["gout"]).to(dev)).sum().backward()
    assert_close(I["x"].grad, z["grad.x"], TOL, "grad.x")
    for n, p in m.named_parameters():          # attention-score gradients vanish analytically (softmax over heads)
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        assert_close(g, z[f"gw.{n}"], TOL, f"gw.{n}")


def test_jumping_knowledge():
    from analysisgnn_amd.core_layers import JumpingKnowledge
    dev = torch.device("cuda:0")
    z = load_golden("jk")
    m = _load(JumpingKnowledge(n_hidden=8, n_layers=3), z, dev).train()
    I = _inputs(z, dev, ("x0", "x1", "x2"))
    out = m([I["x0"], I["x1"], I["x2"]])
    _check(z, m, out, I, ("x0", "x1", "x2"))


@pytest.mark.parametrize("tag", ["eq", "ragged"])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_intree_metrical_gnn(tag, mode):
    """In-tree MetricalGNN(metrical=True): note->beat/measure scatter sums on the SpMM kernel, library GRU (hidden 8),
    BatchNorm.  Train-mode gradients use the 1e-4 budget the fp32 BatchNorm backward needs (see test_oracle_intree);
    eval mode checks the forward only (the library RNN refuses backward in eval mode, as cuDNN does)."""
    from analysisgnn_amd.core_layers import MetricalGNN
    dev = torch.device("cuda:0")
    z = load_golden(f"metrical_{tag}_{mode}")
    rels = [str(r) for r in z["meta.rels"]]
    m = _load(MetricalGNN(8, 8, 8, etypes={r: i for i, r in enumerate(rels)}, num_layers=3, dropout=0.0, metrical=True), z, dev)
    m.train(mode == "train")
    I = _inputs(z, dev)
    out = m(I["x"], I["edge_index"], I["edge_type"], I["beat_nodes"], I["measure_nodes"], I["beat_edges"],
            I["measure_edges"], beat_lengths=I.get("beat_lengths"), measure_lengths=I.get("measure_lengths"))
    assert_close(out, z["out"], TOL, "out")
    if mode == "train":
        (out * torch.from_numpy(z["gout"]).to(dev)).sum().backward()
        assert_close(I["x"].grad, z["grad.x"], 3e-4, "grad.x")
        for n, p in m.named_parameters():
            assert_close(p.grad, z[f"gw.{n}"], 3e-4, f"gw.{n}")


@pytest.mark.parametrize("reduce", ["sum", "mean"])
@pytest.mark.parametrize("with_out", [True, False])
def test_scatter_surface(reduce, with_out):
    """analysisgnn_amd.scatter (torch_scatter-compatible names) vs the restated semantics (oracle/scatter_ref.py)."""
    from analysisgnn_amd import scatter as S
    from oracle import scatter_ref as R
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    n, e, w = 37, 140, 10
    src = torch.randn(e, w, generator=g)
    idx = torch.randint(0, n, (e,), generator=g)
    out0 = torch.randn(n, w, generator=g) if with_out else None
    sc, oc = src.clone().requires_grad_(True), (out0.clone().requires_grad_(True) if with_out else None)
    ref = R.scatter(sc, idx, 0, out=(oc.clone() if with_out else None), dim_size=None if with_out else n, reduce=reduce)
    sg = src.to(dev).requires_grad_(True)
    og = out0.to(dev).requires_grad_(True) if with_out else None
    got = S.scatter(sg, idx.to(dev), 0, out=og, dim_size=None if with_out else n, reduce=reduce)
    assert_close(got, ref, 1e-5, "scatter")
    go = torch.randn(ref.shape, generator=g)
    (ref * go).sum().backward()
    (got * go.to(dev)).sum().backward()
    assert_close(sg.grad, sc.grad, 1e-5, "dsrc")
    if with_out:
        assert_close(og.grad, oc.grad, 1e-5, "dout")
    assert S.scatter_add is S.scatter_sum and callable(S.scatter_mean)


def test_scatter_is_a_registered_op_and_traces_without_a_graph_break():
    """the reference may wrap the model in torch.compile(dynamic=True) (train/train_analysisgnn.py:202-203) — custom
    ops must be registered or break the graph cleanly.  `analysisgnn_amd::scatter_reduce` carries a fake (meta) kernel and an
    autograd formula: torch.library.opcheck passes, and a SageConvScatter-shaped function (core/gnn.py:62-76) compiles with
    fullgraph=True (any graph break raises) and gives the eager values and gradients."""
    from analysisgnn_amd import scatter as S
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    n, e, w = 50, 180, 16
    x = torch.randn(n, w, generator=g).to(dev)
    wt = (torch.randn(w, w, generator=g) * 0.3).to(dev)
    idx = torch.randint(0, n, (2, e), generator=g).to(dev)
    for mean in (False, True):
        src = torch.randn(e, w, device=dev, requires_grad=True)
        out0 = torch.randn(n, w, device=dev, requires_grad=True)
        torch.library.opcheck(S._scatter_op, (src, idx[0], out0, n, mean))
        torch.library.opcheck(S._scatter_op, (src, idx[0], None, n, mean))

    def layer(x, wt, idx):
        h = x @ wt
        s = S.scatter(h[idx[1]], idx[0], 0, out=h.clone(), reduce="mean")
        return torch.cat([x, s], dim=-1).relu()

    xe = x.clone().requires_grad_(True)
    we = wt.clone().requires_grad_(True)
    ref = layer(xe, we, idx)
    ref.sum().backward()
    xc = x.clone().requires_grad_(True)
    wc = wt.clone().requires_grad_(True)
    compiled = torch.compile(layer, fullgraph=True, dynamic=True, backend="aot_eager")
    out = compiled(xc, wc, idx)
    out.sum().backward()
    assert_close(out, ref, 1e-6, "compiled forward")
    assert_close(xc.grad, xe.grad, 1e-5, "compiled dx")
    assert_close(wc.grad, we.grad, 1e-5, "compiled dw")
