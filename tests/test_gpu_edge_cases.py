"""Edge cases the boundary has to tolerate: relations with zero edges, node types absent from a batch,
isolated nodes, a single subgraph of one note, rows with more than 64 neighbours, ragged subgraph lengths.  Each case is
checked against the CPU oracle (1e-4 relative)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import assert_close  # noqa: E402

DEV = "cuda:0"


def _run_hybrid(cls_name, g, I, H=32, L=2, heads=4, **kw):
    from analysisgnn_amd.encoders import HybridGNN
    from analysisgnn_amd.hgt import HybridHGT
    from oracle import encoders_ref as E
    torch.manual_seed(0)
    if cls_name == "sage":
        m = HybridGNN(metadata=g.metadata(), input_channels=H, hidden_channels=H, num_layers=L, dropout=0.0).train()
    else:
        m = HybridHGT(metadata=g.metadata(), input_channels=H, hidden_channels=H, num_layers=L, heads=heads, dropout=0.0).train()
    P = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    m = m.to(DEV)
    xc = {k: v.clone().requires_grad_(True) for k, v in I["x_dict"].items()}
    if cls_name == "sage":
        ref = E.hybrid_gnn(P, "", g.metadata(), L, xc, I["edge_index_dict"], I["batch_dict"], I["batch_size"])
    else:
        ref = E.hybrid_hgt(P, "", g.metadata(), L, heads, xc, I["edge_index_dict"], I["batch_dict"], I["batch_size"])
    xg = {k: v.to(DEV).requires_grad_(True) for k, v in I["x_dict"].items()}
    out = m(x_dict=xg, edge_index_dict={k: v.to(DEV) for k, v in I["edge_index_dict"].items()},
            batch_dict={k: v.to(DEV) for k, v in I["batch_dict"].items()}, batch_size=I["batch_size"],
            neighbor_mask_node=None, neighbor_mask_edge=None)
    assert_close(out, ref, 1e-4, "out")
    ref.sum().backward()
    out.sum().backward()
    for k in xc:
        if xc[k].grad is not None:
            assert_close(xg[k].grad, xc[k].grad, 1e-4, f"dx[{k}]")


@pytest.mark.parametrize("enc", ["sage", "hgt"])
def test_relation_without_edges_and_isolated_nodes(enc):
    from analysisgnn_amd.synth import make_batch, torch_inputs
    g = make_batch(2, 40, first_seed=30)
    g.edge_index[("note", "rest", "note")] = np.zeros((2, 0), dtype=np.int64)           # a relation with zero edges
    e = g.edge_index[("note", "during", "note")]
    g.edge_index[("note", "during", "note")] = e[:, (e[1] % 7) != 0]                     # some destinations lose all during-edges
    I = torch_inputs(g, in_channels=32, seed=1)
    _run_hybrid(enc, g, I)


@pytest.mark.parametrize("enc", ["sage", "hgt"])
def test_node_type_absent_from_batch(enc):
    """metadata declares beat nodes / relations, this batch carries none of them."""
    from analysisgnn_amd.synth import make_batch, torch_inputs
    g_full = make_batch(1, 30, first_seed=3, add_beats=True)
    g = make_batch(1, 30, first_seed=3)
    I = torch_inputs(g, in_channels=32, seed=2)
    class Both:                                     # model built for the full metadata, data without beats
        def metadata(self):
            return g_full.metadata()
    g.metadata = Both().metadata
    _run_hybrid(enc, g, I)


@pytest.mark.parametrize("enc", ["sage", "hgt"])
def test_single_note_and_ragged_subgraphs(enc):
    from analysisgnn_amd.synth import collate, make_score_graph, torch_inputs
    g = collate([make_score_graph(seed=1, n_notes=1), make_score_graph(seed=2, n_notes=37), make_score_graph(seed=3, n_notes=5)])
    I = torch_inputs(g, in_channels=32, seed=3)
    _run_hybrid(enc, g, I)


def test_rows_with_more_than_64_neighbours_h256():
    """A 90-note chord: every note has 90 onset neighbours (several 64-id batches in the SpMM, long HGT segments)."""
    from analysisgnn_amd.encoders import HeteroConv
    from analysisgnn_amd.hgt import HGTConv
    from analysisgnn_amd.synth import ScoreGraph, torch_inputs
    from oracle import pyg_ref as G
    n = 150
    src, dst = np.nonzero(np.ones((90, 90), dtype=bool))
    ring = np.stack([np.arange(n), (np.arange(n) + 1) % n])
    g = ScoreGraph(num_nodes={"note": n}, edge_index={("note", "onset", "note"): np.stack([src, dst]).astype(np.int64),
                                                        ("note", "consecutive", "note"): ring.astype(np.int64)},
                   batch={"note": np.zeros(n, dtype=np.int64)}, onset_div=np.zeros(n, dtype=np.int64),
                   duration_div=np.ones(n, dtype=np.int64), batch_size=n)
    I = torch_inputs(g, in_channels=256, seed=4)
    torch.manual_seed(1)
    for make, ref_fn in ((lambda: HeteroConv(g.edge_types, 256, 256), None), (lambda: HGTConv(256, 256, g.metadata(), 4), "hgt")):
        m = make()
        P = {k: v.detach().clone() for k, v in m.state_dict().items()}
        m = m.to(DEV)
        with torch.no_grad():
            out = m({k: v.to(DEV) for k, v in I["x_dict"].items()}, {k: v.to(DEV) for k, v in I["edge_index_dict"].items()})
            if ref_fn is None:
                ref = G.hetero_conv_sage(P, "", g.edge_types, I["x_dict"], I["edge_index_dict"], "sum")
            else:
                ref = G.hgt_conv(P, "", ["note"], g.edge_types, 4, I["x_dict"], I["edge_index_dict"])
        assert_close(out["note"], ref["note"], 1e-4, type(m).__name__)


def test_empty_batch_kernels_do_not_launch():
    from analysisgnn_amd.graph import SegSpec, build_csr
    from analysisgnn_amd import ops
    e = torch.zeros(0, dtype=torch.int64, device=DEV)
    fwd, bwd = build_csr([SegSpec(e, e, 5), SegSpec(e, e, 5)])
    assert fwd.rowptr.tolist() == [0] * 6
    x = torch.randn(5, 8, device=DEV, requires_grad=True)
    out = ops.aggregate(ops.AggSpec(fwd=[fwd], bwd=[bwd], src_id=[0], n_rows=5, mean=True, shared_slot=True), [x], self_t=x)
    assert torch.equal(out, x)                                  # (x + 0) / max(0, 1)
    out.sum().backward()
    assert torch.equal(x.grad, torch.ones_like(x))
    out0 = ops.aggregate(ops.AggSpec(fwd=[fwd], bwd=[bwd], src_id=[0], n_rows=0, mean=True, shared_slot=True), [x])
    assert out0.shape == (0, 8)


@pytest.mark.parametrize("shared,mean,H", [(True, True, 128), (False, True, 64), (True, False, 256), (False, False, 32)])
def test_self_gradient_of_the_aggregation_matches_torch(shared, mean, H):
    """d/d self of ops.aggregate (agnn_spmm_self_grad_f32) against autograd on the dense formula, with `self` a separate
    matrix and with `self` being the source matrix itself (onset pooling: the two gradients are produced as one)."""
    from analysisgnn_amd import ops
    from analysisgnn_amd.graph import SegSpec, build_csr
    rng = np.random.default_rng(H)
    n, R = 97, 3
    es = [rng.integers(0, n, size=(2, e)) for e in (400, 0, 150)]
    specs = []
    for e in es:
        r, c = torch.from_numpy(e[0]).to(DEV), torch.from_numpy(e[1]).to(DEV)
        specs += [SegSpec(r, c, n_rows=n), SegSpec(c, r, n_rows=n)]
    csrs = build_csr(specs)
    spec = ops.AggSpec(fwd=csrs[0::2], bwd=csrs[1::2], src_id=[0] * R, n_rows=n, mean=mean, shared_slot=shared)
    x0 = torch.randn(n, H)
    g = torch.randn(n, H if shared else R * H)

    def dense(x, s):
        outs = []
        for e in es:
            a = torch.zeros(n, n)
            a.index_put_((torch.from_numpy(e[0]), torch.from_numpy(e[1])), torch.ones(e.shape[1]), accumulate=True)
            cnt = a.sum(1, keepdim=True).clamp(min=1)
            v = a @ x + s
            outs.append(v / cnt if mean else v)
        return sum(outs) if shared else torch.cat(outs, dim=1)

    xr, sr = x0.clone().requires_grad_(True), x0.clone().requires_grad_(True)
    dense(xr, sr).backward(g)
    # separate matrices
    xa, sa = x0.to(DEV).requires_grad_(True), x0.to(DEV).clone().requires_grad_(True)
    out = ops.aggregate(spec, [xa], self_t=sa)
    assert_close(out, dense(x0, x0), 1e-4, "forward")
    out.backward(g.to(DEV))
    assert_close(sa.grad, sr.grad, 1e-4, "d self")
    assert_close(xa.grad, xr.grad, 1e-4, "d src")
    # one matrix in both roles
    xb = x0.to(DEV).requires_grad_(True)
    ops.aggregate(spec, [xb], self_t=xb).backward(g.to(DEV))
    assert_close(xb.grad, xr.grad + sr.grad, 1e-4, "d (src + self)")
