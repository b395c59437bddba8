"""End-to-end parity of the HIP-backed encoders (analysisgnn_amd/encoders.py, models.py) against
the CPU restatement (oracle/encoders_ref.py) on identical weights and inputs: outputs, input
gradients and every weight gradient within 1e-4 (fp32, relative to max(1,|ref|max)) — the
tolerance BASELINE.json's north_star states.  eval mode (dropout = identity)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import assert_close  # noqa: E402

TOL = 1e-4
DEV = "cuda:0"


def _cpu_params(module):
    return {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in module.state_dict().items()}


def _cmp_grads(module, P, tol=TOL):
    n = 0
    for name, p in module.named_parameters():
        if P[name].grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None, f"{name}: no gradient on the HIP path"
        assert_close(p.grad, P[name].grad, tol, f"grad {name}")
        n += 1
    assert n > 0


def _graph(kind):
    from analysisgnn_amd.synth import make_batch, make_score_graph, sample_hops
    if kind == "notes":
        return make_batch(3, 60, first_seed=3)
    if kind == "metrical":
        return make_batch(2, 70, first_seed=5, add_beats=True, add_measures=True, reverse_metrical_edges=True)
    if kind == "sampled":
        return sample_hops(make_score_graph(seed=2, n_notes=300), n_targets=64, num_neighbors=[4, 4], seed=1, random_targets=True)
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["notes", "metrical", "sampled"])
@pytest.mark.parametrize("aggr", ["sum", "mean"])
def test_hybrid_gnn(kind, aggr):
    from analysisgnn_amd.encoders import HybridGNN
    from analysisgnn_amd.synth import torch_inputs
    from oracle import encoders_ref as E
    g = _graph(kind)
    H, L = 32, 3
    torch.manual_seed(0)
    m = HybridGNN(metadata=g.metadata(), input_channels=H, hidden_channels=H, num_layers=L, dropout=0.0,
                  use_jk=(kind == "notes"), aggr=aggr).train()
    P = _cpu_params(m)
    m = m.to(DEV)
    I = torch_inputs(g, in_channels=H, seed=1)
    xc = {k: v.clone().requires_grad_(True) for k, v in I["x_dict"].items()}
    ref = E.hybrid_gnn(P, "", g.metadata(), L, xc, I["edge_index_dict"], I["batch_dict"], I["batch_size"],
                       I["neighbor_mask_node"], I["neighbor_mask_edge"], use_jk=(kind == "notes"), aggr=aggr)
    xg = {k: v.to(DEV).requires_grad_(True) for k, v in I["x_dict"].items()}
    out = m(x_dict=xg, edge_index_dict={k: v.to(DEV) for k, v in I["edge_index_dict"].items()},
            batch_dict={k: v.to(DEV) for k, v in I["batch_dict"].items()}, batch_size=I["batch_size"],
            neighbor_mask_node=I["neighbor_mask_node"], neighbor_mask_edge=I["neighbor_mask_edge"],
            return_edge_index=False, edge_attr_dict=None)
    assert out.shape == (I["batch_size"], H)
    assert_close(out, ref, TOL, "out")
    gout = torch.randn(ref.shape, generator=torch.Generator().manual_seed(9))
    (ref * gout).sum().backward()
    (out * gout.to(DEV)).sum().backward()
    for k in xc:
        if xc[k].grad is not None:
            assert_close(xg[k].grad, xc[k].grad, TOL, f"grad x[{k}]")
    _cmp_grads(m, P)


@pytest.mark.parametrize("kind", ["metrical", "sampled"])
@pytest.mark.parametrize("aggr", ["sum", "mean"])
def test_hybrid_gnn_width_256_one_gemm_layers(kind, aggr):
    """H = 256 is where a SAGE layer runs as ONE GEMM over [mean_1 .. mean_R | x_dst] (agnn_spmm_root_f32, encoders.HeteroConv):
    outputs, input gradients and every parameter gradient against the CPU restatement, on a trimmed sampled batch (the
    root's gradient folds into the transposed launch) and on a heterogeneous batch (destination types whose root operand
    is not among the sources)."""
    from analysisgnn_amd import ops
    from analysisgnn_amd.encoders import HybridGNN
    from analysisgnn_amd.synth import torch_inputs
    from oracle import encoders_ref as E
    g = _graph(kind)
    H, L = 256, 3
    torch.manual_seed(1)
    m = HybridGNN(metadata=g.metadata(), input_channels=H, hidden_channels=H, num_layers=L, dropout=0.0, use_jk=False, aggr=aggr).train()
    P = _cpu_params(m)
    m = m.to(DEV)
    I = torch_inputs(g, in_channels=H, seed=2)
    xc = {k: v.clone().requires_grad_(True) for k, v in I["x_dict"].items()}
    ref = E.hybrid_gnn(P, "", g.metadata(), L, xc, I["edge_index_dict"], I["batch_dict"], I["batch_size"],
                       I["neighbor_mask_node"], I["neighbor_mask_edge"], use_jk=False, aggr=aggr)
    xg = {k: v.to(DEV).requires_grad_(True) for k, v in I["x_dict"].items()}
    ops.SPMM_TRACE = []
    try:
        out = m(x_dict=xg, edge_index_dict={k: v.to(DEV) for k, v in I["edge_index_dict"].items()},
                batch_dict={k: v.to(DEV) for k, v in I["batch_dict"].items()}, batch_size=I["batch_size"],
                neighbor_mask_node=I["neighbor_mask_node"], neighbor_mask_edge=I["neighbor_mask_edge"])
        gout = torch.randn(ref.shape, generator=torch.Generator().manual_seed(9))
        (out * gout.to(DEV)).sum().backward()
        launches = [t[0] for t in ops.SPMM_TRACE]
    finally:
        ops.SPMM_TRACE = None
    assert "fwd" in launches and "bwd" in launches
    assert_close(out, ref, TOL, "out")
    (ref * gout).sum().backward()
    for k in xc:
        if xc[k].grad is not None:
            assert_close(xg[k].grad, xc[k].grad, TOL, f"grad x[{k}]")
    _cmp_grads(m, P)


def test_hop_index_tensor_masks_equal_count_lists():
    """graphmuse-style per-element hop-index tensors (pitch_spelling.py:388-391) give the same result as
    PyG per-hop count lists (datamodules/analysis.py:277)."""
    from analysisgnn_amd.encoders import MetricalGNN
    from analysisgnn_amd.synth import torch_inputs
    g = _graph("sampled")
    H = 16
    torch.manual_seed(1)
    m = MetricalGNN(H, H, 8, 3, g.metadata(), dropout=0.0).eval().to(DEV)
    I = torch_inputs(g, in_channels=H, seed=2, device=DEV)
    a = m(I["x_dict"], I["edge_index_dict"], I["neighbor_mask_node"], I["neighbor_mask_edge"])
    hop_n = {k: torch.repeat_interleave(torch.arange(len(v)), torch.tensor(v)).to(DEV) for k, v in I["neighbor_mask_node"].items()}
    hop_e = {k: torch.repeat_interleave(torch.arange(len(v)), torch.tensor(v)).to(DEV) for k, v in I["neighbor_mask_edge"].items()}
    b = m(I["x_dict"], I["edge_index_dict"], hop_n, hop_e)
    assert torch.equal(a, b)
    c = m(I["x_dict"], I["edge_index_dict"])                      # no trimming: a different (valid) result
    assert c.shape[0] == g.num_nodes["note"]


@pytest.mark.parametrize("enc", ["hybridgnn", "metricalgnn"])
def test_analysis_model_logits(enc):
    """TorchAnalysisGNN mirror (analysis.py:421-591): task logits within 1e-4 of the CPU path."""
    from analysisgnn_amd.models import TorchAnalysisGNN
    from analysisgnn_amd.synth import make_batch, torch_inputs
    from oracle import encoders_ref as E
    g = make_batch(2, 80, first_seed=11, add_beats=True, add_measures=True, reverse_metrical_edges=True)
    tasks = {"cadence": 4, "localkey": 50, "romanNumeral": 185}
    H, L = 32, 3
    torch.manual_seed(2)
    m = TorchAnalysisGNN(g.metadata(), in_channels=25, hidden_channels=H, out_channels=16, task_dict=tasks,
                         num_layers=L, dropout=0.0, use_jk=False, logit_fusion=False, encoder_type=enc).train()
    P = _cpu_params(m)
    m = m.to(DEV)
    I = torch_inputs(g, in_channels=25, seed=3)
    x = E.analysis_encode(P, enc, g.metadata(), L, I["pitch_spelling"], I["key_signature"], I["x_dict"],
                          I["edge_index_dict"], I["batch_dict"], I["batch_size"])
    ref = E.analysis_logits(P, x, list(tasks))
    J = {k: ({kk: vv.to(DEV) for kk, vv in v.items()} if isinstance(v, dict) and v and isinstance(next(iter(v.values())), torch.Tensor)
             else (v.to(DEV) if isinstance(v, torch.Tensor) else v)) for k, v in I.items()}
    out = m(J["pitch_spelling"], J["key_signature"], J["x_dict"], J["edge_index_dict"], J["batch_dict"],
            J["batch_size"], None, None)
    loss_r = sum((v ** 2).mean() for v in ref.values())
    loss_g = sum((v ** 2).mean() for v in out.values())
    for t in tasks:
        assert_close(out[t], ref[t], TOL, f"logits[{t}]")
    loss_r.backward()
    loss_g.backward()
    _cmp_grads(m, P)


def test_c1_shape_forward_h256():
    """BASELINE config C1: one ~500-note score graph, HybridGNN L=3 H=256, forward, no sampling."""
    from analysisgnn_amd.encoders import HybridGNN
    from analysisgnn_amd.synth import make_score_graph, torch_inputs
    from oracle import encoders_ref as E
    g = make_score_graph(seed=0, n_notes=500)
    torch.manual_seed(0)
    m = HybridGNN(metadata=g.metadata(), input_channels=256, hidden_channels=256, num_layers=3, dropout=0.3).eval()
    P = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(DEV)
    I = torch_inputs(g, in_channels=256, seed=0)
    with torch.no_grad():
        ref = E.hybrid_gnn(P, "", g.metadata(), 3, I["x_dict"], I["edge_index_dict"], I["batch_dict"], 500)
        out = m(x_dict={k: v.to(DEV) for k, v in I["x_dict"].items()},
                edge_index_dict={k: v.to(DEV) for k, v in I["edge_index_dict"].items()},
                batch_dict={k: v.to(DEV) for k, v in I["batch_dict"].items()}, batch_size=500,
                neighbor_mask_node=None, neighbor_mask_edge=None)
    assert_close(out, ref, TOL, "C1 out")


def test_module_obligations():
    """state_dict round trip, deepcopy, freezing, no_grad (SURVEY.md §8b behavioural obligations)."""
    import copy
    from analysisgnn_amd.encoders import HybridGNN
    from analysisgnn_amd.synth import make_batch, torch_inputs
    g = make_batch(2, 40)
    m = HybridGNN(metadata=g.metadata(), input_channels=16, hidden_channels=16, num_layers=2, dropout=0.5).to(DEV)
    m2 = copy.deepcopy(m)
    m2.load_state_dict(m.state_dict())
    I = torch_inputs(g, in_channels=16, device=DEV)
    kw = dict(x_dict=I["x_dict"], edge_index_dict=I["edge_index_dict"], batch_dict=I["batch_dict"],
              batch_size=I["batch_size"], neighbor_mask_node=None, neighbor_mask_edge=None)
    m.eval(); m2.eval()
    with torch.no_grad():
        assert torch.equal(m(**kw), m2(**kw))
    m.train()
    a, b = m(**kw), m(**kw)
    assert not torch.equal(a, b)                      # dropout active in train mode
    m.requires_grad_(False)
    assert not m(**kw).requires_grad
    with pytest.raises(Exception):
        m.cpu()(**{**kw, "x_dict": {k: v.cpu() for k, v in I["x_dict"].items()},
                   "edge_index_dict": {k: v.cpu() for k, v in I["edge_index_dict"].items()},
                   "batch_dict": {k: v.cpu() for k, v in I["batch_dict"].items()}})   # no CPU fallback


def test_torch_compile_wrapping_runs_the_same_path():
    """`torch.compile(model, dynamic=True)` as the reference's --compile flag does (train/train_analysisgnn.py:202-203): the
    model's entry points are excluded from tracing, so the compiled module runs the HIP path unchanged — same logits,
    gradients flow."""
    from analysisgnn_amd.models import TorchAnalysisGNN
    from analysisgnn_amd.synth import make_batch, torch_inputs
    g = make_batch(2, 60)
    tasks = {"cadence": 4, "localkey": 50}
    torch.manual_seed(0)
    m = TorchAnalysisGNN(g.metadata(), 25, 32, 128, tasks, 2, dropout=0.0, use_jk=False, logit_fusion=True).to(DEV).train()
    I = torch_inputs(g, 25, DEV, 0)
    args = (I["pitch_spelling"], I["key_signature"], I["x_dict"], I["edge_index_dict"], I["batch_dict"], I["batch_size"], None, None)
    ref = m(*args)
    cm = torch.compile(m, dynamic=True)
    out = cm(*args)
    for t in tasks:
        assert torch.equal(out[t], ref[t])
    sum(v.pow(2).mean() for v in out.values()).backward()
    assert all(p.grad is not None for n, p in m.named_parameters() if n.startswith("clf_dict"))
