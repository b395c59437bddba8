"""Round-2 additions against golden vectors produced by RUNNING the reference's own code (oracle/gen_golden_r2.py):
RelEdgeConv (core/gnn.py:79-106), HeteroRelEdgeConvLayer (core/hgnn.py:66-95), the `lstm` / `none` reductions with
HeteroAttention (core/hgnn.py:8-23), and onsetwise_logit_aggregation (models/analysis.py:44-101).  fp32; tolerance 1e-4
relative to max(1, |ref|max)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import assert_close, load_golden  # noqa: E402
from test_gpu_core_layers import _check, _inputs, _load  # noqa: E402

DEV = torch.device("cuda:0")
ETYPES = {"onset": 0, "consecutive": 1, "during": 2, "rest": 3}


@pytest.mark.parametrize("name", ["r2_reledge", "r2_reledge_edgefeat"])
def test_rel_edge_conv(name):
    from analysisgnn_amd.core_layers import RelEdgeConv
    z = load_golden(name)
    ef = 5 if "in.edge_features" in z.files else None
    m = _load(RelEdgeConv(8, 12, in_edge_features=ef), z, DEV)
    gk = ("x", "edge_features") if ef else ("x",)
    I = _inputs(z, DEV, gk)
    out = m(I["x"], I["edge_index"], I.get("edge_features"))
    _check(z, m, out, I, gk)


@pytest.mark.parametrize("name", ["r2_hreledge", "r2_hreledge_nodefeat", "r2_hreledge_edgefeat"])
def test_hetero_rel_edge_conv_layer(name):
    from analysisgnn_amd.core_layers import HeteroRelEdgeConvLayer
    z = load_golden(name)
    has_ef = "in.edge_features" in z.files
    m = _load(HeteroRelEdgeConvLayer(8, 8, etypes=ETYPES, in_edge_features=3 if has_ef else None), z, DEV)
    gk = ("x", "edge_features") if has_ef else ("x",)
    I = _inputs(z, DEV, gk)
    out = m(I["x"], I["edge_index"], I["edge_type"], I.get("edge_features"))
    _check(z, m, out, I, gk)


@pytest.mark.parametrize("name,kind,red", [("r2_hsage_lstm", "sage", "lstm"), ("r2_hresgated_lstm", "gated", "lstm"),
                                           ("r2_hresgated_none", "gated", "none")])
def test_lstm_and_none_reductions(name, kind, red):
    from analysisgnn_amd.core_layers import HeteroResGatedGraphConvLayer, HeteroSageConvLayer
    z = load_golden(name)
    cls = HeteroSageConvLayer if kind == "sage" else HeteroResGatedGraphConvLayer
    m = _load(cls(8, 8, etypes=ETYPES, reduction=red), z, DEV)
    I = _inputs(z, DEV)
    out = m(I["x"], I["edge_index"], I["edge_type"])
    _check(z, m, out, I)


def test_unrunnable_reductions_raise_as_in_the_reference():
    from analysisgnn_amd.core_layers import HeteroSageConvLayer
    for red in ("max", "min", "concat"):
        with pytest.raises(NotImplementedError):
            HeteroSageConvLayer(8, 8, etypes=ETYPES, reduction=red)


@pytest.mark.parametrize("tag", ["plain", "tpc", "halo"])
def test_onsetwise_logit_aggregation_matches_reference_output(tag):
    from analysisgnn_amd.postprocess import onsetwise_logit_aggregation
    z = load_golden("r2_onsetwise_agg")
    probs = {k[len(tag) + 4:]: torch.from_numpy(z[k]).to(DEV) for k in z.files if k.startswith(f"{tag}.in.")}
    bs = int(z[f"{tag}.batch_size"])
    eid = {("note", "onset", "note"): torch.from_numpy(z["onset_edges"]).to(DEV)}
    out = onsetwise_logit_aggregation(probs, edge_index_dict=eid, batch_size=bs, batch=torch.zeros(120, dtype=torch.long, device=DEV),
                                      onset_div=torch.from_numpy(z["onset_div"]).to(DEV))
    keys = [k[len(tag) + 5:] for k in z.files if k.startswith(f"{tag}.out.")]
    assert sorted(keys) == sorted(out)
    for k in keys:
        assert_close(out[k], z[f"{tag}.out.{k}"], 1e-5, k)


def test_onsetwise_logit_aggregation_at_score_size_against_oracle():
    """A whole score (2 000 notes) against the CPU restatement (itself pinned by the reference-generated fixture)."""
    from analysisgnn_amd.postprocess import onsetwise_logit_aggregation
    from analysisgnn_amd.synth import make_score_graph
    from oracle import intree_ref as R
    g = make_score_graph(seed=3, n_notes=2000)
    gen = torch.Generator().manual_seed(0)
    classes = {"quality": 15, "inversion": 4, "degree1": 22, "degree2": 22, "localkey": 50, "tpc_in_label": 2}
    probs = {k: torch.softmax(6.0 * torch.randn(40, c, generator=gen).repeat_interleave(50, dim=0) + torch.randn(2000, c, generator=gen), -1)
             for k, c in classes.items()}
    e = torch.from_numpy(g.edge_index[("note", "onset", "note")])
    on = torch.from_numpy(g.onset_div)
    ref = R.onsetwise_logit_aggregation({k: v.clone() for k, v in probs.items()}, e, torch.zeros(2000, dtype=torch.long), on)
    out = onsetwise_logit_aggregation({k: v.to(DEV) for k, v in probs.items()}, edge_index_dict={("note", "onset", "note"): e.to(DEV)},
                                      batch=torch.zeros(2000, dtype=torch.long, device=DEV), onset_div=on.to(DEV))
    for k in ref:
        assert_close(out[k], ref[k], 1e-5, k)


def test_predict_runs_whole_score_and_returns_distributions():
    from analysisgnn_amd.models import TorchAnalysisGNN
    from analysisgnn_amd.postprocess import predict
    from analysisgnn_amd.synth import make_score_graph, torch_inputs
    tasks = {"quality": 15, "inversion": 4, "degree1": 22, "degree2": 22, "localkey": 50, "cadence": 4}
    g = make_score_graph(seed=1, n_notes=300)
    torch.manual_seed(0)
    m = TorchAnalysisGNN(g.metadata(), 25, 256, 128, tasks, 3, dropout=0.3, use_jk=False, logit_fusion=True).to(DEV).train()
    I = torch_inputs(g, 25, DEV, 0)
    out = predict(m, I["pitch_spelling"], I["key_signature"], I["x_dict"], I["edge_index_dict"], I["batch_dict"],
                  onset_div=torch.from_numpy(g.onset_div).to(DEV))
    assert m.training                                       # mode restored
    assert sorted(out) == sorted(tasks)
    for k, v in out.items():
        assert v.shape == (300, tasks[k])
        assert torch.allclose(v.sum(-1), torch.ones(300, device=DEV), atol=1e-4), k
