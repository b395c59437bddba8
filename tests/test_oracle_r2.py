"""Pin the round-2 CPU restatements (oracle/intree_ref.py: RelEdgeConv, HeteroRelEdgeConvLayer, HeteroAttention /
the `lstm` and `none` reductions, onsetwise_logit_aggregation) against fixtures produced by running the reference's own
code (oracle/gen_golden_r2.py).  fp32, tolerance 1e-5 relative to max(1, max|ref|)."""
import numpy as np
import pytest
import torch

from oracle import intree_ref as R
from helpers import assert_close, inputs_from_npz, load_golden, params_from_npz
from test_oracle_intree import _check

RELS = ["onset", "consecutive", "during", "rest"]


@pytest.mark.parametrize("name", ["r2_reledge", "r2_reledge_edgefeat"])
def test_rel_edge_conv(name):
    z = load_golden(name)
    gk = ["x"] + (["edge_features"] if "in.edge_features" in z.files else [])
    P, I = params_from_npz(z), inputs_from_npz(z, gk)
    out = R.rel_edge_conv(P, "", I["x"], I["edge_index"], I.get("edge_features"))
    _check(z, out, P, I, gk)


@pytest.mark.parametrize("name", ["r2_hreledge", "r2_hreledge_nodefeat", "r2_hreledge_edgefeat"])
def test_hetero_rel_edge_layer(name):
    z = load_golden(name)
    gk = ["x"] + (["edge_features"] if "in.edge_features" in z.files else [])
    P, I = params_from_npz(z), inputs_from_npz(z, gk)
    out = R.hetero_rel_edge_layer(P, "", RELS, I["x"], I["edge_index"], I["edge_type"], I.get("edge_features"))
    _check(z, out, P, I, gk)


@pytest.mark.parametrize("name,conv,red", [("r2_hsage_lstm", "sage", "lstm"), ("r2_hresgated_lstm", "gated", "lstm"),
                                           ("r2_hresgated_none", "gated", "none")])
def test_hetero_layer_lstm_and_none_reductions(name, conv, red):
    z = load_golden(name)
    P, I = params_from_npz(z), inputs_from_npz(z, ["x"])
    fn = R.sage_conv_scatter if conv == "sage" else R.res_gated_conv
    out = R.hetero_layer_reduce(P, "", RELS, fn, I["x"], I["edge_index"], I["edge_type"], red)
    _check(z, out, P, I, ["x"], gtol=2e-5)


@pytest.mark.parametrize("tag", ["plain", "tpc", "halo"])
def test_onsetwise_logit_aggregation(tag):
    z = load_golden("r2_onsetwise_agg")
    probs = {k[len(tag) + 4:]: torch.from_numpy(z[k]).clone() for k in z.files if k.startswith(f"{tag}.in.")}
    bs = int(z[f"{tag}.batch_size"])
    out = R.onsetwise_logit_aggregation(probs, torch.from_numpy(z["onset_edges"]), torch.zeros(120, dtype=torch.long),
                                        torch.from_numpy(z["onset_div"]), batch_size=bs)
    keys = [k[len(tag) + 5:] for k in z.files if k.startswith(f"{tag}.out.")]
    assert sorted(keys) == sorted(out)
    changed = 0
    for k in keys:
        assert_close(out[k], z[f"{tag}.out.{k}"], 1e-6, k)
        changed += int(not np.allclose(z[f"{tag}.out.{k}"], z[f"{tag}.in.{k}"]))
    assert changed >= 4                                   # the four RNA keys are aggregated, the others pass through
