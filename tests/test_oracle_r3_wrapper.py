"""oracle/encoders_ref.py's restatement of the WRAPPER (TorchAnalysisGNN.encode + the 21-head block, models/analysis.py:546-591)
against fixtures produced by running the reference's own class source (oracle/gen_golden_r3.py).  The encoder inside the
fixtures IS oracle/encoders_ref.py (graphmuse exists nowhere offline), so what this pins is everything around it: embedding
cat, input MLPs, onset pool, project_enc, task heads.  Float64, 1e-9."""
import numpy as np
import pytest
import torch

from helpers import R3_CASES, assert_close_rel, r3_case


@pytest.mark.parametrize("name", [n for n in R3_CASES if "h256" not in n])
def test_oracle_wrapper_matches_reference_run(name):
    from oracle import encoders_ref as E
    z, cfg, g, I, labels = r3_case(name)
    P = {k[2:]: torch.from_numpy(np.asarray(z[k])).double() for k in z.files if k.startswith("w.") and not k.startswith("w.clf_loss")}
    x = E.analysis_encode(P, cfg["enc"], g.metadata(), cfg["L"], I["pitch_spelling"], I["key_signature"],
                          {k: v.double() for k, v in I["x_dict"].items()}, I["edge_index_dict"], I["batch_dict"], I["batch_size"],
                          I["neighbor_mask_node"], I["neighbor_mask_edge"], use_jk=cfg["use_jk"])
    assert_close_rel(x, z["x"], 1e-9, "encode")
    if not cfg["fusion"]:
        logits = E.analysis_logits(P, x, list(cfg["tasks"]))
        for t in cfg["tasks"]:
            assert_close_rel(logits[t], z[f"logits.{t}"], 1e-9, f"logits[{t}]")


def test_fixture_objective_is_the_reference_composition():
    """loss.total of a fixture recomputed from its own logits with plain torch: sum_t (0.5 / p_t^2 CE_t + log(1 + p_t^2)) / T
    + 0.1 * mean(x^2)  (models/chord.py:39-49, models/analysis.py:1034-1036, :984, :1072) — guards the generator itself."""
    z, cfg, g, I, labels = r3_case("r3_wrapper_hybrid_plain_sampled")
    p = torch.from_numpy(z["w.clf_loss.params"]).double()
    tot = 0.0
    for i, t in enumerate(cfg["tasks"]):
        ce = torch.nn.functional.cross_entropy(torch.from_numpy(z[f"logits.{t}"]), labels[i], ignore_index=-1, label_smoothing=0.1)
        assert abs(float(ce) - float(z["loss.per_task"][i])) < 1e-12
        tot = tot + 0.5 / p[i] ** 2 * ce + torch.log(1 + p[i] ** 2)
    tot = tot / len(cfg["tasks"]) + 0.1 * torch.from_numpy(z["x"]).pow(2).mean()
    assert abs(float(tot) - float(z["loss.total"])) < 1e-7      # the reference keeps `params` in fp32: 0.5 / p^2 is an fp32 value
