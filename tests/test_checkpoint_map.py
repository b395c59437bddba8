"""Lightning `.ckpt` of the reference -> this build's state_dict (analysisgnn_amd/postprocess.py; reference
inference/predict_analysis.py:150-159).  A synthetic checkpoint in the Lightning layout (written by this test, so only
tensors and plain containers) goes through `torch.save` / `torch.load(weights_only=True)` and must restore every
in-tree parameter by name.  CPU only: no kernel runs."""
import torch

from analysisgnn_amd.heads import MultiTaskLoss
from analysisgnn_amd.models import TorchAnalysisGNN
from analysisgnn_amd.postprocess import checkpoint_state_dict, load_reference_checkpoint
from analysisgnn_amd.synth import make_score_graph

TASKS = {"cadence": 4, "localkey": 50, "romanNumeral": 185}


def _model(seed):
    g = make_score_graph(seed=0, n_notes=30)
    torch.manual_seed(seed)
    return TorchAnalysisGNN(g.metadata(), 25, 32, 128, TASKS, 2, dropout=0.0, use_jk=False, logit_fusion=True)


def test_lightning_style_checkpoint_round_trip(tmp_path):
    src, dst = _model(1), _model(2)
    loss = MultiTaskLoss(list(TASKS))
    ckpt = {"epoch": 3, "global_step": 120, "pytorch-lightning_version": "2.2.0",
            "state_dict": {**{f"model.{k}": v.clone() for k, v in src.state_dict().items()},
                           "clf_loss.params": torch.tensor([0.5, 1.5, 2.0]),
                           **{f"memory_model.{k}": torch.zeros_like(v) for k, v in list(src.state_dict().items())[:3]}},
            "hyper_parameters": {"num_layers": 2, "hidden_channels": 32, "task_dict": dict(TASKS), "mt_strategy": "wloss"}}
    path = tmp_path / "model.ckpt"
    torch.save(ckpt, path)
    missing, unexpected, hp = load_reference_checkpoint(dst, str(path), strict=True, clf_loss=loss)
    assert missing == [] and unexpected == []
    assert hp["task_dict"] == TASKS
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k
    assert torch.equal(loss.params.detach(), torch.tensor([0.5, 1.5, 2.0]))
    model_sd, rest = checkpoint_state_dict(ckpt)
    assert set(model_sd) == set(src.state_dict()) and "clf_loss.params" in rest and any(k.startswith("memory_model.") for k in rest)


def test_in_tree_parameter_names_are_the_reference_names():
    """The names the reference's constructor creates in-tree (models/analysis.py:423-511) — checked literally, so that a
    reference checkpoint maps by name for everything outside the third-party encoder."""
    names = set(_model(0).state_dict())
    for n in ["pitch_embedding.weight", "key_embedding.weight", "project_dict.note.0.weight", "project_dict.note.2.bias",
              "project_dict.note.4.weight", "project_enc.0.weight", "project_enc.1.weight", "project_enc.3.weight", "project_enc.5.bias",
              "project_enc.7.weight", "project_enc.9.weight", "clf_dict.cadence.0.weight", "clf_dict.cadence.2.weight",
              "clf_dict.romanNumeral.3.bias", "clf_proj_layers.localkey.0.weight", "clf_proj_layers.localkey.2.bias",
              "cross_task_transformer.multihead_attn.in_proj_weight", "cross_task_transformer.multihead_attn.in_proj_bias",
              "cross_task_transformer.multihead_attn.out_proj.weight", "cross_task_transformer.norm.weight",
              "fusion_layers.cadence.weight", "fusion_layers.romanNumeral.bias"]:
        assert n in names, n
    assert all(k.split(".")[0] in ("pitch_embedding", "key_embedding", "project_dict", "encoder", "project_enc", "clf_dict",
                                   "clf_proj_layers", "cross_task_transformer", "fusion_layers") for k in names)


def test_partial_checkpoint_reports_missing_and_unexpected():
    src, dst = _model(1), _model(2)
    sd = {f"model.{k}": v for k, v in src.state_dict().items() if not k.startswith("encoder.")}
    sd["model.encoder.some_graphmuse_name.weight"] = torch.zeros(3)
    import pytest
    from analysisgnn_amd._lib import AgnnError
    with pytest.raises(AgnnError, match="encoder"):             # a half-initialised encoder is never silent
        load_reference_checkpoint(_model(2), {"state_dict": sd})
    missing, unexpected, _ = load_reference_checkpoint(dst, {"state_dict": sd}, allow_encoder_mismatch=True)
    assert unexpected == ["encoder.some_graphmuse_name.weight"]
    assert missing and all(k.startswith("encoder.") for k in missing)
    assert torch.equal(dst.state_dict()["project_enc.9.weight"], src.state_dict()["project_enc.9.weight"])
