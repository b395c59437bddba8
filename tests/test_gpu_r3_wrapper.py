"""The HIP path's wrapper — TorchAnalysisGNN.encode / forward_clf / the training objective — against fixtures made by running
the REFERENCE'S OWN class source in float64 (oracle/gen_golden_r3.py: models/analysis.py:408-602 `CrossTaskTransformer`,
`TorchAnalysisGNN`; models/chord.py:16-49 `MultiTaskLoss`; the encoder slot holds the CPU oracle, graphmuse being absent).
Logits (fusion on and off), encoder output, loss, input gradient and every weight gradient within 1e-4 of the tensor's own
largest magnitude (north star: "task logits within 1e-4 of reference")."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import R3_CASES, assert_close_rel, r3_case  # noqa: E402

DEV = torch.device("cuda:0")
TOL = 1e-4


def _to_dev(I):
    out = {}
    for k, v in I.items():
        if isinstance(v, dict) and v and isinstance(next(iter(v.values())), torch.Tensor):
            out[k] = {kk: vv.to(DEV) for kk, vv in v.items()}
        elif isinstance(v, torch.Tensor):
            out[k] = v.to(DEV)
        else:
            out[k] = v
    lens = getattr(I["batch_dict"]["note"], "agnn_target_lengths", None)
    if lens is not None:
        out["batch_dict"]["note"].agnn_target_lengths = lens
    return out


def _build(z, cfg, g):
    from analysisgnn_amd.heads import MultiTaskLoss
    from analysisgnn_amd.models import TorchAnalysisGNN
    from oracle.testing import seeded_fill_
    m = TorchAnalysisGNN(g.metadata(), cfg["in_ch"], cfg["H"], cfg["OUT"], cfg["tasks"], cfg["L"], dropout=0.0, use_jk=cfg["use_jk"],
                         logit_fusion=cfg["fusion"], encoder_type=cfg["enc"])
    if cfg["big"]:
        seeded_fill_(m, cfg["seed"], norm_offset=1.0)                # the generator's values, re-drawn from the seed
    else:
        sd = {k[2:]: torch.from_numpy(np.asarray(z[k])) for k in z.files if k.startswith("w.") and not k.startswith("w.clf_loss")}
        missing, unexpected = m.load_state_dict(sd, strict=True)     # the reference's names ARE the build's names
        assert not missing and not unexpected
    clf = MultiTaskLoss(list(cfg["tasks"]), requires_grad=cfg["wloss"])
    if cfg["wloss"]:
        with torch.no_grad():
            clf.params.copy_(torch.from_numpy(z["w.clf_loss.params"]))
    return m.to(DEV).train(), clf.to(DEV)


@pytest.mark.parametrize("fused_objective", [False, True])
@pytest.mark.parametrize("name", R3_CASES)
def test_wrapper_matches_reference_run(name, fused_objective):
    """`fused_objective=False`: the reference's own call sequence on the product modules (encode, forward_clf -> dict,
    clf_loss(dict, dict), / T, + lambda * feature loss).  True: what bench.py times (forward_clf_fused + heads.training_loss:
    CE, weighting, 1/T and the feature term in agnn_train_loss_f32)."""
    from analysisgnn_amd.heads import training_loss
    z, cfg, g, I, labels = r3_case(name)
    m, clf = _build(z, cfg, g)
    J = _to_dev(I)
    J["x_dict"]["note"].requires_grad_(True)
    tasks = list(cfg["tasks"])
    T = len(tasks)
    lab = labels.to(DEV)
    x = m.encode(J["pitch_spelling"], J["key_signature"], J["x_dict"], J["edge_index_dict"], J["batch_dict"], J["batch_size"],
                 J["neighbor_mask_node"], J["neighbor_mask_edge"])
    if fused_objective:
        cat, offs, _ = m.forward_clf_fused(x)
        logits = {t: cat[:, offs[i]:offs[i + 1]] for i, t in enumerate(tasks)}
        total, per = training_loss(cat, offs, lab, x, 0.1, 0.1, -1, task_params=clf.weights())
        per_task = [float(v) for v in per]
    else:
        logits = m.forward_clf(x)
        loss_dict = clf(logits, {t: lab[i] for i, t in enumerate(tasks)})
        total = loss_dict.pop("total") / T + 0.1 * x.pow(2).mean()
        per_task = [float(loss_dict[t]) for t in tasks]
    total.backward()
    torch.cuda.synchronize()
    if cfg["big"]:
        assert_close_rel(x[:8], z["x.head"], TOL, "encode (first rows)")
        for t in tasks:
            assert_close_rel(logits[t][:8], z[f"logits.{t}.head"], TOL, f"logits[{t}] (first rows)")
            s = float(logits[t].double().sum()), float(logits[t].double().abs().sum())
            assert abs(s[1] - z[f"logits.{t}.sum"][1]) <= TOL * z[f"logits.{t}.sum"][1], f"logits[{t}] checksum"
    else:
        assert_close_rel(x, z["x"], TOL, "encode")
        for t in tasks:
            assert_close_rel(logits[t], z[f"logits.{t}"], TOL, f"logits[{t}]")
    assert abs(float(total) - float(z["loss.total"])) <= TOL * abs(float(z["loss.total"])), (float(total), float(z["loss.total"]))
    np.testing.assert_allclose(per_task, z["loss.per_task"], rtol=TOL)
    gx = J["x_dict"]["note"].grad
    if cfg["big"]:
        assert_close_rel(gx[:8], z["grad.x_note.head"], TOL, "d loss / d x_note (first rows)", floor=1e-9)
    else:
        assert_close_rel(gx, z["grad.x_note"], TOL, "d loss / d x_note", floor=1e-9)
    n = 0
    for k, p in m.named_parameters():
        key = f"gw.{k}.head" if cfg["big"] else f"gw.{k}"
        if key not in z.files:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None, f"{k}: no gradient on the HIP path"
        if cfg["big"]:
            # the first 32 elements against the tensor's scale, taken from the checksum (sum |g| / numel = mean magnitude)
            ref_head = torch.from_numpy(z[key]).double()
            scale = max(float(ref_head.abs().max()), float(z[f"gw.{k}.sum"][1]) / p.numel())
            err = float((p.grad.reshape(-1)[:32].double().cpu() - ref_head).abs().max())
            assert err <= 10 * TOL * scale + 1e-9, f"grad {k}: {err:.3e} vs scale {scale:.3e}"
            tot = float(p.grad.double().abs().sum())
            assert abs(tot - z[f"gw.{k}.sum"][1]) <= 1e-3 * z[f"gw.{k}.sum"][1] + 1e-9, f"grad {k}: checksum"
        else:
            assert_close_rel(p.grad, z[key], TOL, f"grad {k}", floor=1e-9)
        n += 1
    assert n > 20
    if cfg["wloss"]:
        assert_close_rel(clf.params.grad, z["gw.clf_loss.params"], TOL, "d loss / d task weights")
