"""Shared test helpers (CPU side)."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def params_from_npz(z, requires_grad=True, device="cpu"):
    """'w.<name>' arrays -> dict of leaf tensors."""
    P = {}
    for k in z.files:
        if k.startswith("w."):
            t = torch.from_numpy(np.asarray(z[k])).to(device)
            if t.is_floating_point() and requires_grad:
                t.requires_grad_(True)
            P[k[2:]] = t
    return P


def inputs_from_npz(z, grad_keys=(), device="cpu"):
    I = {}
    for k in z.files:
        if k.startswith("in."):
            t = torch.from_numpy(np.asarray(z[k])).to(device)
            if k[3:] in grad_keys:
                t.requires_grad_(True)
            I[k[3:]] = t
    return I


def assert_close(a, b, tol, what=""):
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    scale = max(1.0, float(b.abs().max())) if b.numel() else 1.0
    err = float((a - b).abs().max()) if b.numel() else 0.0
    assert err <= tol * scale, f"{what}: max abs err {err:.3e} > {tol:.1e} * {scale:.3g}"


def assert_close_rel(a, b, tol, what="", floor=1e-7):
    """max |a - b| <= tol * max |b| (+ floor): the tolerance is relative to the tensor's OWN largest magnitude,
    not to max(1, .) as `assert_close` — small-valued gradients are held to the same relative bound as O(1) logits."""
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    if not b.numel():
        return
    scale = float(b.abs().max())
    err = float((a - b).abs().max())
    assert err <= tol * scale + floor, f"{what}: max abs err {err:.3e} > {tol:.1e} * max|ref| {scale:.3g}"


R3_CASES = ["r3_wrapper_hybrid_fusion", "r3_wrapper_hybrid_plain_sampled", "r3_wrapper_hybrid_jk_sum", "r3_wrapper_hgt_fusion",
            "r3_wrapper_metrical_plain", "r3_wrapper_c2_h256", "r3_wrapper_c2_h256_fusion"]


def r3_case(name):
    """A round-3 wrapper fixture (oracle/gen_golden_r3.py: the reference's own TorchAnalysisGNN / MultiTaskLoss class source run
    in float64) -> (npz, cfg dict, graph, inputs on CPU, labels): everything the generator fed the reference, rebuilt."""
    from analysisgnn_amd.synth import torch_inputs
    from oracle.testing import r3_graphs
    z = load_golden(name)
    H, OUT, L, fusion, use_jk, wloss, in_ch, seed, big = (int(v) for v in z["meta.cfg"])
    cfg = dict(H=H, OUT=OUT, L=L, fusion=bool(fusion), use_jk=bool(use_jk), wloss=bool(wloss), in_ch=in_ch, seed=seed, big=bool(big),
               enc=str(z["meta.encoder"]), tasks={str(t): int(c) for t, c in zip(z["meta.tasks"], z["meta.classes"])})
    g = r3_graphs()[str(z["meta.graph"])]
    I = torch_inputs(g, in_channels=in_ch, seed=seed + 3)
    labels = torch.from_numpy(np.asarray(z["in.labels"]))
    if not big:
        assert np.array_equal(I["x_dict"]["note"].numpy(), z["in.x_note"]) and np.array_equal(I["pitch_spelling"].numpy(), z["in.pitch_spelling"])
    return z, cfg, g, I, labels


def assert_grads_close_or_relu_flips(names, hip_grads, tap64, loss64, plist, tol=1e-4, label="", max_flips=8):
    """Gradients of the fp32 HIP path (`hip_grads`, in the order of `plist`) against the float64 graph that produced `loss64`
    under `oracle.testing.ReluTap(keep_graph=True)`: each tensor within tol * max(1, |ref|max) — directly, or after removing
    the exactly computed effect of ReLU derivatives that the fp32 evaluation flipped (inputs within rounding of the kink;
    oracle.testing.explain_by_relu_flips, profiles/r03_parity_notes.md).  The flips found are printed, never tolerated blindly:
    the deviation over ALL tensors at once must be a 0/1 combination of the flip directions."""
    import torch
    from oracle.testing import explain_by_relu_flips
    g64 = torch.autograd.grad(loss64, plist, retain_graph=True, allow_unused=True)
    bound = lambda ref: tol * max(1.0, float(ref.abs().max()))                        # noqa: E731
    beyond = []
    for k, gh, gr in zip(names, hip_grads, g64):
        if gr is None:
            assert gh is None or float(gh.abs().max()) == 0.0, k
            continue
        err = float((gh.detach().cpu().double() - gr).abs().max())
        if err > bound(gr):
            beyond.append((k, f"{err:.1e}"))
    if not beyond:
        return []
    hip = [gh if gh is not None else torch.zeros_like(p) for gh, p in zip(hip_grads, plist)]
    rep = explain_by_relu_flips(tap64, loss64, plist, hip, bound)
    print(f"[{label}] {len(beyond)} gradient tensors beyond {tol:g} of float64 (e.g. {beyond[:3]}); {rep['n_candidates']} ReLU inputs near "
          f"the kink examined, flipped on the HIP path: {rep['flips']}; worst residual after removing them: {rep['worst']}")
    assert rep["ok"], f"{label}: gradient deviation is not explained by flipped ReLU derivatives: {rep['worst']}, coefficients {rep['coeffs']}"
    assert 0 < len(rep["flips"]) <= max_flips, rep["flips"]
    return rep["flips"]
