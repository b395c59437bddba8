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
