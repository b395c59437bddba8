"""Helpers shared by the golden-vector generator and the tests.  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import hashlib
import os
from typing import Dict

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def seeded_fill_(module: torch.nn.Module, seed: int, scale: float = 0.08) -> None:
    """Overwrite every parameter/buffer-free tensor of `module` deterministically from `seed`.

    Used for the large (H=256) fixtures whose weights are too big to commit: generator and
    test re-create identical weights from the seed.  Parameters are filled in
    `named_parameters()` order from one stream per parameter name (order independent).
    """
    with torch.no_grad():
        for name, p in module.named_parameters():
            h = int.from_bytes(hashlib.sha256(name.encode()).digest()[:4], "little")
            g = torch.Generator().manual_seed((seed * 1000003 + h) % (2 ** 31))
            p.copy_(torch.randn(p.shape, generator=g, dtype=torch.float32) * scale)


def seeded_randn(seed: int, *shape: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float32)


def state_to_np(module: torch.nn.Module, prefix: str = "w.") -> Dict[str, np.ndarray]:
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def load_state_from_npz(module: torch.nn.Module, z, prefix: str = "w.") -> None:
    sd = {}
    for k in z.files:
        if k.startswith(prefix):
            sd[k[len(prefix):]] = torch.from_numpy(np.asarray(z[k]))
    missing, unexpected = module.load_state_dict(sd, strict=True)
    assert not missing and not unexpected


def grads_to_np(module: torch.nn.Module, prefix: str = "gw.") -> Dict[str, np.ndarray]:
    out = {}
    for k, p in module.named_parameters():
        if p.grad is not None:
            out[prefix + k] = p.grad.detach().cpu().numpy().copy()
    return out


def checksum(t: torch.Tensor) -> np.ndarray:
    """Order-stable float64 summary of a tensor: [sum, sum|x|, sum x^2]."""
    d = t.detach().double().cpu()
    return np.asarray([d.sum().item(), d.abs().sum().item(), (d * d).sum().item()], dtype=np.float64)


def golden_path(name: str) -> str:
    return os.path.join(GOLDEN_DIR, name)
