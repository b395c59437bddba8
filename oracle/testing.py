"""Helpers shared by the golden-vector generator and the tests.  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import hashlib
import os
from typing import Dict

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def seeded_fill_(module: torch.nn.Module, seed: int, scale: float = 0.08, rename=None, norm_offset: float = 0.0) -> None:
    """Overwrite every parameter/buffer-free tensor of `module` deterministically from `seed`.

    Used for the large (H=256) fixtures whose weights are too big to commit: generator and
    test re-create identical weights from the seed.  Parameters are filled in
    `named_parameters()` order from one stream per parameter name (order independent).
    `rename`: maps a parameter name to the name its stream is keyed by (two modules whose names differ by a prefix get
    the same values); `norm_offset` is added to every 1-D `*.weight` (LayerNorm scales around 1 instead of around 0).
    """
    with torch.no_grad():
        for name, p in module.named_parameters():
            name = rename(name) if rename is not None else name
            h = int.from_bytes(hashlib.sha256(name.encode()).digest()[:4], "little")
            g = torch.Generator().manual_seed((seed * 1000003 + h) % (2 ** 31))
            p.copy_(torch.randn(p.shape, generator=g, dtype=torch.float32) * scale)
            if norm_offset and p.dim() == 1 and name.endswith(".weight"):
                p.add_(norm_offset)


class ReluTap:
    """Records the input of every F.relu call while active (nn.ReLU and the CPU oracle both end there): the only points
    where the model is not differentiable (dropout = 0).  `margin()` = the smallest |input| relative to the largest of its
    call; `at_risk(delta)` = how many inputs lie within delta * max|input| of the kink — an fp32 evaluation (absolute error
    ~1e-6 of the largest term) may put those on the other side, which flips a derivative and moves gradients by far more
    than rounding does."""

    def __init__(self):
        self.inputs = []

    def __enter__(self):
        import torch.nn.functional as F
        self._F, self._orig = F, F.relu

        def relu(x, *a, **k):
            self.inputs.append(x.detach())
            return self._orig(x, *a, **k)
        F.relu = relu
        return self

    def __exit__(self, *exc):
        self._F.relu = self._orig
        return False

    def margin(self) -> float:
        return min(float(x.abs().min() / x.abs().max()) for x in self.inputs if x.numel())

    def at_risk(self, delta: float = 2e-5) -> int:
        return sum(int((x.abs() < delta * x.abs().max()).sum()) for x in self.inputs if x.numel())


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


def r3_graphs() -> dict:
    """The synthetic batches of the round-3 wrapper fixtures (oracle/gen_golden_r3.py), rebuilt from seeds by generator and
    tests alike: the graph generator is the build's own (analysisgnn_amd/synth.py), deterministic in its seed."""
    from analysisgnn_amd.synth import make_batch, make_score_graph, merge_sampled, sample_hops
    return {
        "whole": make_batch(2, 80, first_seed=31),
        "sampled": merge_sampled([sample_hops(make_score_graph(seed=sd, n_notes=160), 60, (5, 5), seed=sd, first_target=10)
                                  for sd in (41, 42)]),
        "hetero": make_batch(2, 80, first_seed=51, add_beats=True, add_measures=True, reverse_metrical_edges=True),
        "one": make_batch(1, 64, first_seed=61),
    }
