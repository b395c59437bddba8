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
    than rounding does.  `keep_graph=True` keeps the inputs attached and the outputs' gradients retained (for
    `explain_by_relu_flips`)."""

    def __init__(self, keep_graph: bool = False):
        self.inputs, self.outputs, self.keep_graph = [], [], keep_graph

    def __enter__(self):
        import torch.nn.functional as F
        self._F, self._orig = F, F.relu

        def relu(x, *a, **k):
            y = self._orig(x, *a, **k)
            if self.keep_graph:
                self.inputs.append(x)
                self.outputs.append(y)
            else:
                self.inputs.append(x.detach())
            return y
        F.relu = relu
        return self

    def __exit__(self, *exc):
        self._F.relu = self._orig
        return False

    def margin(self) -> float:
        return min(float(x.detach().abs().min() / x.detach().abs().max()) for x in self.inputs if x.numel())

    def risky(self, delta: float):
        """[(call, flat index)] of the inputs within delta * max|input of that call| of the kink."""
        out = []
        for c, x in enumerate(self.inputs):
            xd = x.detach()
            if xd.numel():
                out += [(c, int(i)) for i in (xd.abs() < delta * xd.abs().max()).reshape(-1).nonzero().reshape(-1)]
        return out

    def at_risk(self, delta: float = 2e-5) -> int:
        return len(self.risky(delta))


def explain_by_relu_flips(tap: ReluTap, loss: torch.Tensor, params, grads_other, bound, deltas=(1e-6, 3e-6, 1e-5)):
    """Is `grads_other` (another evaluation's gradients of `loss` w.r.t. `params`: the fp32 HIP path) the float64 gradient with
    a few ReLU derivatives flipped, and nothing else?

    Flipping the derivative of ReLU input k of call c (mask m_k -> 1 - m_k) changes dL/dtheta by EXACTLY
        direction_k = (1 - 2 m_k) * dL/d post_c[k] * d pre_c[k] / d theta
    (the forward value does not move: relu(pre) ~ 0 either way), one backward pass through the float64 graph per candidate.
    Candidates = inputs within delta of the kink, delta widened step by step; the difference over ALL parameters at once is
    fitted by least squares, g_other - g_64 ~ sum_k b_k direction_k.  Accepted when every fitted b_k is 0 or 1 (+-0.02) and
    the residual of every tensor is within `bound(reference tensor)`.
    -> dict(ok, flips=[(call, index)], n_candidates, worst=(name index, residual, bound), coeffs)."""
    params = list(params)
    g64 = torch.autograd.grad(loss, params, retain_graph=True, allow_unused=True)
    gpost = torch.autograd.grad(loss, tap.outputs, retain_graph=True, allow_unused=True)
    used = [i for i, t in enumerate(g64) if t is not None]
    flat = lambda ts: torch.cat([ts[i].detach().double().cpu().reshape(-1) for i in used])       # noqa: E731
    ref = flat(g64)
    d = flat(grads_other) - ref
    sizes = [g64[i].numel() for i in used]
    cands, dirs, res = [], [], None
    for delta in deltas:
        for (c, i) in tap.risky(delta):
            if (c, i) in cands or gpost[c] is None:
                continue
            x = tap.inputs[c]
            mk = 1.0 if float(x.detach().reshape(-1)[i]) > 0 else 0.0
            scale = (1.0 - 2.0 * mk) * float(gpost[c].reshape(-1)[i])
            gd = torch.autograd.grad(x.reshape(-1)[i], params, retain_graph=True, allow_unused=True)
            dirs.append(scale * torch.cat([(gd[j] if gd[j] is not None else torch.zeros_like(params[j])).detach().double().reshape(-1)
                                           for j in used]))
            cands.append((c, i))
        if not dirs:
            continue
        A = torch.stack(dirs, dim=1)
        keep = A.norm(dim=0) > 0
        sol = torch.zeros(A.shape[1], dtype=torch.float64)
        if bool(keep.any()):
            sol[keep] = torch.linalg.lstsq(A[:, keep], d.unsqueeze(1)).solution.squeeze(1)
        res = d - A @ sol
        binary = bool((((sol - 0).abs() < 0.02) | ((sol - 1).abs() < 0.02)).all())
        worst, off, ok = (None, 0.0, 0.0), 0, binary
        for n, i in zip(sizes, used):
            r = float(res[off:off + n].abs().max())
            bnd = float(bound(g64[i]))
            if bnd > 0 and r / bnd > (worst[1] / worst[2] if worst[2] else -1.0):
                worst = (i, r, bnd)
            ok = ok and r <= bnd
            off += n
        if ok:
            break
    if res is None:
        return dict(ok=False, flips=[], n_candidates=0, worst=None, coeffs=[], g64=g64)
    return dict(ok=ok, flips=[cands[k] for k in range(len(cands)) if abs(float(sol[k]) - 1.0) < 0.02], n_candidates=len(cands),
                worst=worst, coeffs=[round(float(v), 3) for v in sol], g64=g64)


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
