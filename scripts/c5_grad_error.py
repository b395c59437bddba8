#!/usr/bin/env python3
"""How far are the C5-shape parameter gradients from a float64 evaluation of the CPU restatement — for the HIP path with the
one-GEMM SAGE layers, the HIP path with two GEMMs per layer, and the float32 CPU restatement itself?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from analysisgnn_amd import ops  # noqa: E402
from analysisgnn_amd.models import TorchAnalysisGNN  # noqa: E402
from analysisgnn_amd.synth import make_sampled_batch, torch_inputs  # noqa: E402
from oracle import encoders_ref as E  # noqa: E402

SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 4
TASKS = {"cadence": 4, "localkey": 50, "romanNumeral": 185}
g = make_sampled_batch(4, 500, (5, 5, 5), first_seed=40)
H, L = 512, 4
I = torch_inputs(g, in_channels=25, seed=5)


def cpu(dtype):
    torch.manual_seed(SEED)
    m = TorchAnalysisGNN(g.metadata(), 25, H, 128, TASKS, L, dropout=0.0, use_jk=False, logit_fusion=False, encoder_type="metricalgnn").train()
    P = {k: v.detach().clone().to(dtype if v.is_floating_point() else v.dtype).requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    xd = {k: v.to(dtype) for k, v in I["x_dict"].items()}
    x = E.analysis_encode(P, "metricalgnn", g.metadata(), L, I["pitch_spelling"], I["key_signature"], xd, I["edge_index_dict"], I["batch_dict"],
                          I["batch_size"], I["neighbor_mask_node"], I["neighbor_mask_edge"])
    out = E.analysis_logits(P, x, list(TASKS))
    sum((v ** 2).mean() for v in out.values()).backward()
    return {k: v.grad.double() for k, v in P.items() if v.grad is not None}


def gpu(widths):
    ops.ROOT_WIDTHS = widths
    torch.manual_seed(SEED)
    m = TorchAnalysisGNN(g.metadata(), 25, H, 128, TASKS, L, dropout=0.0, use_jk=False, logit_fusion=False, encoder_type="metricalgnn").train().to("cuda:0")
    J = torch_inputs(g, in_channels=25, seed=5, device="cuda:0")
    out = m(J["pitch_spelling"], J["key_signature"], J["x_dict"], J["edge_index_dict"], J["batch_dict"], J["batch_size"], J["neighbor_mask_node"],
            J["neighbor_mask_edge"])
    sum((v ** 2).mean() for v in out.values()).backward()
    return {k: p.grad.double().cpu() for k, p in m.named_parameters() if p.grad is not None}


ref = cpu(torch.float64)
cands = {"cpu fp32": cpu(torch.float32), "hip one-GEMM": gpu((256, 512)), "hip two-GEMM": gpu(())}
worst = {}
for name, G in cands.items():
    errs = {k: float((G[k] - ref[k]).abs().max()) / max(1.0, float(ref[k].abs().max())) for k in ref if k in G}
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
    print(f"{name:14s} max err {max(errs.values()):.3e}   worst: " + ", ".join(f"{k} {v:.2e}" for k, v in top))
