#!/usr/bin/env python3
"""Round-3 golden vectors: the WRAPPER around the encoders, produced by RUNNING THE REFERENCE'S OWN CLASS SOURCE on CPU.
TEST INFRASTRUCTURE ONLY; runs only in the build container (needs /root/reference); only the arrays it writes
(tests/golden/r3_*.npz) are committed — the reference never travels.

What is executed from the reference (its statements, unchanged):
  * models/analysis.py:408-418 `CrossTaskTransformer` and :421-602 `TorchAnalysisGNN` (constructor, `encode`: embeddings cat,
    input MLPs, encoder call, onset pool, `project_enc`; `forward_clf`: 21-head block, logit fusion; `forward`);
  * models/chord.py:16-49 `MultiTaskLoss` with `nn.CrossEntropyLoss(ignore_index=-1, label_smoothing=0.1)` per task, as
    models/analysis.py:881-908 builds it.
Neither module can be imported (pytorch_lightning, torchmetrics, graphmuse, torch_scatter, torch_sparse are absent), so the
CLASS definitions are cut out of the files with `ast`, compiled and executed in a namespace that holds `torch`, `nn`, `F`,
the `torch_scatter` stand-in (oracle/scatter_ref.py) and — in the slot of the three graphmuse encoders, which exist
nowhere offline — thin `nn.Module`s around the CPU restatement oracle/encoders_ref.py.  So these fixtures pin the
reference's wrapper exactly; the encoder inside stays "parity unpinned"  and the scatter semantics are the
stand-in's.  The objective's composition (`/ len(labels_dict)`, `+ lambda_featl * x.pow(2).mean()`, analysis.py:984,
:1034-1036, :1072) sits inside a LightningModule method and is restated in `total_loss` below, line by line.

Everything runs in float64 (the fixtures are the truth the fp32 HIP path is held against, tolerance 1e-4).
Usage: python oracle/gen_golden_r3.py   (idempotent; seeds fixed)
"""
from __future__ import annotations

import ast
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import encoders_ref as E  # noqa: E402
from oracle import scatter_ref  # noqa: E402
from oracle.testing import GOLDEN_DIR, ReluTap, checksum, r3_graphs, seeded_fill_  # noqa: E402
from analysisgnn_amd.synth import torch_inputs  # noqa: E402

REF_ANALYSIS = "/root/reference/analysisgnn/models/analysis.py"
REF_CHORD = "/root/reference/analysisgnn/models/chord.py"


def load_reference_classes(path: str, names, namespace: dict) -> dict:
    """Execute the ClassDef nodes `names` of the file at `path` (in file order) inside `namespace`."""
    tree = ast.parse(open(path).read())
    nodes = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in names]
    assert [n.name for n in nodes] == [n for n in [c.name for c in tree.body if isinstance(c, ast.ClassDef)] if n in names]
    assert len(nodes) == len(names), f"{path}: {set(names) - {n.name for n in nodes}} not found"
    exec(compile(ast.Module(body=nodes, type_ignores=[]), path, "exec"), namespace)
    return namespace


class _OracleEncoder(nn.Module):
    """In the slot of a graphmuse encoder: the constructor keywords the reference passes (analysis.py:445-473), the forward
    keywords it calls with (:576-579).  Parameters are created by the BUILD's module of that name (never run here: it has no
    CPU path) so that names / shapes / initialisation are the ones the HIP path loads; forward = oracle/encoders_ref.py."""
    kind = ""

    def __init__(self, **kw):
        super().__init__()
        from analysisgnn_amd import encoders, hgt
        cls = {"hybridgnn": encoders.HybridGNN, "hgt": hgt.HybridHGT, "metricalgnn": encoders.MetricalGNN}[self.kind]
        self.impl = cls(**kw)
        self.kw = dict(kw)

    def forward(self, x_dict, edge_index_dict, batch_dict, batch_size, neighbor_mask_node, neighbor_mask_edge,
                return_edge_index=False, edge_attr_dict=None):
        P = {"encoder." + k: v for k, v in self.impl.named_parameters()}
        md, L, jk = self.kw["metadata"], self.kw["num_layers"], self.kw.get("use_jk", False)
        if self.kind == "hybridgnn":
            return E.hybrid_gnn(P, "encoder.", md, L, x_dict, edge_index_dict, batch_dict, batch_size, neighbor_mask_node,
                                neighbor_mask_edge, jk)
        if self.kind == "hgt":
            return E.hybrid_hgt(P, "encoder.", md, L, self.kw.get("heads", 4), x_dict, edge_index_dict, batch_dict, batch_size,
                                neighbor_mask_node, neighbor_mask_edge, jk)
        return E.metrical_gnn(P, "encoder.", md, L, x_dict, edge_index_dict, batch_size, neighbor_mask_node, neighbor_mask_edge, jk)


def reference_namespace() -> dict:
    ts = types.ModuleType("torch_scatter")
    ts.scatter = scatter_ref.scatter
    ts.scatter_add = scatter_ref.scatter_add
    ts.scatter_mean = scatter_ref.scatter_mean
    ns = {"torch": torch, "nn": nn, "F": F, "torch_scatter": ts,
          "HybridGNN": type("HybridGNN", (_OracleEncoder,), {"kind": "hybridgnn"}),
          "HybridHGT": type("HybridHGT", (_OracleEncoder,), {"kind": "hgt"}),
          "MetricalGNN": type("MetricalGNN", (_OracleEncoder,), {"kind": "metricalgnn"})}
    load_reference_classes(REF_ANALYSIS, ["CrossTaskTransformer", "TorchAnalysisGNN"], ns)
    load_reference_classes(REF_CHORD, ["MultiTaskLoss"], ns)
    return ns


def build_names(sd_key: str) -> str:
    """reference-side state_dict key -> the build's (the stand-in keeps the encoder's parameters one attribute deeper)."""
    return sd_key.replace("encoder.impl.", "encoder.")


def total_loss(model, clf_loss, I, labels, tasks, lambda_featl=0.1):
    """models/analysis.py common_step, the lines that touch the path: :973-984 encode + feature loss, :1031-1036 heads +
    `clf_loss` + `/ len(labels_dict)`, :1072 `+ feature_loss * lambda_featl` (no memories, no edge loss, no SMOTE)."""
    x = model.encode(pitch_spelling=I["pitch_spelling"], key_signature=I["key_signature"], x_dict=I["x_dict"],
                     edge_index_dict=I["edge_index_dict"], batch_dict=I["batch_dict"], batch_size=I["batch_size"],
                     neighbor_mask_node=I["neighbor_mask_node"], neighbor_mask_edge=I["neighbor_mask_edge"])
    feature_loss = x.pow(2).mean()
    logits_dict = model.forward_clf(x)
    labels_dict = {t: labels[i] for i, t in enumerate(tasks)}
    logits_dict = {k: logits_dict[k] for k in labels_dict.keys()}
    loss_dict = clf_loss(logits_dict, labels_dict)
    total = loss_dict.pop("total") / len(labels_dict.keys())
    total = total + feature_loss * lambda_featl
    return x, logits_dict, loss_dict, total


def make_case(ns, name, enc, gname, tasks, H, OUT, L, fusion, use_jk, seed, wloss=True, big=False, in_ch=25):
    """A ReLU's derivative at 0 is a convention and within fp32 rounding of 0 it is a coin toss: a gradient fixture that sits on
    such a point pins nothing (one flipped derivative among 160 notes moves every encoder gradient by ~2e-4 — measured, see
    profiles/r03_parity_notes.md).  So the fixture's float64 run records every ReLU input and the weights are re-drawn
    (seed + 1000, ...) until none lies within 3e-6 of its call's largest magnitude (~15x the fp32 path's absolute error there); the margin reached is stored
    (`meta.relu_margin`).  What flips DO to the HIP path's gradients is measured where the oracle can be re-run:
    tests/test_gpu_c5.py, four seeds, flips counted."""
    for attempt in range(80):
        rec, margin, risk = _run_case(ns, enc, gname, tasks, H, OUT, L, fusion, use_jk, seed + 1000 * attempt, wloss, big, in_ch)
        if risk == 0:
            break
        print(f"  {name}: seed {seed + 1000 * attempt}: {risk} ReLU inputs within 3e-6 of the kink, re-drawing")
    assert risk == 0
    rec["meta.relu_margin"] = np.array(margin)
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    np.savez_compressed(path, **rec)
    print(f"  wrote {name}.npz ({os.path.getsize(path) / 1024:.1f} KiB)  seed {int(rec['meta.cfg'][7])}  total loss {float(rec['loss.total']):.6f}  "
          f"ReLU margin {margin:.1e}")


def _run_case(ns, enc, gname, tasks, H, OUT, L, fusion, use_jk, seed, wloss, big, in_ch):
    torch.manual_seed(seed)
    graph = r3_graphs()[gname]
    md = graph.metadata()
    model = ns["TorchAnalysisGNN"](md, in_ch, H, OUT, tasks, L, dropout=0.0, use_jk=use_jk, logit_fusion=fusion,
                                   use_rnn=False, encoder_type=enc)
    loss_ft = nn.ModuleDict({t: nn.CrossEntropyLoss(ignore_index=-1, label_smoothing=0.1) for t in tasks})   # analysis.py:881-888
    clf_loss = ns["MultiTaskLoss"](tasks=list(tasks), loss_ft=loss_ft, requires_grad=wloss)                   # :899-908
    if big:
        # H = 256 weights are too large to commit: generator and test refill them from the seed (LayerNorm scales around 1)
        seeded_fill_(model, seed, rename=build_names, norm_offset=1.0)
    else:
        with torch.no_grad():                # biases / LayerNorm shifts away from their zero initialisation
            g = torch.Generator().manual_seed(seed + 1)
            for p in model.parameters():
                if p.dim() == 1:
                    p.add_(torch.randn(p.shape, generator=g) * 0.1)
    if wloss:
        with torch.no_grad():
            clf_loss.params.copy_(1.0 + 0.25 * torch.randn(len(tasks), generator=torch.Generator().manual_seed(seed + 2)))
    model.double().train()
    I = torch_inputs(graph, in_channels=in_ch, seed=seed + 3)
    I["x_dict"] = {k: v.double().requires_grad_(k == "note") for k, v in I["x_dict"].items()}
    bs = I["batch_size"]
    gl = torch.Generator().manual_seed(seed + 4)
    labels = torch.stack([torch.randint(0, c, (bs,), generator=gl) for c in tasks.values()])
    labels[torch.rand(labels.shape, generator=gl) < 0.15] = -1                       # ignore_index rows (analysis.py:883)
    with ReluTap() as tap:
        x, logits, per_task, total = total_loss(model, clf_loss, I, labels, list(tasks))
    total.backward()
    rec = {"meta.tasks": np.array(list(tasks)), "meta.classes": np.array(list(tasks.values()), dtype=np.int64),
           "meta.cfg": np.array([H, OUT, L, int(fusion), int(use_jk), int(wloss), in_ch, seed, int(big)], dtype=np.int64),
           "meta.encoder": np.array(enc), "meta.graph": np.array(gname), "in.labels": labels.numpy(), "in.batch_size": np.array(bs, dtype=np.int64),
           "loss.total": np.array(total.item()), "loss.per_task": np.array([per_task[t].item() for t in tasks])}
    if wloss:
        rec["w.clf_loss.params"] = clf_loss.params.detach().float().numpy()
        rec["gw.clf_loss.params"] = clf_loss.params.grad.numpy()
    if big:
        rec["x.head"] = x[:8].detach().numpy()
        rec["x.sum"] = checksum(x)
        for t in tasks:
            rec[f"logits.{t}.head"] = logits[t][:8].detach().numpy()
            rec[f"logits.{t}.sum"] = checksum(logits[t])
        rec["grad.x_note.head"] = I["x_dict"]["note"].grad[:8].numpy()
        rec["grad.x_note.sum"] = checksum(I["x_dict"]["note"].grad)
        for k, p in model.named_parameters():
            if p.grad is not None:
                rec[f"gw.{build_names(k)}.sum"] = checksum(p.grad)
                rec[f"gw.{build_names(k)}.head"] = p.grad.reshape(-1)[:32].float().numpy()
    else:
        for k, v in model.state_dict().items():
            rec["w." + build_names(k)] = v.detach().float().numpy()                  # the fp32 values the HIP path loads
        rec["in.x_note"] = I["x_dict"]["note"].detach().float().numpy()
        for k, v in I["x_dict"].items():
            if k != "note":
                rec[f"in.x_{k}"] = v.detach().float().numpy()
        rec["in.pitch_spelling"], rec["in.key_signature"] = I["pitch_spelling"].numpy(), I["key_signature"].numpy()
        rec["x"] = x.detach().numpy()
        for t in tasks:
            rec[f"logits.{t}"] = logits[t].detach().numpy()
        rec["grad.x_note"] = I["x_dict"]["note"].grad.numpy()
        for k, p in model.named_parameters():
            if p.grad is not None:
                rec["gw." + build_names(k)] = p.grad.float().numpy()     # fp32 storage of the float64 result: 6e-8 relative
    return rec, tap.margin(), tap.at_risk(3e-6)


SMALL_TASKS = {"cadence": 4, "localkey": 50, "romanNumeral": 185, "hrythm": 2}
# the 21 heads of train/train_analysisgnn.py:22-45 are rebuilt in the big case from bench.TASK_DICT by the tests as well


def main():
    only = sys.argv[1:]
    if only:                     # regenerate the named cases only
        global make_case
        _mk = make_case

        def make_case(ns, name, *a, **k):
            if name in only:
                _mk(ns, name, *a, **k)
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    torch.set_num_threads(4)
    ns = reference_namespace()
    make_case(ns, "r3_wrapper_hybrid_fusion", "hybridgnn", "whole", SMALL_TASKS, 32, 32, 2, True, False, 101)
    make_case(ns, "r3_wrapper_hybrid_plain_sampled", "hybridgnn", "sampled", SMALL_TASKS, 32, 32, 3, False, False, 102)
    make_case(ns, "r3_wrapper_hybrid_jk_sum", "hybridgnn", "whole", SMALL_TASKS, 32, 16, 2, True, True, 103, wloss=False)
    make_case(ns, "r3_wrapper_hgt_fusion", "hgt", "hetero", SMALL_TASKS, 32, 32, 2, True, False, 104)
    make_case(ns, "r3_wrapper_metrical_plain", "metricalgnn", "whole", {"cadence": 4, "localkey": 50, "romanNumeral": 185}, 32, 32, 3,
              False, False, 105)
    import bench
    make_case(ns, "r3_wrapper_c2_h256", "hybridgnn", "sampled", dict(bench.TASK_DICT), 256, 128, 3, False, False, 106, big=True)
    make_case(ns, "r3_wrapper_c2_h256_fusion", "hybridgnn", "one", dict(bench.TASK_DICT), 256, 128, 3, True, False, 107, big=True)


if __name__ == "__main__":
    main()
