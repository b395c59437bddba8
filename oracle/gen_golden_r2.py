#!/usr/bin/env python3
"""Round-2 golden vectors, again produced by RUNNING THE REFERENCE'S OWN CODE on CPU.  TEST INFRASTRUCTURE ONLY;
runs only in the build container (needs /root/reference); only the arrays it writes (tests/golden/r2_*.npz) are committed.

What is executed from the reference:
  * core/gnn.py RelEdgeConv, core/hgnn.py HeteroRelEdgeConvLayer / HeteroSageConvLayer(reduction='lstm') /
    HeteroResGatedGraphConvLayer(reduction='lstm'|'none') / HeteroAttention — loaded as files exactly as
    oracle/gen_golden.py does (torch_scatter = oracle/scatter_ref.py, the restated semantics);
  * models/analysis.py:44-101 `onsetwise_logit_aggregation` — that module cannot be imported (pytorch_lightning,
    torchmetrics, graphmuse are absent), so the FUNCTION's source segment is read from the file with `ast`, compiled
    and executed in a namespace that holds `torch` and the `torch_scatter` stand-in: the reference's statements run
    unchanged, nothing of the module around them does.  Its `graph` argument is a PyG HeteroData; a minimal
    attribute/mapping object with the four fields the function touches stands in for it.

Usage: python oracle/gen_golden_r2.py   (idempotent; seeds fixed)
"""
from __future__ import annotations

import ast
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle.gen_golden import cat_edges, load_reference_core, run_case, t  # noqa: E402
from oracle.testing import GOLDEN_DIR, seeded_randn  # noqa: E402
from analysisgnn_amd.synth import make_score_graph  # noqa: E402

REF_ANALYSIS = "/root/reference/analysisgnn/models/analysis.py"


def load_reference_function(path: str, name: str, namespace: dict):
    src = open(path).read()
    tree = ast.parse(src)
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    code = compile(ast.Module(body=[node], type_ignores=[]), path, "exec")
    exec(code, namespace)
    return namespace[name]


class _Store(dict):
    __getattr__ = dict.__getitem__


class _Graph(dict):
    """graph["note"].x / .batch / .onset_div and graph.edge_index_dict — what analysis.py:46-47,70,72 read."""

    def __init__(self, note, edge_index_dict):
        super().__init__(note=_Store(note))
        self.edge_index_dict = edge_index_dict


def main():
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(1)
    gnn, hgnn = load_reference_core()
    rels = ["onset", "consecutive", "during", "rest"]
    etypes = {r: i for i, r in enumerate(rels)}
    g = make_score_graph(seed=5, n_notes=40)
    eih, eth = cat_edges(g, rels)
    x = seeded_randn(4, 40, 8)

    # ---------------- RelEdgeConv (core/gnn.py:79-106) ------------------------------------
    e_du = t(g.edge_index[("note", "during", "note")])
    torch.manual_seed(21)
    m = gnn.RelEdgeConv(8, 12)
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.uniform_(-0.2, 0.2)
    run_case("r2_reledge", m, lambda mod, i: mod(i["x"], i["edge_index"]), {"x": x, "edge_index": e_du}, ["x"])
    torch.manual_seed(22)
    m = gnn.RelEdgeConv(8, 12, in_edge_features=5)
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.uniform_(-0.2, 0.2)
    ef = seeded_randn(6, e_du.shape[1], 5)
    run_case("r2_reledge_edgefeat", m, lambda mod, i: mod(i["x"], i["edge_index"], i["edge_features"]),
             {"x": x, "edge_index": e_du, "edge_features": ef}, ["x", "edge_features"])

    # ---------------- HeteroRelEdgeConvLayer (core/hgnn.py:66-95) -------------------------
    torch.manual_seed(23)
    m = hgnn.HeteroRelEdgeConvLayer(8, 8, etypes=etypes)
    run_case("r2_hreledge", m, lambda mod, i: mod(i["x"], i["edge_index"], i["edge_type"]),
             {"x": x, "edge_index": eih, "edge_type": eth}, ["x"], extra={"meta.rels": np.array(rels)})
    torch.manual_seed(24)
    m = hgnn.HeteroRelEdgeConvLayer(8, 8, etypes=etypes, in_edge_features=3)
    nf = seeded_randn(7, 40, 3)                               # node-level features: |f_i - f_j| per edge (:82-83)
    run_case("r2_hreledge_nodefeat", m, lambda mod, i: mod(i["x"], i["edge_index"], i["edge_type"], i["edge_features"]),
             {"x": x, "edge_index": eih, "edge_type": eth, "edge_features": nf}, ["x", "edge_features"],
             extra={"meta.rels": np.array(rels)})
    pf = seeded_randn(8, eih.shape[1], 3)                     # per-edge features (:84-85)
    run_case("r2_hreledge_edgefeat", m, lambda mod, i: mod(i["x"], i["edge_index"], i["edge_type"], i["edge_features"]),
             {"x": x, "edge_index": eih, "edge_type": eth, "edge_features": pf}, ["x", "edge_features"],
             extra={"meta.rels": np.array(rels)})

    # ---------------- the remaining reductions (core/hgnn.py:102-116, :30-46; HeteroAttention :8-23) ----
    for red in ("lstm",):          # 'concat' raises in the reference (torch.cat on a tensor, hgnn.py:112), 'max'/'min' too (:108-110)
        torch.manual_seed(25)
        m = hgnn.HeteroSageConvLayer(8, 8, etypes=etypes, reduction=red)
        run_case(f"r2_hsage_{red}", m, lambda mod, i: mod(i["x"], i["edge_index"], i["edge_type"]),
                 {"x": x, "edge_index": eih, "edge_type": eth}, ["x"], extra={"meta.rels": np.array(rels)})
    for red in ("lstm", "none"):
        torch.manual_seed(26)
        m = hgnn.HeteroResGatedGraphConvLayer(8, 8, etypes=etypes, reduction=red)
        run_case(f"r2_hresgated_{red}", m, lambda mod, i: mod(i["x"], i["edge_index"], i["edge_type"]),
                 {"x": x, "edge_index": eih, "edge_type": eth}, ["x"], extra={"meta.rels": np.array(rels)})

    # ---------------- onsetwise_logit_aggregation (models/analysis.py:44-101) -------------
    from oracle import scatter_ref
    ts = types.ModuleType("torch_scatter")
    ts.scatter_mean = scatter_ref.scatter_mean
    fn = load_reference_function(REF_ANALYSIS, "onsetwise_logit_aggregation", {"torch": torch, "torch_scatter": ts})
    g2 = make_score_graph(seed=9, n_notes=120)
    order = np.argsort(g2.onset_div, kind="stable")
    assert (order == np.arange(120)).all()                    # notes are onset-ordered, as the function assumes (:79-81)
    onset_e = t(g2.edge_index[("note", "onset", "note")])
    classes = {"quality": 15, "inversion": 4, "degree1": 22, "degree2": 22, "localkey": 50, "tpc_in_label": 2}
    rec = {}
    for tag, with_tpc, n_extra in (("plain", False, 0), ("tpc", True, 0), ("halo", False, 17)):
        bs = 120 - n_extra                                    # "halo": the graph holds 17 more notes than the predictions cover
        n = bs                                                # (rows beyond batch_size make analysis.py:68 raise: callers pass [:batch_size])
        gen = torch.Generator().manual_seed(31 + len(tag))
        probs = {}
        for k, c in classes.items():
            if k == "tpc_in_label" and not with_tpc:
                continue
            # temperature 8 on a slowly varying signal: the argmax runs are several onsets long, as in a real analysis
            base = torch.randn(12, c, generator=gen).repeat_interleave(10, dim=0)[:n]
            probs[k] = torch.softmax(8.0 * base + torch.randn(n, c, generator=gen), dim=-1)
        ins = {k: v.clone() for k, v in probs.items()}
        graph = _Graph({"x": torch.zeros(120, 1), "batch": torch.zeros(120, dtype=torch.long), "onset_div": t(g2.onset_div)},
                       {("note", "onset", "note"): onset_e})
        out = fn({k: v.clone() for k, v in probs.items()}, graph, batch_size=(bs if n_extra else None))
        for k, v in ins.items():
            rec[f"{tag}.in.{k}"] = v.numpy().copy()
        for k, v in out.items():
            rec[f"{tag}.out.{k}"] = v.numpy().copy()
        rec[f"{tag}.batch_size"] = np.int64(bs)
    rec["onset_edges"] = onset_e.numpy().copy()
    rec["onset_div"] = g2.onset_div.copy()
    path = os.path.join(GOLDEN_DIR, "r2_onsetwise_agg.npz")
    np.savez_compressed(path, **rec)
    print(f"  wrote r2_onsetwise_agg.npz ({os.path.getsize(path)/1024:.1f} KiB)")
    print("done")


if __name__ == "__main__":
    main()
