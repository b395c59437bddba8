#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE'S OWN in-tree layers on CPU.

TEST INFRASTRUCTURE ONLY.  Runs only in the build container (needs /root/reference); the
fixtures it writes are committed, the reference never travels.

What is executed from the reference (as files, read-only, no bytecode written):
  /root/reference/analysisgnn/models/core/gnn.py   (SageConvScatter, ResGatedGraphConv,
                                                    GATConvLayer, JumpingKnowledge, GCN,
                                                    MetricalConvLayer)
  /root/reference/analysisgnn/models/core/hgnn.py  (HeteroSageConvLayer, HGCN, HeteroConv,
                                                    in-tree MetricalGNN, HResGatedConv)
They import three functions of the third-party package `torch_scatter`, which is not in this
image.  `oracle/scatter_ref.py` (a restatement of its published semantics, pinned by
hand-computed KATs) is registered under that name.  The fixtures are therefore "the
reference's layer wiring + the restated scatter semantics" — stated in DESIGN.md §Oracle.
`analysisgnn/__init__.py` is NOT executed (it needs GitPython etc.): the `core` directory is
mounted as a synthetic package.

Usage:  python oracle/gen_golden.py        (idempotent; seeds fixed)
"""
from __future__ import annotations

import importlib
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import scatter_ref  # noqa: E402
from oracle.testing import (GOLDEN_DIR, checksum, grads_to_np, seeded_fill_, seeded_randn,  # noqa: E402
                            state_to_np)
from analysisgnn_amd.synth import make_batch, make_score_graph  # noqa: E402

REF_CORE = "/root/reference/analysisgnn/models/core"


def load_reference_core():
    ts = types.ModuleType("torch_scatter")
    ts.scatter = scatter_ref.scatter
    ts.scatter_add = scatter_ref.scatter_add
    ts.scatter_sum = scatter_ref.scatter_sum
    ts.scatter_mean = scatter_ref.scatter_mean
    sys.modules["torch_scatter"] = ts
    pkg = types.ModuleType("_agnn_refcore")
    pkg.__path__ = [REF_CORE]
    sys.modules["_agnn_refcore"] = pkg
    gnn = importlib.import_module("_agnn_refcore.gnn")
    hgnn = importlib.import_module("_agnn_refcore.hgnn")
    return gnn, hgnn


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def run_case(name, module, fwd, inputs, grad_inputs, extra=None, big=False):
    """Run fwd(module, **inputs) -> out; backward with a fixed cotangent; save fixture."""
    for k in grad_inputs:
        inputs[k] = inputs[k].clone().requires_grad_(True)
    out = fwd(module, inputs)
    gout = seeded_randn(12345, *out.shape)
    module.zero_grad(set_to_none=True)
    (out * gout).sum().backward()
    rec = {}
    if big:
        rec["out.head"] = out[:16].detach().numpy().copy()
        rec["out.sum"] = checksum(out)
        for k in grad_inputs:
            rec[f"grad.{k}.head"] = inputs[k].grad[:16].numpy().copy()
            rec[f"grad.{k}.sum"] = checksum(inputs[k].grad)
        for k, p in module.named_parameters():
            rec[f"gw.{k}.sum"] = checksum(p.grad)
    else:
        rec.update(state_to_np(module))
        for k, v in inputs.items():
            if isinstance(v, torch.Tensor):
                rec[f"in.{k}"] = v.detach().numpy().copy()
        rec["out"] = out.detach().numpy().copy()
        rec["gout"] = gout.numpy().copy()
        for k in grad_inputs:
            rec[f"grad.{k}"] = inputs[k].grad.numpy().copy()
        rec.update(grads_to_np(module))
    if extra:
        rec.update(extra)
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    np.savez_compressed(path, **rec)
    print(f"  wrote {name}.npz  ({os.path.getsize(path)/1024:.1f} KiB)  out{tuple(out.shape)}")


def cat_edges(g, rels):
    """Homogeneous [2,E] + edge_type[E] from a note-only ScoreGraph, in-tree convention."""
    eis, ets = [], []
    for code, rel in enumerate(rels):
        key = ("note", rel, "note")
        if key in g.edge_index:
            e = g.edge_index[key]
            eis.append(e)
            ets.append(np.full(e.shape[1], code, dtype=np.int64))
    return t(np.concatenate(eis, axis=1)), t(np.concatenate(ets))


def main():
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(1)
    gnn, hgnn = load_reference_core()
    print("reference core loaded from", REF_CORE)

    # ---------------- F2: SageConvScatter (core/gnn.py:39-76) -----------------------------
    rng = np.random.default_rng(7)
    N, F, O = 8, 4, 6
    ei = t(rng.integers(0, N, size=(2, 14)).astype(np.int64))
    ei[:, 3] = ei[:, 2]                      # a duplicated edge
    x = seeded_randn(1, N, F)
    torch.manual_seed(1)
    m = gnn.SageConvScatter(F, O)
    with torch.no_grad():
        m.linear.bias.uniform_(-0.3, 0.3)
        m.neigh_linear.bias.uniform_(-0.3, 0.3)
    run_case("sage_small", m, lambda mod, i: mod(i["x"], i["edge_index"]),
             {"x": x, "edge_index": ei}, ["x"])
    run_case("sage_empty", m, lambda mod, i: mod(i["x"], i["edge_index"]),
             {"x": x, "edge_index": torch.zeros(2, 0, dtype=torch.long)}, ["x"])
    torch.manual_seed(2)
    m = gnn.SageConvScatter(F, O, in_edge_features=3)
    ef = seeded_randn(2, ei.shape[1], 3)
    run_case("sage_small_edgefeat", m,
             lambda mod, i: mod(i["x"], i["edge_index"], i["edge_features"]),
             {"x": x, "edge_index": ei, "edge_features": ef}, ["x", "edge_features"])
    # big: N=500, F=256, onset relation of synthetic seed 0 (weights from seeded_fill_)
    g0 = make_score_graph(seed=0, n_notes=500)
    m = gnn.SageConvScatter(256, 256)
    seeded_fill_(m, 11)
    run_case("sage_big", m, lambda mod, i: mod(i["x"], i["edge_index"]),
             {"x": seeded_randn(3, 500, 256), "edge_index": t(g0.edge_index[("note", "onset", "note")])},
             ["x"], big=True,
             extra={"meta.seed_graph": np.int64(0), "meta.seed_x": np.int64(3), "meta.seed_w": np.int64(11)})

    # ---------------- F3: HeteroSageConvLayer (core/hgnn.py:98-140) -----------------------
    g = make_score_graph(seed=5, n_notes=40)
    rels = ["onset", "consecutive", "during", "rest", "never"]      # "never" has zero edges
    etypes = {r: i for i, r in enumerate(rels)}
    eih, eth = cat_edges(g, rels)
    x = seeded_randn(4, 40, 8)
    for red in ("mean", "sum"):
        torch.manual_seed(3)
        m = hgnn.HeteroSageConvLayer(8, 8, etypes=etypes, reduction=red)
        with torch.no_grad():
            for p in m.parameters():
                if p.dim() == 1:
                    p.uniform_(-0.2, 0.2)
        run_case(f"hsage_{red}", m, lambda mod, i: mod(i["x"], i["edge_index"], i["edge_type"]),
                 {"x": x, "edge_index": eih, "edge_type": eth}, ["x"],
                 extra={"meta.rels": np.array(rels)})
    # dict input form (hgnn.py:130-133): same module, edge_index as {relation: [2,E_r]}
    eid = {r: eih[:, eth == c] for r, c in etypes.items()}
    out_dict = m(x, eid).detach().numpy()
    np.savez_compressed(os.path.join(GOLDEN_DIR, "hsage_sum_dictform.npz"), out=out_dict)

    # ---------------- F4: HGCN (core/hgnn.py:144-179) -------------------------------------
    for jk in (False, True):
        torch.manual_seed(4)
        m = hgnn.HGCN(8, 16, 8, n_layers=2, etypes={r: i for i, r in enumerate(rels[:4])},
                      dropout=0.0, jk=jk)
        run_case("hgcn3_jk" if jk else "hgcn3", m,
                 lambda mod, i: mod(i["x"], i["edge_index"], i["edge_type"]),
                 {"x": x, "edge_index": eih, "edge_type": eth}, ["x"],
                 extra={"meta.rels": np.array(rels[:4])})

    # ---------------- F5: ResGatedGraphConv (core/gnn.py:212-258) -------------------------
    torch.manual_seed(5)
    m = gnn.ResGatedGraphConv(8, 12)
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.uniform_(-0.2, 0.2)
    e_on = t(g.edge_index[("note", "during", "note")])
    run_case("resgated", m, lambda mod, i: mod(i["x"], i["edge_index"]),
             {"x": x, "edge_index": e_on}, ["x"])
    torch.manual_seed(6)
    m = gnn.ResGatedGraphConv(8, 12, in_edge_features=5)
    ef = seeded_randn(6, e_on.shape[1], 5)
    run_case("resgated_edgefeat", m, lambda mod, i: mod(i["x"], i["edge_index"], i["edge_features"]),
             {"x": x, "edge_index": e_on, "edge_features": ef}, ["x", "edge_features"])

    # ---------------- F6: GATConvLayer (core/gnn.py:154-209) ------------------------------
    torch.manual_seed(7)
    m = gnn.GATConvLayer(8, 10, num_heads=3, dropout=0.0)
    run_case("gat", m, lambda mod, i: mod(i["x"], i["edge_index"]),
             {"x": x, "edge_index": t(g.edge_index[("note", "onset", "note")])}, ["x"])

    # ---------------- a14: JumpingKnowledge (core/gnn.py:345-365) -------------------------
    torch.manual_seed(8)
    m = gnn.JumpingKnowledge(n_hidden=8, n_layers=3)
    run_case("jk", m, lambda mod, i: mod([i["x0"], i["x1"], i["x2"]]),
             {"x0": seeded_randn(20, 40, 8), "x1": seeded_randn(21, 40, 8), "x2": seeded_randn(22, 40, 8)},
             ["x0", "x1", "x2"])

    # ---------------- F7: in-tree MetricalGNN(metrical=True) (core/hgnn.py:323-433) -------
    def metrical_inputs(gb):
        eih_, eth_ = cat_edges(gb, rels[:4])
        nb, nm = gb.num_nodes["beat"], gb.num_nodes["measure"]
        return {
            "x": seeded_randn(9, gb.num_nodes["note"], 8),
            "edge_index": eih_, "edge_type": eth_,
            "beat_nodes": torch.arange(nb), "measure_nodes": torch.arange(nm),
            "beat_edges": t(gb.edge_index[("note", "connects", "beat")]),
            "measure_edges": t(gb.edge_index[("note", "connects", "measure")]),
        }

    def metrical_fwd(mod, i):
        return mod(i["x"], i["edge_index"], i["edge_type"], i["beat_nodes"], i["measure_nodes"],
                   i["beat_edges"], i["measure_edges"], beat_lengths=i["beat_lengths"],
                   measure_lengths=i["measure_lengths"])

    g1 = make_score_graph(seed=5, n_notes=40, add_beats=True, add_measures=True)
        # Equal-length branch (gnn.py:518-521,537-538): the reference's `h.view(-1, H)` after the two
    # einsum transposes raises for more than one sequence (ordinary RuntimeError, non-contiguous
    # view), so the only runnable equal-length case is ONE sequence with lengths=None (gnn.py:507-508).
    gb_eq = g1
    gb_rg = make_batch(3, n_notes=36, first_seed=20, add_beats=True, add_measures=True)  # ragged
    for tag, gb in (("eq", gb_eq), ("ragged", gb_rg)):
        ins = metrical_inputs(gb)
        nb_per = np.bincount(gb.batch["beat"])
        nm_per = np.bincount(gb.batch["measure"])
        if tag == "eq":
            ins["beat_lengths"] = None
            ins["measure_lengths"] = None
        else:                                        # gnn.py:514: cumulative boundaries, diff = lengths
            ins["beat_lengths"] = t(np.concatenate([[0], np.cumsum(nb_per)]).astype(np.int64))
            ins["measure_lengths"] = t(np.concatenate([[0], np.cumsum(nm_per)]).astype(np.int64))
        for mode in ("train", "eval"):
            torch.manual_seed(9)
            m = hgnn.MetricalGNN(8, 8, 8, etypes={r: i for i, r in enumerate(rels[:4])}, num_layers=3,
                                 dropout=0.0, metrical=True)
            # non-trivial BatchNorm running stats so eval mode is not the identity
            with torch.no_grad():
                for mod in m.modules():
                    if isinstance(mod, torch.nn.BatchNorm1d):
                        mod.running_mean.uniform_(-0.1, 0.1)
                        mod.running_var.uniform_(0.5, 1.5)
                        mod.weight.uniform_(0.5, 1.5)
                        mod.bias.uniform_(-0.2, 0.2)
            m.train(mode == "train")
            run_case(f"metrical_{tag}_{mode}", m, metrical_fwd, dict(ins), ["x"],
                     extra={"meta.rels": np.array(rels[:4])})

    # ---------------- hetero wrapper HeteroConv (core/hgnn.py:435-484) with ResGated ------
    torch.manual_seed(10)
    m = hgnn.HeteroConv(8, 8, etypes={r: i for i, r in enumerate(rels[:4])}, module=gnn.ResGatedGraphConv)
    run_case("heteroconv_resgated", m, lambda mod, i: mod(i["x"], i["edge_index"], i["edge_type"]),
             {"x": x, "edge_index": eih, "edge_type": eth}, ["x"], extra={"meta.rels": np.array(rels[:4])})
    print("done")


if __name__ == "__main__":
    main()
