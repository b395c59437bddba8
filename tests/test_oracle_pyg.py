"""Known-answer tests for the PyG-semantics restatement (oracle/pyg_ref.py).  torch_geometric is absent,
so these hand-computed cases are the only pins: "parity unpinned" vs the package."""
import math

import torch

from oracle import pyg_ref as G


def test_sage_conv_kat():
    # 3 nodes, edges 0->2, 1->2 ; lin_l = 2*I (+bias 1), lin_r = -I
    x = torch.tensor([[1., 2.], [3., 4.], [10., 20.]])
    ei = torch.tensor([[0, 1], [2, 2]])
    P = {"lin_l.weight": 2 * torch.eye(2), "lin_l.bias": torch.ones(2), "lin_r.weight": -torch.eye(2)}
    out = G.sage_conv(P, "", x, x, ei)
    # node 2: mean = (2,3) -> 2*mean+1 = (5,7), minus x_2 -> (-5,-13); isolated nodes: bias - x
    assert torch.allclose(out, torch.tensor([[0., -1.], [-2., -3.], [-5., -13.]]))


def test_hetero_conv_groups_by_destination_and_drops_unreached_types():
    x = {"a": torch.ones(2, 2), "b": torch.full((3, 2), 2.0)}
    ets = [("a", "to", "b"), ("b", "self", "b")]
    ei = {ets[0]: torch.tensor([[0, 1], [0, 0]]), ets[1]: torch.tensor([[1], [2]])}
    P = {}
    for et in ets:
        k = G.et_key(et)
        P[f"convs.{k}.lin_l.weight"] = torch.eye(2)
        P[f"convs.{k}.lin_l.bias"] = torch.zeros(2)
        P[f"convs.{k}.lin_r.weight"] = torch.zeros(2, 2)
    out = G.hetero_conv_sage(P, "", ets, x, ei, aggr="sum")
    assert set(out) == {"b"}                                   # "a" receives nothing -> dropped (A.3)
    assert torch.allclose(out["b"], torch.tensor([[1., 1.], [0., 0.], [2., 2.]]))
    out_m = G.hetero_conv_sage(P, "", ets, x, ei, aggr="mean")
    assert torch.allclose(out_m["b"], out["b"] / 2)


def test_trim_to_layer_narrows_cumulatively():
    x = {"n": torch.arange(10.).view(10, 1)}
    ei = {("n", "r", "n"): torch.zeros(2, 7, dtype=torch.long)}
    nodes, edges = {"n": [4, 3, 3]}, {("n", "r", "n"): [5, 2]}
    x1, e1 = G.trim_to_layer(1, nodes, edges, x, ei)
    assert x1["n"].shape[0] == 7 and e1[("n", "r", "n")].shape[1] == 5
    x2, e2 = G.trim_to_layer(2, nodes, edges, x1, e1)
    assert x2["n"].shape[0] == 4 and e2[("n", "r", "n")].shape[1] == 0
    x0, e0 = G.trim_to_layer(0, nodes, edges, x, ei)
    assert x0["n"].shape[0] == 10


def _hgt_params(node_types, edge_types, C, heads, seed=0):
    g = torch.Generator().manual_seed(seed)
    D = C // heads
    P = {}
    for t in node_types:
        P[f"kqv_lin.lins.{t}.weight"] = torch.randn(3 * C, C, generator=g) * 0.3
        P[f"kqv_lin.lins.{t}.bias"] = torch.randn(3 * C, generator=g) * 0.1
        P[f"out_lin.lins.{t}.weight"] = torch.randn(C, C, generator=g) * 0.3
        P[f"out_lin.lins.{t}.bias"] = torch.randn(C, generator=g) * 0.1
        P[f"skip.{t}"] = torch.tensor([0.3])
    P["k_rel.weight"] = torch.randn(heads * len(edge_types), D, D, generator=g) * 0.5
    P["v_rel.weight"] = torch.randn(heads * len(edge_types), D, D, generator=g) * 0.5
    for et in edge_types:
        P["p_rel." + "__".join(et)] = torch.rand(1, heads, generator=g) + 0.5
    return P


def test_hgt_conv_against_explicit_loops():
    """Single destination node, two relations: softmax spans both relations (A.4)."""
    C, heads = 8, 2
    D = C // heads
    nts, ets = ["n"], [("n", "a", "n"), ("n", "b", "n")]
    P = _hgt_params(nts, ets, C, heads)
    x = torch.randn(4, C, generator=torch.Generator().manual_seed(1))
    ei = {ets[0]: torch.tensor([[1, 2], [0, 0]]), ets[1]: torch.tensor([[3], [0]])}
    out = G.hgt_conv(P, "", nts, ets, heads, {"n": x}, ei)["n"]
    kqv = x @ P["kqv_lin.lins.n.weight"].t() + P["kqv_lin.lins.n.bias"]
    k, q, v = kqv[:, :C].view(4, heads, D), kqv[:, C:2 * C].view(4, heads, D), kqv[:, 2 * C:].view(4, heads, D)
    msg = torch.zeros(heads, D)
    for h in range(heads):
        logits, vals = [], []
        for e_idx, (et, srcs) in enumerate(((ets[0], [1, 2]), (ets[1], [3]))):
            Wk, Wv = P["k_rel.weight"][e_idx * heads + h], P["v_rel.weight"][e_idx * heads + h]
            for j in srcs:
                logits.append(float(q[0, h] @ (k[j, h] @ Wk)) * float(P["p_rel." + "__".join(et)][0, h]) / math.sqrt(D))
                vals.append(v[j, h] @ Wv)
        a = torch.softmax(torch.tensor(logits), 0)
        msg[h] = sum(ai * vi for ai, vi in zip(a, vals))
    o = torch.nn.functional.gelu(msg.reshape(C)) @ P["out_lin.lins.n.weight"].t() + P["out_lin.lins.n.bias"]
    beta = torch.sigmoid(P["skip.n"])
    assert torch.allclose(out[0], beta * o + (1 - beta) * x[0], atol=1e-6)
    # nodes without incoming edges: message 0 -> out_lin(0) = bias, blended with the skip
    o_iso = beta * P["out_lin.lins.n.bias"] + (1 - beta) * x[1]
    assert torch.allclose(out[1], o_iso, atol=1e-6)
