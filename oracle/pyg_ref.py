"""CPU restatement of the torch_geometric operators the reference's encoders are built from.
TEST INFRASTRUCTURE ONLY.

torch_geometric (`>=2.3.0`, /root/reference/requirements.txt:4) is absent from /root/reference
and from this image, and the reference's tests hold no vector for it, so these functions
restate PyG's PUBLISHED semantics and are anchored on the reference's
call sites:
  SAGEConv / HeteroConv(aggr='sum') : models/cadence.py:147-159,174 ; models/pitch_spelling.py:18-30,45
  trim_to_layer                     : models/cadence.py:167-173 ; models/pitch_spelling.py:38-44
  HGTConv (via graphmuse HybridHGT) : models/analysis.py:445-453
PARITY UNPINNED against real torch_geometric (no vectors, package not importable).  The only
pins are hand-computed KATs (tests/test_oracle_pyg.py) and internal consistency (fused vs
per-relation forms).

Functional form: `P` maps parameter names (the product modules' state_dict keys) to tensors.
PyG convention: edge_index[0] = source j, edge_index[1] = target i.
"""
from __future__ import annotations

import math
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

EdgeType = Tuple[str, str, str]
Params = Mapping[str, torch.Tensor]


def et_key(et: EdgeType) -> str:
    """Module-dict key of an edge type (PyG's `ModuleDict` spelling: '<a___b___c>')."""
    return "<" + "___".join(et) + ">"


def segment_mean(x_src: torch.Tensor, ei: torch.Tensor, n_dst: int) -> torch.Tensor:
    out = x_src.new_zeros(n_dst, x_src.shape[1])
    if ei.shape[1] == 0:
        return out
    out = out.index_add(0, ei[1], x_src[ei[0]])
    deg = torch.zeros(n_dst, dtype=x_src.dtype).index_add(0, ei[1], torch.ones(ei.shape[1], dtype=x_src.dtype))
    return out / deg.clamp(min=1).unsqueeze(-1)


# A.2  SAGEConv defaults: out_i = lin_l(mean_{j->i} x_j) + lin_r(x_i); lin_l has bias, lin_r none
def sage_conv(P: Params, pre: str, x_src, x_dst, ei):
    agg = segment_mean(x_src, ei, x_dst.shape[0])
    return agg @ P[pre + "lin_l.weight"].t() + P[pre + "lin_l.bias"] + x_dst @ P[pre + "lin_r.weight"].t()


# A.3  HeteroConv: per edge type conv, grouped by destination type, reduced with `aggr`;
#      edge types whose source/destination type is missing are skipped; destination types that
#      receive nothing are dropped.
def hetero_conv_sage(P: Params, pre: str, edge_types: Sequence[EdgeType], x_dict, ei_dict, aggr: str = "sum"):
    buckets: Dict[str, List[torch.Tensor]] = {}
    for et in edge_types:
        s, _, d = et
        if et not in ei_dict or s not in x_dict or d not in x_dict:
            continue
        out = sage_conv(P, f"{pre}convs.{et_key(et)}.", x_dict[s], x_dict[d], ei_dict[et])
        buckets.setdefault(d, []).append(out)
    res = {}
    for d, outs in buckets.items():
        st = torch.stack(outs, dim=0)
        res[d] = st.sum(0) if aggr == "sum" else st.mean(0)
    return res


# A.5  trim_to_layer
def trim_to_layer(layer: int, nodes_per_hop, edges_per_hop, x_dict, ei_dict):
    if layer <= 0:
        return x_dict, ei_dict
    x2 = {k: v.narrow(0, 0, v.shape[0] - nodes_per_hop[k][-layer]) for k, v in x_dict.items()}
    e2 = {k: v.narrow(1, 0, v.shape[1] - edges_per_hop[k][-layer]) for k, v in ei_dict.items()}
    return x2, e2


# A.4  HGTConv (PyG >= 2.3)
def hgt_conv(P: Params, pre: str, node_types: Sequence[str], edge_types: Sequence[EdgeType], heads: int,
             x_dict, ei_dict):
    any_x = next(iter(x_dict.values()))
    out_ch = P[f"{pre}out_lin.lins.{node_types[0]}.weight"].shape[0]
    D = out_ch // heads
    k_d, q_d, v_d = {}, {}, {}
    for t, x in x_dict.items():
        kqv = x @ P[f"{pre}kqv_lin.lins.{t}.weight"].t() + P[f"{pre}kqv_lin.lins.{t}.bias"]
        k, q, v = kqv.split(out_ch, dim=1)
        k_d[t], q_d[t], v_d[t] = k.view(-1, heads, D), q.view(-1, heads, D), v.view(-1, heads, D)
    n_dst = {t: x.shape[0] for t, x in x_dict.items()}
    # logits per destination type, over ALL incoming edge types (the softmax spans edge types)
    msgs: Dict[str, List[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]] = {t: [] for t in x_dict}
    for e_idx, et in enumerate(edge_types):
        s, _, d = et
        if et not in ei_dict or s not in x_dict or d not in x_dict:
            continue
        ei = ei_dict[et]
        Wk = P[pre + "k_rel.weight"][e_idx * heads:(e_idx + 1) * heads]          # [heads, D, D]
        Wv = P[pre + "v_rel.weight"][e_idx * heads:(e_idx + 1) * heads]
        k2 = torch.einsum("nhd,hde->nhe", k_d[s], Wk)
        v2 = torch.einsum("nhd,hde->nhe", v_d[s], Wv)
        p = P[f"{pre}p_rel.{'__'.join(et)}"].view(1, heads)
        logit = (q_d[d][ei[1]] * k2[ei[0]]).sum(-1) * p / math.sqrt(D)          # [E, heads]
        msgs[d].append((ei[1], logit, v2[ei[0]]))
    out_dict = {}
    for t in x_dict:
        n = n_dst[t]
        if msgs[t]:
            dst = torch.cat([m[0] for m in msgs[t]])
            logit = torch.cat([m[1] for m in msgs[t]])
            val = torch.cat([m[2] for m in msgs[t]])
            mx = torch.full((n, heads), -float("inf"), dtype=any_x.dtype).scatter_reduce(
                0, dst.unsqueeze(-1).expand(-1, heads), logit.detach(), reduce="amax", include_self=True)
            ex = torch.exp(logit - mx[dst])
            den = torch.zeros(n, heads, dtype=any_x.dtype).index_add(0, dst, ex)
            alpha = ex / (den[dst] + 1e-16)
            m = torch.zeros(n, heads, D, dtype=any_x.dtype).index_add(0, dst, val * alpha.unsqueeze(-1))
            m = m.reshape(n, out_ch)
        else:
            m = any_x.new_zeros(n, out_ch)
        o = F.gelu(m) @ P[f"{pre}out_lin.lins.{t}.weight"].t() + P[f"{pre}out_lin.lins.{t}.bias"]
        if o.shape[-1] == x_dict[t].shape[-1]:
            beta = torch.sigmoid(P[f"{pre}skip.{t}"])
            o = beta * o + (1 - beta) * x_dict[t]
        out_dict[t] = o
    return out_dict
