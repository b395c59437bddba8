"""CPU restatement of the encoder stacks behind `TorchAnalysisGNN.encode`.  TEST INFRASTRUCTURE ONLY.

The encoder classes (`HybridGNN`, `HybridHGT`, `MetricalGNN`) live in the third-party package
graphmuse (unpinned; /root/reference/requirements.txt:19, README.md:74), which is absent from
/root/reference and from this image.  What is restated here is the BUILD SPEC of SURVEY.md
App. A.6, assembled from the reference's in-tree analogs:
  * GNN stack      : HeteroConv{SAGEConv} per layer with trim_to_layer (models/cadence.py:142-176),
                     LayerNorm -> ReLU -> dropout between layers.
  * hybrid branch  : per-subgraph padded 2-layer bi-GRU over the TARGET notes, LayerNorm, MLP,
                     concat with the GNN output, Linear(2H -> H)   (models/cadence.py:248-303,
                     models/analysis.py:527-537).
  * wrapper        : embeddings, per-type input MLPs, encoder call, onset pool, project_enc, task
                     heads (models/analysis.py:421-591) — this part IS in the reference tree and is
                     followed line by line.
PARITY UNPINNED for the graphmuse part (no vectors exist).  eval mode only
(dropout = identity).
"""
from __future__ import annotations

from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import pyg_ref, rnn_ref
from .intree_ref import jumping_knowledge, onset_pool

Params = Mapping[str, torch.Tensor]
EdgeType = Tuple[str, str, str]


def _lin(P, name, x):
    y = x @ P[name + ".weight"].t()
    return y + P[name + ".bias"] if (name + ".bias") in P else y


def _ln(P, name, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), P[name + ".weight"], P[name + ".bias"], eps)


def hop_counts(mask_node, mask_edge, x_dict, ei_dict, num_layers):
    """Accept the three `neighbor_mask_*` conventions of SURVEY.md §8(b): per-hop count lists,
    per-element hop-index tensors (0 = target), or None (no trimming)."""
    if mask_node is None or mask_edge is None:
        return None, None

    def depth(masks):
        t = [m for m in masks.values() if isinstance(m, torch.Tensor)]
        return 1 + max((int(m.max()) if m.numel() else 0) for m in t) if t else 0
    dn, de = depth(mask_node), depth(mask_edge)

    def conv(m, d):
        if isinstance(m, torch.Tensor):
            return torch.bincount(m, minlength=d).tolist()
        return list(m)
    nodes = {k: conv(mask_node[k], dn) for k in x_dict if k in mask_node}
    edges = {k: conv(mask_edge[k], de) for k in ei_dict if k in mask_edge}
    return nodes, edges


def sage_stack(P: Params, pre: str, edge_types: Sequence[EdgeType], num_layers: int, x_dict, ei_dict,
               nodes_per_hop=None, edges_per_hop=None, aggr: str = "sum", collect: Optional[list] = None):
    for i in range(num_layers):
        if nodes_per_hop is not None:
            x_dict, ei_dict = pyg_ref.trim_to_layer(i, nodes_per_hop, edges_per_hop, x_dict, ei_dict)
        x_dict = pyg_ref.hetero_conv_sage(P, f"{pre}convs.{i}.", edge_types, x_dict, ei_dict, aggr)
        if i < num_layers - 1:
            x_dict = {k: F.relu(_ln(P, f"{pre}layer_norms.{i}", v)) for k, v in x_dict.items()}
        if collect is not None:
            collect.append(x_dict["note"])
    return x_dict


def hgt_stack(P: Params, pre: str, node_types, edge_types, heads: int, num_layers: int, x_dict, ei_dict,
              nodes_per_hop=None, edges_per_hop=None, collect: Optional[list] = None):
    for i in range(num_layers):
        if nodes_per_hop is not None:
            x_dict, ei_dict = pyg_ref.trim_to_layer(i, nodes_per_hop, edges_per_hop, x_dict, ei_dict)
        x_dict = pyg_ref.hgt_conv(P, f"{pre}convs.{i}.", node_types, edge_types, heads, x_dict, ei_dict)
        if i < num_layers - 1:
            x_dict = {k: F.relu(v) for k, v in x_dict.items()}
        if collect is not None:
            collect.append(x_dict["note"])
    return x_dict


def hybrid_branch(P: Params, pre: str, x_target, batch_target):
    """models/cadence.py:276-285 / models/analysis.py:527-537: bincount -> split -> pad ->
    2-layer bi-GRU (runs over the zero padding) -> LayerNorm -> MLP -> unpad -> cat."""
    lengths = torch.bincount(batch_target).tolist()
    T = max(lengths)
    B = len(lengths)
    pad = x_target.new_zeros(B, T, x_target.shape[1])
    o = 0
    for b, n in enumerate(lengths):
        pad[b, :n] = x_target[o:o + n]
        o += n
    y = rnn_ref.gru(P, pre + "rnn.", pad, num_layers=2, bidirectional=True)
    y = _ln(P, pre + "rnn_norm", y)
    y = _lin(P, pre + "rnn_mlp.0", y)
    y = _ln(P, pre + "rnn_mlp.2", F.relu(y))
    y = _lin(P, pre + "rnn_mlp.4", y)
    return torch.cat([y[b, :n] for b, n in enumerate(lengths)], dim=0)


def _finish_hybrid(P, pre, x_gnn, outs, x_in, batch_dict, batch_size, use_jk):
    x = x_gnn[:batch_size]
    if use_jk:
        x = jumping_knowledge(P, pre + "jk.", [o[:batch_size] for o in outs])
    z = hybrid_branch(P, pre, x_in[:batch_size], batch_dict["note"][:batch_size])
    return _lin(P, pre + "cat_proj", torch.cat([x, z], dim=-1))


def hybrid_gnn(P: Params, pre: str, metadata, num_layers: int, x_dict, ei_dict, batch_dict, batch_size: int,
               neighbor_mask_node=None, neighbor_mask_edge=None, use_jk: bool = False, aggr: str = "sum"):
    nodes, edges = hop_counts(neighbor_mask_node, neighbor_mask_edge, x_dict, ei_dict, num_layers)
    outs: list = []
    h = sage_stack(P, pre + "gnn.", metadata[1], num_layers, x_dict, ei_dict, nodes, edges, aggr, outs)
    return _finish_hybrid(P, pre, h["note"], outs, x_dict["note"], batch_dict, batch_size, use_jk)


def hybrid_hgt(P: Params, pre: str, metadata, num_layers: int, heads: int, x_dict, ei_dict, batch_dict,
               batch_size: int, neighbor_mask_node=None, neighbor_mask_edge=None, use_jk: bool = False):
    nodes, edges = hop_counts(neighbor_mask_node, neighbor_mask_edge, x_dict, ei_dict, num_layers)
    outs: list = []
    h = hgt_stack(P, pre + "gnn.", metadata[0], metadata[1], heads, num_layers, x_dict, ei_dict, nodes, edges, outs)
    return _finish_hybrid(P, pre, h["note"], outs, x_dict["note"], batch_dict, batch_size, use_jk)


def metrical_gnn(P: Params, pre: str, metadata, num_layers: int, x_dict, ei_dict, batch_size: Optional[int] = None,
                 neighbor_mask_node=None, neighbor_mask_edge=None, use_jk: bool = False, aggr: str = "sum"):
    nodes, edges = hop_counts(neighbor_mask_node, neighbor_mask_edge, x_dict, ei_dict, num_layers)
    outs: list = []
    h = sage_stack(P, pre + "gnn.", metadata[1], num_layers, x_dict, ei_dict, nodes, edges, aggr, outs)["note"]
    if batch_size is not None:
        h = h[:batch_size]
        outs = [o[:batch_size] for o in outs]
    if use_jk:
        h = jumping_knowledge(P, pre + "jk.", outs)
    y = _lin(P, pre + "mlp.0", h)
    y = _ln(P, pre + "mlp.2", F.relu(y))
    return _lin(P, pre + "mlp.4", y)


# ------------------------------------------------------------------------------------------
# models/analysis.py:571-591 TorchAnalysisGNN.encode and :546-548 forward_clf (logit_fusion=False,
# use_rnn=False — the CLI defaults, train/train_analysisgnn.py:70,97)
# ------------------------------------------------------------------------------------------
def analysis_encode(P: Params, encoder_type: str, metadata, num_layers: int, pitch_spelling, key_signature,
                    x_dict, ei_dict, batch_dict, batch_size, neighbor_mask_node=None, neighbor_mask_edge=None,
                    use_jk: bool = False, heads: int = 4, aggr: str = "sum"):
    z = {k: v for k, v in x_dict.items()}
    z["note"] = torch.cat([z["note"], P["pitch_embedding.weight"][pitch_spelling],
                           P["key_embedding.weight"][key_signature]], dim=-1)                 # :574
    h = {}
    for k in metadata[0]:                                                                      # :575, :429-443
        y = F.relu(_lin(P, f"project_dict.{k}.0", z[k]))
        h[k] = _lin(P, f"project_dict.{k}.4", _ln(P, f"project_dict.{k}.2", y))
    if encoder_type == "hybridgnn":
        x = hybrid_gnn(P, "encoder.", metadata, num_layers, h, ei_dict, batch_dict, batch_size,
                       neighbor_mask_node, neighbor_mask_edge, use_jk, aggr)
    elif encoder_type == "hgt":
        x = hybrid_hgt(P, "encoder.", metadata, num_layers, heads, h, ei_dict, batch_dict, batch_size,
                       neighbor_mask_node, neighbor_mask_edge, use_jk)
    elif encoder_type == "metricalgnn":
        x = metrical_gnn(P, "encoder.", metadata, num_layers, h, ei_dict, batch_size, neighbor_mask_node,
                         neighbor_mask_edge, use_jk, aggr)
    else:
        raise ValueError(encoder_type)
    x = onset_pool(x, ei_dict[("note", "onset", "note")], batch_size)                           # :580-587
    x = _ln(P, "project_enc.0", x)                                                              # :474-485
    x = F.relu(_lin(P, "project_enc.1", x))
    x = _ln(P, "project_enc.3", x)
    x = F.relu(_lin(P, "project_enc.5", x))
    x = _ln(P, "project_enc.7", x)
    return _lin(P, "project_enc.9", x)


def analysis_logits(P: Params, x, tasks: Sequence[str]) -> Dict[str, torch.Tensor]:
    out = {}
    for t in tasks:                                                                             # :488-493
        y = F.relu(_lin(P, f"clf_dict.{t}.0", x))
        out[t] = _lin(P, f"clf_dict.{t}.3", _ln(P, f"clf_dict.{t}.2", y))
    return out
