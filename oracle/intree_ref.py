"""CPU restatement of the reference's in-tree message-passing layers.  TEST INFRASTRUCTURE ONLY.

Functional form: every function takes a flat mapping `P` of parameter name -> tensor (the
reference module's `state_dict()` keys, optionally under a `pre` prefix) and plain tensors.
Each function cites the reference lines it follows.  In the in-tree layers messages flow from
`edge_index[1]` to `edge_index[0]` (core/gnn.py:70,74) — the opposite of PyG.

PARITY: pinned against fixtures produced by running the reference's own files
(oracle/gen_golden.py -> tests/golden/*.npz; checked in tests/test_oracle_intree.py), modulo
the `torch_scatter` semantics restated in oracle/scatter_ref.py.
"""
from __future__ import annotations

from typing import Dict, List, Mapping, Optional, Sequence

import torch
import torch.nn.functional as F

from . import rnn_ref

Params = Mapping[str, torch.Tensor]


def _lin(P: Params, name: str, x: torch.Tensor) -> torch.Tensor:
    y = x @ P[name + ".weight"].t()
    b = P.get(name + ".bias") if hasattr(P, "get") else (P[name + ".bias"] if name + ".bias" in P else None)
    return y if b is None else y + b


def _segment_sum(msg: torch.Tensor, dst: torch.Tensor, n: int) -> torch.Tensor:
    out = msg.new_zeros((n,) + tuple(msg.shape[1:]))
    return out.index_add(0, dst, msg) if dst.numel() else out


def _degree(dst: torch.Tensor, n: int, dtype) -> torch.Tensor:
    d = torch.zeros(n, dtype=dtype)
    if dst.numel():
        d = d.index_add(0, dst, torch.ones(dst.numel(), dtype=dtype))
    return d


# ------------------------------------------------------------------------------------------
# core/gnn.py:62-76  SageConvScatter.forward
#   h = W_n x + b_n ; empty E: z = W[x || h] + b ; else s_i = (x_i + sum_{(i,j)} h_j [+ W_e e_ij]) / max(deg_i,1)
#   z = W [x || s] + b
# ------------------------------------------------------------------------------------------
def sage_conv_scatter(P: Params, pre: str, x, ei, edge_features=None):
    h = _lin(P, pre + "neigh_linear", x)
    if ei is None or ei.shape[1] == 0:
        return _lin(P, pre + "linear", torch.cat([x, h], dim=-1))
    msg = h[ei[1]]
    if edge_features is not None and (pre + "edge_linear.weight") in P:
        msg = msg + _lin(P, pre + "edge_linear", edge_features)
    n = x.shape[0]
    s = (x + _segment_sum(msg, ei[0], n)) / _degree(ei[0], n, x.dtype).clamp(min=1).unsqueeze(-1)
    return _lin(P, pre + "linear", torch.cat([x, s], dim=-1))


# ------------------------------------------------------------------------------------------
# core/gnn.py:243-258  ResGatedGraphConv.forward
#   gate_ij = sigmoid(W3 x_i + W4 x_j [+ W5 e_ij]) ; out_i = 2 * W1 x_i + sum_j gate_ij * W2 x_j
#   (scatter into out=h1.clone() then `h1 + s`  =>  the root term appears twice)
# ------------------------------------------------------------------------------------------
def res_gated_conv(P: Params, pre: str, x, ei, edge_features=None):
    h1 = _lin(P, pre + "W1", x)
    h2 = _lin(P, pre + "W2", x)
    a = _lin(P, pre + "W3", x)
    b = _lin(P, pre + "W4", x)
    pre_act = a[ei[0]] + b[ei[1]]
    if edge_features is not None and (pre + "W5.weight") in P:
        pre_act = pre_act + _lin(P, pre + "W5", edge_features)
    msg = torch.sigmoid(pre_act) * h2[ei[1]]
    return 2.0 * h1 + _segment_sum(msg, ei[0], x.shape[0])


# ------------------------------------------------------------------------------------------
# core/gnn.py:192-209  GATConvLayer.forward  (dropout 0 / eval)
#   per-edge, per-head logit = leaky_relu(<W_el x_i, a_l> + <W_er x_j, a_r>); softmax over the
#   HEAD axis (dim=1, the reference's own remark at :205), mean over heads; out_i = Wx_i + sum_j a_ij Wx_j
# ------------------------------------------------------------------------------------------
def gat_conv(P: Params, pre: str, x, ei, num_heads: int, negative_slope: float = 0.2):
    n, f = x.shape
    s_l = (_lin(P, pre + "el", x).view(n, num_heads, f) * P[pre + "attnl"]).sum(-1)   # [N, heads]
    s_r = (_lin(P, pre + "er", x).view(n, num_heads, f) * P[pre + "attnr"]).sum(-1)
    e = F.leaky_relu(s_l[ei[0]] + s_r[ei[1]], negative_slope)                          # [E, heads]
    a = torch.softmax(e, dim=1).mean(dim=1, keepdim=True)                              # [E, 1]
    h = _lin(P, pre + "linear", x)
    return h + _segment_sum(a * h[ei[1]], ei[0], n)


# ------------------------------------------------------------------------------------------
# core/gnn.py:360-365  JumpingKnowledge.forward ; core/hgnn.py:19-23 HeteroAttention
#   alpha = softmax_over_layers( att( biLSTM(stack(xs)) ) ) ; out = sum_l alpha_l x_l
# ------------------------------------------------------------------------------------------
def jumping_knowledge(P: Params, pre: str, xs: Sequence[torch.Tensor]):
    x = torch.stack(list(xs), dim=1)
    a = rnn_ref.lstm(P, pre + "lstm.", x, num_layers=1, bidirectional=True)
    a = _lin(P, pre + "att", a).squeeze(-1)
    a = torch.softmax(a, dim=-1)
    return (x * a.unsqueeze(-1)).sum(dim=1)


# ------------------------------------------------------------------------------------------
# core/hgnn.py:128-140 HeteroSageConvLayer.forward ; :479-484 HeteroConv.forward ;
# :58-63 HeteroResGatedGraphConvLayer.forward
#   out[r] = conv_r(x, edge_index[:, edge_type == code_r]) for EVERY relation slot (empty
#   relations included: SageConvScatter takes its empty branch); reduce over the R slots.
# ------------------------------------------------------------------------------------------
def hetero_layer(P: Params, pre: str, rels: Sequence[str], conv, x, ei, et, reduction: str = "mean",
                 edge_features=None):
    outs = []
    for code, rel in enumerate(rels):
        if isinstance(ei, dict):
            sub, ef = ei[rel], None
        else:
            m = et == code
            sub = ei[:, m]
            ef = edge_features[m] if edge_features is not None else None
        outs.append(conv(P, f"{pre}conv.{rel}.", x, sub, ef))
    stack = torch.stack(outs, dim=0)
    if reduction == "mean":
        return stack.mean(dim=0)
    if reduction == "sum":
        return stack.sum(dim=0)
    raise NotImplementedError(reduction)


# ------------------------------------------------------------------------------------------
# core/hgnn.py:167-179  HGCN.forward (n_total = n_layers + 1 hetero SAGE layers)
#   hidden layers: conv -> relu -> L2 normalize -> dropout(0) ; optional JK over hidden outputs ; last conv
# ------------------------------------------------------------------------------------------
def hgcn(P: Params, rels: Sequence[str], n_total: int, x, ei, et, jk: bool = False):
    h, hs = x, []
    for l in range(n_total - 1):
        h = hetero_layer(P, f"layers.{l}.", rels, sage_conv_scatter, h, ei, et)
        h = F.normalize(F.relu(h))
        hs.append(h)
    if jk:
        h = jumping_knowledge(P, "jk.", hs)
    return hetero_layer(P, f"layers.{n_total - 1}.", rels, sage_conv_scatter, h, ei, et)


# ------------------------------------------------------------------------------------------
# core/gnn.py:506-540  MetricalConvLayer.forward
# ------------------------------------------------------------------------------------------
def _batchnorm_bt(P: Params, pre: str, h: torch.Tensor, training: bool, eps: float = 1e-5):
    """BatchNorm1d over channel dim of h [B,T,C] exactly as applied to its [B,C,T] transpose
    (gnn.py:528-531): statistics over all B*T positions, padded ones included."""
    if training:
        flat = h.reshape(-1, h.shape[-1])
        mean = flat.mean(dim=0)
        var = flat.var(dim=0, unbiased=False)
    else:
        mean, var = P[pre + "running_mean"], P[pre + "running_var"]
    return (h - mean) / torch.sqrt(var + eps) * P[pre + "weight"] + P[pre + "bias"]


def metrical_conv_layer(P: Params, pre: str, x_metrical, x, edges, lengths, training: bool):
    """edges[0] = note index, edges[1] = beat/measure index.  Returns (out_notes, h_metrical)."""
    nm, in_dim = x_metrical.shape[0], x.shape[1]
    if lengths is None:
        lengths = torch.tensor([nm], dtype=torch.long)
    ragged = not bool(torch.all(lengths == lengths[0]))
    hn = _lin(P, pre + "neigh", x)
    agg = _segment_sum(hn[edges[0]], edges[1], nm)                     # notes -> beats (sum)  :511
    zs = torch.cat([agg, x_metrical], dim=-1)
    if ragged:                                                          # :512-517 cumulative boundaries
        sizes = torch.diff(lengths).tolist()
        T = max(sizes)
        B = len(sizes)
        seq = agg.new_zeros(B, T, in_dim)
        zpad = zs.new_zeros(B, T, zs.shape[1])
        o = 0
        for b, s in enumerate(sizes):
            seq[b, :s] = agg[o:o + s]
            zpad[b, :s] = zs[o:o + s]
            o += s
        zs = zpad
    else:                                                               # :518-521 equal lengths
        L0 = int(lengths[0])
        seq = agg.view(-1, L0, in_dim)
        zs = zs.view(-1, L0, zs.shape[1])
    hseq = rnn_ref.gru(P, pre + "seq.", seq, num_layers=1, bidirectional=True)   # GRU runs over padding too
    h = _lin(P, pre + "conv_out", torch.cat([zs, hseq], dim=-1))
    h = F.relu(h)                                                       # activation=F.relu at hgnn.py:352-358
    h = _batchnorm_bt(P, pre + "normalize.", h, training)
    if ragged:                                                          # :532-536 drop padded positions
        keep = torch.arange(h.shape[1]).unsqueeze(0) < torch.tensor(sizes).unsqueeze(1)
        h = h[keep]
    else:
        h = h.reshape(-1, h.shape[-1])
    out = _segment_sum(h[edges[1]], edges[0], x.shape[0])               # beats -> notes (sum)   :539
    return out, h


# ------------------------------------------------------------------------------------------
# core/hgnn.py:373-433  in-tree MetricalGNN.forward (metrical=True, use_reledge=False, jk=False)
# ------------------------------------------------------------------------------------------
def metrical_gnn(P: Params, rels: Sequence[str], num_layers: int, x, ei, et, n_beats: int, n_measures: int,
                 beat_edges, measure_edges, beat_lengths=None, measure_lengths=None, training: bool = False):
    n_convs = num_layers                                                # 1 + (num_layers-2) + 1
    h_beat = _segment_sum(_lin(P, "emb_beats", x)[beat_edges[0]], beat_edges[1], n_beats)          # :406
    h_meas = _segment_sum(_lin(P, "emb_measures", x)[measure_edges[0]], measure_edges[1], n_measures)  # :407

    def metrical_block(k, h, h_beat, h_meas):
        bc, h_beat = metrical_conv_layer(P, f"beat_convs.{k}.", h_beat, h, beat_edges, beat_lengths, training)
        mc, h_meas = metrical_conv_layer(P, f"measure_convs.{k}.", h_meas, h, measure_edges, measure_lengths, training)
        h = _lin(P, f"project_metrical.{k}", torch.cat([h, bc, mc], dim=-1))
        return F.normalize(F.relu(h), p=2.0, dim=-1), h_beat, h_meas

    h = x
    for i in range(n_convs - 1):
        if i != 0:
            h, h_beat, h_meas = metrical_block(i - 1, h, h_beat, h_meas)                         # :411-415
        h = hetero_layer(P, f"convs.{i}.", rels, sage_conv_scatter, h, ei, et)                  # :420
        h = F.relu(F.normalize(h, p=2.0, dim=-1))                                               # :421-422
    h, h_beat, h_meas = metrical_block(n_convs - 2, h, h_beat, h_meas)                           # :427-431
    return hetero_layer(P, f"convs.{n_convs - 1}.", rels, sage_conv_scatter, h, ei, et)         # :432


# ------------------------------------------------------------------------------------------
# models/analysis.py:580-587  onset pooling inside TorchAnalysisGNN.encode
#   keep onset edges with both ends < batch_size, drop self loops,
#   x_pool_i = (x_i + sum_{(i,j)} x_j) / max(cnt_i, 1)   [messages from edge row 1 to row 0]
# ------------------------------------------------------------------------------------------
def onset_pool(x: torch.Tensor, onset_edges: torch.Tensor, batch_size: int) -> torch.Tensor:
    e = onset_edges
    e = e[:, (e[0] < batch_size) & (e[1] < batch_size)]
    e = e[:, e[0] != e[1]]
    n = x.shape[0]
    pooled = (x + _segment_sum(x[e[1]], e[0], n)) / _degree(e[0], n, x.dtype).clamp(min=1).unsqueeze(-1)
    return torch.cat([x, pooled], dim=-1)


# ------------------------------------------------------------------------------------------
# core/gnn.py:99-106  RelEdgeConv.forward
#   h = W_n x + b ; e_ij = |h_i - h_j| unless edge features are given ; m_ij = W_e [h_j || e_ij] + b_e
#   s_i = (h_i + sum_{(i,j)} m_ij) / max(deg_i, 1)   [scatter onto edge row 0, out=h.clone(), mean]
#   z = W [x || s] + b
# ------------------------------------------------------------------------------------------
def rel_edge_conv(P: Params, pre: str, x, ei, edge_features=None):
    h = _lin(P, pre + "neigh_linear", x)
    ef = (h[ei[0]] - h[ei[1]]).abs() if edge_features is None else edge_features
    msg = _lin(P, pre + "edge_linear", torch.cat([h[ei[1]], ef], dim=-1))
    n = x.shape[0]
    s = (h + _segment_sum(msg, ei[0], n)) / _degree(ei[0], n, x.dtype).clamp(min=1).unsqueeze(-1)
    return _lin(P, pre + "linear", torch.cat([x, s], dim=-1))


# ------------------------------------------------------------------------------------------
# core/hgnn.py:81-95  HeteroRelEdgeConvLayer.forward: node-level edge features [N, F_e] become |f_i - f_j| per edge
# (:82-83), per-edge ones [E, F_e] are used as they are (:84-85), anything else is dropped (:86-87); mean over relations
# ------------------------------------------------------------------------------------------
def hetero_rel_edge_layer(P: Params, pre: str, rels: Sequence[str], x, ei, et, edge_features=None):
    if edge_features is not None and edge_features.shape[0] == x.shape[0]:
        edge_features = (edge_features[ei[0]] - edge_features[ei[1]]).abs()
    elif edge_features is not None and edge_features.shape[0] == ei.shape[1]:
        pass
    else:
        edge_features = None
    outs = []
    for code, rel in enumerate(rels):
        m = et == code
        outs.append(rel_edge_conv(P, f"{pre}conv.{rel}.", x, ei[:, m], edge_features[m] if edge_features is not None else None))
    return torch.stack(outs, dim=0).mean(dim=0)


# ------------------------------------------------------------------------------------------
# core/hgnn.py:19-23  HeteroAttention.forward on the relation stack x [R, N, H] (the `lstm` reduction, :115, :41):
# the bi-LSTM is batch_first, so it runs over the N NODES as the sequence with the R relations as the batch; the
# softmax is over the last axis of [R, N] (the nodes); the result is the weighted sum over relations.
# ------------------------------------------------------------------------------------------
def hetero_attention(P: Params, pre: str, stack: torch.Tensor) -> torch.Tensor:
    a = rnn_ref.lstm(P, pre + "lstm.", stack, num_layers=1, bidirectional=True)
    a = _lin(P, pre + "att", a).squeeze(-1)
    a = torch.softmax(a, dim=-1)
    return (stack * a.unsqueeze(-1)).sum(dim=0)


def hetero_layer_reduce(P: Params, pre: str, rels: Sequence[str], conv, x, ei, et, reduction: str):
    """hetero_layer with the remaining reductions of core/hgnn.py:102-116 / :30-46 that can run at all: 'lstm'
    (HeteroAttention) and 'none' ([R, N, H]).  'max' / 'min' return a (values, indices) pair whose `.to()` raises in
    the reference and 'concat' calls torch.cat on a tensor (TypeError): nothing to restate for those."""
    outs = [conv(P, f"{pre}conv.{rel}.", x, ei[:, et == code], None) for code, rel in enumerate(rels)]
    stack = torch.stack(outs, dim=0)
    if reduction == "lstm":
        return hetero_attention(P, pre + "reduction.", stack)
    if reduction == "none":
        return stack
    raise NotImplementedError(reduction)


# ------------------------------------------------------------------------------------------
# models/analysis.py:44-101  onsetwise_logit_aggregation (inference post-processing of the softmaxed predictions)
# ------------------------------------------------------------------------------------------
def onsetwise_logit_aggregation(probs: Dict[str, torch.Tensor], onset_edges: torch.Tensor, batch: torch.Tensor,
                                onset_div: torch.Tensor, batch_size: Optional[int] = None, valid_label_mask=None,
                                rna_keys=("quality", "inversion", "degree1", "degree2")) -> Dict[str, torch.Tensor]:
    out = dict(probs)
    if not (rna_keys and all(k in probs for k in rna_keys)):                                    # :45
        return out
    n_all = next(iter(probs.values())).shape[0]
    batch_size = n_all if batch_size is None else batch_size                                   # :46
    valid = torch.ones(batch_size, dtype=torch.bool) if valid_label_mask is None else valid_label_mask   # :48
    e = onset_edges
    e = e[:, (e[0] < batch_size) & (e[1] < batch_size)]                                        # :50-53
    e = e[:, e[0] != e[1]]                                                                      # :55
    tpc = None
    if "tpc_in_label" in probs:                                                                 # :57-59
        tpc = probs["tpc_in_label"].argmax(-1).bool()
        e = e[:, tpc[e[0]] & tpc[e[1]]]
    agg = {}
    for k, v in probs.items():                                                                  # :64-66: scatter_mean(out=v).softmax
        if k in rna_keys:
            n = v.shape[0]
            s = (v + _segment_sum(v[e[0]], e[1], n)) / _degree(e[1], n, v.dtype).clamp(min=1).unsqueeze(-1)
            agg[k] = torch.softmax(s, dim=-1)
    agg = {k: torch.softmax(v[valid], dim=-1) for k, v in agg.items()}                          # :68
    out.update(agg)                                                                             # :69
    bid = batch[:batch_size][valid]                                                             # :70
    if bool(torch.all(bid == bid[0])):                                                          # :71
        onsets = onset_div[:batch_size][valid]
        onsets = onsets - onsets.min()
        if tpc is not None:                                                                     # :74-76
            onsets_f = onsets[tpc]
            agg = {k: v[tpc] for k, v in agg.items()}
        else:
            onsets_f = onsets
        uniq, inv = torch.unique(onsets_f, return_inverse=True)                                 # :79
        first = torch.cat([torch.zeros(1, dtype=torch.long), (inv[1:] != inv[:-1]).nonzero(as_tuple=True)[0] + 1])   # :80-81
        per_onset = {k: v[first] for k, v in agg.items()}                                       # :82
        for k in rna_keys:                                                                      # :84-100
            pred = per_onset[k].argmax(-1)
            cp = torch.cat([torch.zeros(1, dtype=torch.long), (pred[1:] != pred[:-1]).nonzero(as_tuple=True)[0] + 1])
            vals = uniq[cp]
            rows = per_onset[k][cp]
            for i in range(len(cp) - 1):                                                        # the last segment stays as it is
                m = (vals[i] <= onsets) & (onsets < vals[i + 1])
                out[k][m] = rows[i]
    return out
