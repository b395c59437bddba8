"""`TorchAnalysisGNN`: the caller-side boundary of the hot path, mirroring
analysisgnn/models/analysis.py:421-591 (same constructor arguments, parameter names and
`encode` / `forward_clf` / `forward` signatures) with the encoder and the onset pooling running
on the HIP kernels.  Lightning training policy (losses, continual learning, schedulers) is out
of scope."""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops
from .encoders import HybridGNN, MetricalGNN
from .fused import FusedSequential, advance_rng
from .embedding import embed_cat
from .heads import CrossTaskTransformer, fused_head_logits, fused_logit_fusion
from .linear import Linear
from .graph import SegSpec, build_csr


def onset_pool(x: torch.Tensor, onset_edges: torch.Tensor, batch_size: int, index=None) -> torch.Tensor:
    """analysis.py:580-587: onset edges with both ends < batch_size, self loops removed,
    x_pool_i = (x_i + sum_j x_j) / max(cnt_i, 1), returned as cat([x, x_pool]).
    The two boolean-mask compactions become kernel-side predicates (col < limit, col != row).  Messages flow from
    edge row 1 to edge row 0, i.e. along the onset relation's by-SOURCE CSR — when the batch index is at hand
    (`index`), that CSR and its transpose are reused instead of built again."""
    _lib.require_gpu(x, onset_edges)
    n = int(x.shape[0])
    et = ("note", "onset", "note")
    if index is not None and et in index.bwd and index.bwd[et].n_rows >= n:
        fwd, bwd = index.bwd[et], index.fwd[et]
    else:
        fwd, bwd = build_csr([SegSpec(onset_edges[0], onset_edges[1], n_rows=max(n, batch_size)),
                              SegSpec(onset_edges[1], onset_edges[0], n_rows=max(n, batch_size))])
    spec = ops.AggSpec(fwd=[fwd], bwd=[bwd], src_id=[0], n_rows=min(n, batch_size), mean=True, shared_slot=True,
                       skip_self=True, col_limit=batch_size)
    pooled = ops.aggregate(spec, [x], self_t=x)
    if pooled.shape[0] < n:                     # rows beyond batch_size keep their own value
        pooled = torch.cat([pooled, x[pooled.shape[0]:]], dim=0)
    return torch.cat([x, pooled], dim=-1)


def _input_mlp(i, h, dropout):
    return FusedSequential(Linear(i, h), nn.ReLU(), nn.LayerNorm(h), nn.Dropout(dropout), Linear(h, h))


class TorchAnalysisGNN(nn.Module):
    def __init__(self, metadata, in_channels, hidden_channels, out_channels, task_dict, num_layers, dropout=0.5,
                 use_jk=True, logit_fusion=True, use_rnn=False, encoder_type="hybridgnn"):
        """Same defaults as the reference constructor (models/analysis.py:422: `logit_fusion=True`; the training CLI
        passes False unless --logit_fusion is given, train/train_analysisgnn.py:97)."""
        super().__init__()
        if use_rnn:
            # the reference's own use_rnn branch cannot run (rnn_mlp = Linear(o, o/2) -> LayerNorm(o),
            # GRU output 2*o wide into LayerNorm(o), models/analysis.py:512-521): there is no behaviour to reproduce
            raise NotImplementedError("use_rnn=True: the reference branch is shape-inconsistent (models/analysis.py:512-521)")
        self.pitch_embedding = nn.Embedding(35, 64)
        self.key_embedding = nn.Embedding(15, 64)
        self.logit_fusion = logit_fusion
        self.use_rnn = use_rnn
        self.hidden_channels = hidden_channels
        self.project_dict = nn.ModuleDict({
            k: _input_mlp(in_channels + (128 if k == "note" else 0), hidden_channels, dropout) for k in metadata[0]})
        if encoder_type == "hgt":
            from .hgt import HybridHGT
            self.encoder = HybridHGT(metadata=metadata, input_channels=hidden_channels, hidden_channels=hidden_channels,
                                     num_layers=num_layers, heads=4, dropout=dropout, use_jk=use_jk)
        elif encoder_type == "hybridgnn":
            self.encoder = HybridGNN(metadata=metadata, input_channels=hidden_channels, hidden_channels=hidden_channels,
                                     num_layers=num_layers, dropout=dropout, use_jk=use_jk)
        elif encoder_type == "metricalgnn":
            self.encoder = MetricalGNN(metadata=metadata, input_channels=hidden_channels, hidden_channels=hidden_channels,
                                       output_channels=hidden_channels, num_layers=num_layers, dropout=dropout,
                                       use_jk=use_jk, fast=True)
        else:
            raise ValueError(encoder_type)
        h, o = hidden_channels, out_channels
        self.project_enc = FusedSequential(
            nn.LayerNorm(2 * h), Linear(2 * h, h), nn.ReLU(), nn.LayerNorm(h), nn.Dropout(dropout),
            Linear(h, o), nn.ReLU(), nn.LayerNorm(o), nn.Dropout(dropout), Linear(o, o))
        self.clf_dict = nn.ModuleDict({
            t: nn.Sequential(nn.Linear(o, o // 2), nn.ReLU(), nn.LayerNorm(o // 2), nn.Linear(o // 2, c))
            for t, c in task_dict.items()})
        if logit_fusion:                                        # models/analysis.py:497-511
            self.clf_proj_layers = nn.ModuleDict({
                t: nn.Sequential(nn.Linear(c, o // 2), nn.ReLU(), nn.LayerNorm(o // 2)) for t, c in task_dict.items()})
            self.cross_task_transformer = CrossTaskTransformer(o // 2, num_heads=4, dropout=dropout)
            self.fusion_layers = nn.ModuleDict({t: nn.Linear(o // 2, c) for t, c in task_dict.items()})
        self.rnn, self.rnn_norm, self.rnn_mlp = nn.Identity(), nn.Identity(), nn.Identity()   # use_rnn=False, :522-525

    def adjacent_parameter_groups(self):
        """Parameters the fused schedule consumes concatenated (heads.fused_head_logits: per-layer cats over the tasks;
        gru._stack_gru_params: direction pairs): dp.plan_parameters lays them out back to back so that the cats are views."""
        mods = list(self.clf_dict.values())
        groups = [[m[0].weight for m in mods], [m[0].bias for m in mods], [m[2].weight for m in mods],
                  [m[2].bias for m in mods], [m[3].weight for m in mods], [m[3].bias for m in mods]]
        rnn = getattr(self.encoder, "rnn", None)
        if isinstance(rnn, nn.GRU) and rnn.bidirectional:
            for layer in range(rnn.num_layers):
                for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                    groups.append([getattr(rnn, f"{n}_l{layer}"), getattr(rnn, f"{n}_l{layer}_reverse")])
        return groups

    # `torch.compile(model, dynamic=True)` (reference train/train_analysisgnn.py:202-203): the hot path is hand-written
    # kernels behind ctypes calls inside autograd.Functions — nothing for a tracing compiler to fuse, and nothing it can trace
    # through.  The three entry points are excluded from tracing as a whole, so a compiled wrapper runs this code unchanged
    # (one clean graph break at the boundary instead of one per kernel call).
    # Data-parallel steps (dp.FlatGradBuffer(late=model.late_parameters())): with `split_backward = True` the autograd graph is
    # cut behind the input layers, `loss.backward()` ends there — every gradient but theirs exists and can be all-reduced —
    # and `finish_backward()` runs the input layers' backward beside the collective.  Same kernels, same order; the gradients
    # equal the unsplit step's to fp32 rounding (deferred weight-gradient products are batched per flush: the groups differ).
    split_backward = False

    def late_parameters(self):
        """The parameters whose gradients the backward pass produces last: the input MLPs and the two embedding tables."""
        return [*self.project_dict.parameters(), *self.pitch_embedding.parameters(), *self.key_embedding.parameters()]

    def finish_backward(self) -> None:
        cut, self._cut = getattr(self, "_cut", None), None
        if cut:
            live = [(h, leaf.grad) for h, leaf in cut if leaf.grad is not None]
            if live:
                torch.autograd.backward([h for h, _ in live], [g for _, g in live])

    @torch._dynamo.disable
    def encode(self, pitch_spelling, key_signature, x_dict, edge_index_dict, batch_dict, batch_size,
               neighbor_mask_node, neighbor_mask_edge):
        if self.training:
            advance_rng(x_dict["note"].device)          # fresh dropout masks for this step (device-side counter)
        z_dict = dict(x_dict)
        # analysis.py:574 — one launch; rows padded to 16 bytes (the spare columns are zero and stay out of the view)
        z_dict["note"] = embed_cat(z_dict["note"], [pitch_spelling, key_signature],
                                   [self.pitch_embedding.weight, self.key_embedding.weight])
        for k, z in z_dict.items():
            # an odd input width with unpadded rows (the 25 beat / measure features): one spare zero column behind the last one, out of
            # the view — the first projection's weight gradient then runs on the MFMA kernel and can be deferred with the others
            # (as a library product on the step's tail it took 38 + 70 us for 3 688 rows: dW with K = N, and a column sum)
            if k != "note" and z.is_cuda and z.dim() == 2 and (z.shape[1] & 1) and z.stride(1) == 1 and z.stride(0) == z.shape[1] and z.shape[0] > 0:
                z_dict[k] = F.pad(z, (0, 1))[:, :z.shape[1]]
        h_dict = {k: self.project_dict[k](z_dict[k]) for k in self.project_dict.keys()}
        self._cut = None
        if self.split_backward and torch.is_grad_enabled():
            self._cut = []
            for k, hk in list(h_dict.items()):
                if hk.requires_grad:
                    leaf = hk.detach().requires_grad_(True)
                    self._cut.append((hk, leaf))
                    h_dict[k] = leaf
        x = self.encoder(x_dict=h_dict, edge_index_dict=edge_index_dict, batch_dict=batch_dict,
                         batch_size=batch_size, neighbor_mask_node=neighbor_mask_node,
                         neighbor_mask_edge=neighbor_mask_edge, return_edge_index=False, edge_attr_dict=None)
        index = getattr(getattr(self.encoder, "gnn", None), "last_index", None)    # the CSR the encoder just built
        x = onset_pool(x, edge_index_dict[("note", "onset", "note")], batch_size, index)
        return self.project_enc(x)

    @torch._dynamo.disable
    def forward_clf(self, x, tasks=None):
        """Same dict of logits as analysis.py:546-548, computed by the fused head schedule (heads.py);
        the values are column views of one [N, sum C] matrix."""
        logits, offs, tasks = self.forward_clf_fused(x, tasks)
        return {t: logits[:, offs[i]:offs[i + 1]] for i, t in enumerate(tasks)}

    @torch._dynamo.disable
    def forward_clf_fused(self, x, tasks=None):
        tasks = list(self.clf_dict.keys() if tasks is None else tasks)
        logits, offs = fused_head_logits(self.clf_dict, x, tasks)
        if self.logit_fusion:                                   # models/analysis.py:550-565: refined logits replace the raw ones
            logits = fused_logit_fusion(self.clf_proj_layers, self.cross_task_transformer, self.fusion_layers, logits, offs,
                                        tasks, self.training)
        return logits, offs, tasks

    def forward(self, pitch_spelling, key_signature, x_dict, edge_index_dict, batch_dict, batch_size,
                neighbor_mask_node, neighbor_mask_edge):
        return self.forward_clf(self.encode(pitch_spelling, key_signature, x_dict, edge_index_dict, batch_dict,
                                            batch_size, neighbor_mask_node, neighbor_mask_edge))

    def clf_task(self, x, task_name):
        return self.clf_dict[task_name](x)
