"""Post-processing and wire formats either side of the hot path, so that the reference's
predict pipeline can switch imports:

  * `onsetwise_logit_aggregation` — analysisgnn/models/analysis.py:44-101: the softmaxed Roman-numeral predictions are
    averaged over the notes of an onset (`torch_scatter.scatter_mean(..., out=v)`, :66) and smoothed between the onsets
    where the predicted class changes.  The scatter runs on the gather-reduce kernel (all four RNA keys in ONE launch);
    the per-change-point Python loop of the reference (:96-100) is one vectorised assignment.
  * `predict` — what `ContinualAnalysisGNN.predict` does after graph construction (models/analysis.py:1533-1588):
    eval-mode forward with `neighbor_mask_* = None`, softmax, onset-wise aggregation.
  * `checkpoint_state_dict` / `load_reference_checkpoint` — a Lightning `.ckpt` of the reference
    (inference/predict_analysis.py:150-159: `ContinualAnalysisGNN.load_from_checkpoint`) -> this build's `state_dict`.
    Read with `torch.load(weights_only=True)` only.

Decoding class indices to label strings (`available_representations`, inference/predict_analysis.py:178-200) stays
with the reference's vocabulary tables: label vocabularies are outside the hot-path scope."""
from __future__ import annotations

from typing import Dict, Iterable, List, Mapping, Optional, Sequence, Tuple

import torch

from . import _lib, ops
from .graph import SegSpec, build_csr

RNA_KEYS = ("quality", "inversion", "degree1", "degree2")


def _field(graph, name):
    note = graph["note"]
    return note[name] if isinstance(note, Mapping) else getattr(note, name)


def onsetwise_logit_aggregation(logits_softmax_dict: Dict[str, torch.Tensor], graph=None, edge_index_dict=None, batch_size=None,
                                valid_label_mask=None, rna_keys: Sequence[str] = RNA_KEYS, batch: Optional[torch.Tensor] = None,
                                onset_div: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """Same arguments and result as the reference function (models/analysis.py:44).  `graph` is whatever the caller has
    (a PyG HeteroData or any mapping with graph["note"].x / .batch / .onset_div and `.edge_index_dict`); without one,
    pass `edge_index_dict`, `batch` and `onset_div` directly.  The input dict is updated and returned, as in the
    reference; unlike there, the input tensors themselves are not modified in place (`out=v`, :66)."""
    d = logits_softmax_dict
    rna_keys = list(rna_keys)
    if not (rna_keys and all(k in d for k in rna_keys)):                                        # :45
        return d
    first = d[rna_keys[0]]
    dev = _lib.require_gpu(first)
    if batch_size is None:                                                                      # :46
        batch_size = int(_field(graph, "x").shape[0]) if graph is not None else int(first.shape[0])
    if edge_index_dict is None:
        edge_index_dict = graph.edge_index_dict
    if valid_label_mask is None:                                                                # :48
        valid_label_mask = torch.ones(batch_size, dtype=torch.bool, device=dev)
    if batch is None:
        batch = _field(graph, "batch") if graph is not None else torch.zeros(batch_size, dtype=torch.long, device=dev)
    if onset_div is None and graph is not None:
        onset_div = _field(graph, "onset_div")
    onset_edges = edge_index_dict["note", "onset", "note"]
    n = int(first.shape[0])
    # --- :50-66  mean over the onset neighbours (both ends < batch_size, no self loops, optional tpc mask), numerator
    #     includes the note itself (`out=v`), all RNA keys side by side in one launch
    e0, e1 = onset_edges[0], onset_edges[1]
    tpc = None
    if "tpc_in_label" in d:                                                                     # :57-59
        tpc = d["tpc_in_label"].argmax(-1).bool()
        # an edge survives when both ends carry the flag: fold the node flag into the row / column ids (dropped edges
        # get row = n, which the CSR build discards as out of range)
        ok = tpc[e0.clamp(max=n - 1)] & tpc[e1.clamp(max=n - 1)] & (e0 < n) & (e1 < n)
        e1 = torch.where(ok, e1, torch.full_like(e1, n))
    widths = [int(d[k].shape[1]) for k in rna_keys]
    W = sum(widths)
    Wp = (W + 3) & ~3
    v = torch.zeros((n, Wp), dtype=torch.float32, device=dev)
    o = 0
    for k, w in zip(rna_keys, widths):
        v[:, o:o + w] = d[k]
        o += w
    fwd, bwd = build_csr([SegSpec(e1, e0, n), SegSpec(e0, e1, n)])                              # rows = edge row 1 (the scatter index)
    lim = min(batch_size, n)
    spec = ops.AggSpec(fwd=[fwd], bwd=[bwd], src_id=[0], n_rows=lim, mean=True, shared_slot=True, skip_self=True, col_limit=lim)
    with torch.no_grad():
        s = ops.aggregate(spec, [v], self_t=v)
    if lim < n:
        s = torch.cat([s, v[lim:]], dim=0)
    agg = {}
    o = 0
    for k, w in zip(rna_keys, widths):
        agg[k] = torch.softmax(torch.softmax(s[:, o:o + w], dim=-1)[valid_label_mask], dim=-1)   # :66 and :68: softmax twice
        o += w
    d.update(agg)                                                                               # :69
    # --- :70-100  hold the prediction between the onsets where the predicted class changes (single score only)
    bid = batch[:batch_size][valid_label_mask]
    if onset_div is not None and bid.numel() and bool(torch.all(bid == bid[0])):                # :71 (host sync, as in the reference)
        onsets = onset_div[:batch_size][valid_label_mask]
        onsets = onsets - onsets.min()
        if tpc is not None:                                                                     # :74-76
            onsets_f = onsets[tpc]
            aggf = {k: t[tpc] for k, t in agg.items()}
        else:
            onsets_f, aggf = onsets, agg
        uniq, inv = torch.unique(onsets_f, return_inverse=True)                                 # :79
        zero = torch.zeros(1, dtype=torch.long, device=dev)
        firsts = torch.cat([zero, (inv[1:] != inv[:-1]).nonzero(as_tuple=True)[0] + 1])         # :80-81
        for k in rna_keys:                                                                      # :84-100
            per_onset = aggf[k][firsts]
            pred = per_onset.argmax(-1)
            cp = torch.cat([zero, (pred[1:] != pred[:-1]).nonzero(as_tuple=True)[0] + 1])
            vals = uniq[cp]
            rows = per_onset[cp]
            # note -> segment [vals[i], vals[i+1]); the last segment keeps its own values (the loop stops at len - 1)
            seg = torch.searchsorted(vals, onsets, right=True) - 1
            m = (seg >= 0) & (seg < cp.numel() - 1)
            d[k][m] = rows[seg[m]]
    return d


@torch.no_grad()
def predict(model, pitch_spelling, key_signature, x_dict, edge_index_dict, batch_dict=None, onset_div=None,
            tasks: Optional[Iterable[str]] = None) -> Dict[str, torch.Tensor]:
    """Whole-score inference as models/analysis.py:1560-1588: no sampling masks, every note is a target; returns the
    softmaxed predictions after the onset-wise aggregation.  `model` is a `TorchAnalysisGNN` of this package."""
    was = model.training
    model.eval()
    try:
        n = int(x_dict["note"].shape[0])
        if batch_dict is None:
            batch_dict = {k: torch.zeros(v.shape[0], dtype=torch.long, device=v.device) for k, v in x_dict.items()}
        logits = model(pitch_spelling, key_signature, x_dict, edge_index_dict, batch_dict, n, None, None)
        if tasks is not None:
            logits = {k: v for k, v in logits.items() if k in set(tasks)}
        probs = {k: torch.softmax(v, dim=-1) for k, v in logits.items()}
        return onsetwise_logit_aggregation(probs, edge_index_dict=edge_index_dict, batch_size=n, batch=batch_dict["note"],
                                           onset_div=onset_div)
    finally:
        model.train(was)


# ------------------------------------------------------------------------------------------------------------------
# Lightning checkpoint -> state_dict
# ------------------------------------------------------------------------------------------------------------------
def checkpoint_state_dict(ckpt: Mapping, prefix: str = "model.") -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
    """(model_state, rest) from a loaded Lightning checkpoint of the reference's `ContinualAnalysisGNN` (or from a bare
    state_dict).  The Lightning module keeps the network under `self.model` (models/analysis.py:853-862), so its
    parameters are stored as `model.<name>`; `<name>` is exactly this build's `TorchAnalysisGNN.state_dict()` key for
    everything the reference defines in-tree (`pitch_embedding`, `key_embedding`, `project_dict.*`, `project_enc.*`,
    `clf_dict.*`, `clf_proj_layers.*`, `cross_task_transformer.*`, `fusion_layers.*`).  `encoder.*` holds graphmuse's
    parameters: they map by name where graphmuse uses the PyG names this build uses (encoders.py header) and are reported
    by `load_reference_checkpoint` otherwise.  `rest` holds what is not the network (`clf_loss.params` = the MultiTaskLoss
    weights, `memory_model.*` = the continual-learning teacher copy, ...)."""
    sd = ckpt["state_dict"] if "state_dict" in ckpt and isinstance(ckpt["state_dict"], Mapping) else ckpt
    model, rest = {}, {}
    for k, v in sd.items():
        if not isinstance(v, torch.Tensor):
            continue
        if k.startswith(prefix):
            model[k[len(prefix):]] = v
        else:
            rest[k] = v
    if not model:                       # a bare TorchAnalysisGNN state_dict
        model, rest = {k: v for k, v in sd.items() if isinstance(v, torch.Tensor)}, {}
    return model, rest


def load_reference_checkpoint(model: torch.nn.Module, path_or_ckpt, strict: bool = False, clf_loss: Optional[torch.nn.Module] = None,
                              allow_encoder_mismatch: bool = False):
    """Load a reference checkpoint into `model` (a `TorchAnalysisGNN` of this package).  Files are read with
    `torch.load(..., weights_only=True)` — nothing from the file is executed; a checkpoint the safe loader refuses
    (Lightning may pickle arbitrary hyper-parameter objects) raises and must be re-saved as a plain state_dict by the
    reference.  Returns (missing_keys, unexpected_keys, hyper_parameters) like `load_state_dict`; `strict=True` raises when
    either list is non-empty.  `clf_loss` (heads.MultiTaskLoss) receives `clf_loss.params` when given.
    The encoder's parameter names follow PyG / the build spec of DESIGN §5 — graphmuse's own names cannot be checked offline
    (parity unpinned) — so ANY missing or unexpected `encoder.*` key raises even with `strict=False`: a silently
    half-initialised encoder would predict garbage.  `allow_encoder_mismatch=True` opts out (the lists are still returned)."""
    if isinstance(path_or_ckpt, (str, bytes)) or hasattr(path_or_ckpt, "__fspath__"):
        ckpt = torch.load(path_or_ckpt, map_location="cpu", weights_only=True)
    else:
        ckpt = path_or_ckpt
    sd, rest = checkpoint_state_dict(ckpt)
    own = model.state_dict()
    missing = [k for k in own if k not in sd]
    unexpected = [k for k in sd if k not in own]
    bad_shape = [k for k in sd if k in own and tuple(sd[k].shape) != tuple(own[k].shape)]
    if bad_shape:
        raise _lib.AgnnError(f"checkpoint tensors with another shape than the model's: {bad_shape[:5]} ...")
    enc = [k for k in (*missing, *unexpected) if k.startswith("encoder.")]
    if enc and not allow_encoder_mismatch:
        raise _lib.AgnnError(f"checkpoint and model disagree on {len(enc)} encoder tensors (e.g. {enc[:4]}): the encoder would "
                             "keep its random initialisation there; map the names first, or pass allow_encoder_mismatch=True")
    if strict and (missing or unexpected):
        raise _lib.AgnnError(f"checkpoint does not match: missing {missing[:5]} ..., unexpected {unexpected[:5]} ...")
    model.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
    if clf_loss is not None and "clf_loss.params" in rest:
        with torch.no_grad():
            clf_loss.params.copy_(rest["clf_loss.params"].to(clf_loss.params.device))
    hp = ckpt.get("hyper_parameters") if isinstance(ckpt, Mapping) else None
    return missing, unexpected, hp
