"""Explicit GRU / LSTM recurrences (CPU, torch tensors).  TEST INFRASTRUCTURE ONLY.

The reference uses `torch.nn.GRU` (core/gnn.py:498; models/cadence.py:249-251) and
`torch.nn.LSTM` (core/gnn.py:352).  torch is present in this image, so these loops are pinned
directly against `torch.nn.GRU/LSTM` in tests/test_oracle_rnn.py; they exist so that the HIP
sequence kernels have a step-by-step oracle whose parameters are plain tensors.

Gate order and equations are torch's documented ones:
  GRU : r,z,n ;  n = tanh(W_in x + b_in + r * (W_hn h + b_hn)) ;  h' = (1-z) n + z h
  LSTM: i,f,g,o ; c' = f c + i g ; h' = o tanh(c')
"""
from __future__ import annotations

from typing import Mapping, Optional

import torch


def _get(P: Mapping[str, torch.Tensor], k: str) -> Optional[torch.Tensor]:
    return P[k] if k in P else None


def gru_direction(x, w_ih, w_hh, b_ih, b_hh, reverse: bool, h0=None):
    """x [B,T,I] -> [B,T,H] for one direction of one layer."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    gi = x @ w_ih.t()
    if b_ih is not None:
        gi = gi + b_ih
    h = x.new_zeros(B, H) if h0 is None else h0
    outs = [None] * T
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        gh = h @ w_hh.t()
        if b_hh is not None:
            gh = gh + b_hh
        i_r, i_z, i_n = gi[:, t].chunk(3, dim=-1)
        h_r, h_z, h_n = gh.chunk(3, dim=-1)
        r = torch.sigmoid(i_r + h_r)
        z = torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + r * h_n)
        h = (1.0 - z) * n + z * h
        outs[t] = h
    return torch.stack(outs, dim=1)


def gru(P: Mapping[str, torch.Tensor], prefix: str, x, num_layers: int = 1, bidirectional: bool = True):
    """Multi-layer (bi)GRU, batch_first, zero initial state, no inter-layer dropout (eval / p=0)."""
    y = x
    for layer in range(num_layers):
        outs = []
        for d, suf in enumerate(["", "_reverse"][: 2 if bidirectional else 1]):
            k = f"l{layer}{suf}"
            outs.append(gru_direction(
                y, P[f"{prefix}weight_ih_{k}"], P[f"{prefix}weight_hh_{k}"],
                _get(P, f"{prefix}bias_ih_{k}"), _get(P, f"{prefix}bias_hh_{k}"), reverse=(d == 1)))
        y = torch.cat(outs, dim=-1)
    return y


def lstm_direction(x, w_ih, w_hh, b_ih, b_hh, reverse: bool):
    B, T, _ = x.shape
    H = w_hh.shape[1]
    gi = x @ w_ih.t()
    if b_ih is not None:
        gi = gi + b_ih
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    outs = [None] * T
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        g = gi[:, t] + h @ w_hh.t()
        if b_hh is not None:
            g = g + b_hh
        i, f, gg, o = g.chunk(4, dim=-1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs[t] = h
    return torch.stack(outs, dim=1)


def lstm(P: Mapping[str, torch.Tensor], prefix: str, x, num_layers: int = 1, bidirectional: bool = True):
    y = x
    for layer in range(num_layers):
        outs = []
        for d, suf in enumerate(["", "_reverse"][: 2 if bidirectional else 1]):
            k = f"l{layer}{suf}"
            outs.append(lstm_direction(
                y, P[f"{prefix}weight_ih_{k}"], P[f"{prefix}weight_hh_{k}"],
                _get(P, f"{prefix}bias_ih_{k}"), _get(P, f"{prefix}bias_hh_{k}"), reverse=(d == 1)))
        y = torch.cat(outs, dim=-1)
    return y


# ------------------------------------------------------------------------------------------
# Fast path for TIMING the CPU baseline only: the same GRU through torch's native CPU kernel
# (what `torch.nn.GRU` — the reference's own call, models/cadence.py:249-251 — executes).
# tests/test_oracle_rnn.py checks it equals the explicit recurrence above.
# ------------------------------------------------------------------------------------------
USE_FAST = False


def gru_fast(P: Mapping[str, torch.Tensor], prefix: str, x, num_layers: int = 1, bidirectional: bool = True):
    flat = []
    for layer in range(num_layers):
        for suf in ["", "_reverse"][: 2 if bidirectional else 1]:
            k = f"l{layer}{suf}"
            flat += [P[f"{prefix}weight_ih_{k}"], P[f"{prefix}weight_hh_{k}"],
                     P[f"{prefix}bias_ih_{k}"], P[f"{prefix}bias_hh_{k}"]]
    H = flat[1].shape[1]
    h0 = x.new_zeros(num_layers * (2 if bidirectional else 1), x.shape[0], H)
    out, _ = torch._VF.gru(x, h0, flat, True, num_layers, 0.0, False, bidirectional, True)
    return out


_gru_loops = gru


def gru(P, prefix, x, num_layers: int = 1, bidirectional: bool = True):  # noqa: F811
    if USE_FAST and (prefix + "bias_ih_l0") in P:
        return gru_fast(P, prefix, x, num_layers, bidirectional)
    return _gru_loops(P, prefix, x, num_layers, bidirectional)
