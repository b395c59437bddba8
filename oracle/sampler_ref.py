"""CPU restatement of the device-side batch sampler (include/agnn.h `agnn_sample_hops`).  TEST INFRASTRUCTURE ONLY.

What the reference does at this point: graphmuse's `MuseNeighborLoader` (data/datamodules/analysis.py:270-293 —
third-party, not in the tree, not installable here) samples `batch_size` windows of `subgraph_size` target notes and up
to `num_neighbors[h]` in-neighbours per relation and hop, PyG NeighborLoader layout (hop-ordered nodes and edges, per-hop
counts).  Its random stream and its tie-breaking are not observable offline, so there is nothing to pin the SAMPLE
against: **parity unpinned** for the choice of neighbours.  What this file pins bit for bit is the build's own contract
(agnn.h): the same windows, seed and step give the same padded batch — integers only, compared exactly.
The invariants the reference's consumers rely on (hop order, every kept edge's target in an earlier hop block, a node
expanded only in the hop after it appeared, at most `fan` in-neighbours per relation and frontier node, no edge across
subgraphs) are checked as properties in tests/test_sampler.py."""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

M32 = 0xFFFFFFFF


def philox4x32_10(c: Sequence[int], k: Sequence[int]) -> Tuple[int, int, int, int]:
    """Philox-4x32-10 (Salmon et al. 2011), the same round function as csrc/sampler.hip / normact.hip."""
    c0, c1, c2, c3 = (int(v) & M32 for v in c)
    k0, k1 = (int(v) & M32 for v in k)
    for _ in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        h0, l0 = p0 >> 32, p0 & M32
        h1, l1 = p1 >> 32, p1 & M32
        c0, c1, c2, c3 = (h1 ^ c1 ^ k0) & M32, l1, (h0 ^ c3 ^ k1) & M32, l0
        k0 = (k0 + 0x9E3779B9) & M32
        k1 = (k1 + 0xBB67AE85) & M32
    return c0, c1, c2, c3


def select_positions(deg: int, fan: int, dst: int, tag: int, step: int, key: Tuple[int, int]) -> List[int]:
    """All positions when deg <= fan, otherwise `fan` of them by selection sampling (Knuth's algorithm S)."""
    if deg <= fan:
        return list(range(deg))
    out: List[int] = []
    rnd = (0, 0, 0, 0)
    for t in range(deg):
        if len(out) >= fan:
            break
        if t & 3 == 0:
            rnd = philox4x32_10((dst, tag, t >> 2, step), key)
        u = rnd[t & 3]
        if ((u * (deg - t)) >> 32) < (fan - len(out)):
            out.append(t)
    return out


def sample_hops(rowptr: Sequence[np.ndarray], col: Sequence[np.ndarray], win_start: Sequence[int], n_targets: int,
                fan: Sequence[int], cap: Sequence[int], seed: int, step: int):
    """-> (node_gid int32 [N_batch], edges: list per relation of int64 [2, e_cap], dropped)."""
    R, B, T, hops = len(rowptr), len(win_start), int(n_targets), len(fan)
    key = (seed & M32, (seed >> 32) & M32)
    n_nodes = B * T + sum(B * c for c in cap)
    e_cap, F = 0, T
    for h in range(hops):
        e_cap += B * F * fan[h]
        F = cap[h]
    node_gid = np.full(n_nodes, -1, dtype=np.int32)
    edges = [np.full((2, e_cap), -1, dtype=np.int64) for _ in range(R)]
    dropped = 0
    for s in range(B):
        w = int(win_start[s])
        node_gid[s * T:(s + 1) * T] = np.arange(w, w + T)
        local: Dict[int, int] = {}
        known = set()
        nbase, ebase = B * T, 0
        frontier = [w + i for i in range(T)]
        Fcap, fr_local0 = T, s * T
        for h in range(hops):
            picks = {}
            new = set()
            for i, dst in enumerate(frontier):
                for r in range(R):
                    st, deg = int(rowptr[r][dst]), int(rowptr[r][dst + 1] - rowptr[r][dst])
                    pos = select_positions(deg, fan[h], dst, r + (h << 8), step & M32, key)
                    srcs = [int(col[r][st + p]) for p in pos]
                    picks[(i, r)] = srcs
                    for g in srcs:
                        if not (w <= g < w + T) and g not in known:
                            new.add(g)
            known |= new
            order = sorted(new)
            kept = order[:cap[h]]
            dropped += len(order) - len(kept)
            for rank, g in enumerate(kept):
                local[g] = nbase + s * cap[h] + rank
                node_gid[nbase + s * cap[h] + rank] = g
            for i in range(len(frontier)):
                for r in range(R):
                    for k, g in enumerate(picks[(i, r)]):
                        lid = s * T + (g - w) if w <= g < w + T else local.get(g, -1)
                        if lid >= 0:
                            slot = ebase + (s * Fcap + i) * fan[h] + k
                            edges[r][0, slot] = lid
                            edges[r][1, slot] = fr_local0 + i
            ebase += B * Fcap * fan[h]
            fr_local0 = nbase + s * cap[h]
            nbase += B * cap[h]
            frontier, Fcap = kept, cap[h]
    return node_gid, edges, dropped


def sample_members(node_gid: np.ndarray, group_of: np.ndarray, win_start: Sequence[int], n_targets: int, cap: Sequence[int], cap_g: int):
    """CPU restatement of `agnn_sample_members` (include/agnn.h): the metrical nodes (beats / measures) of a sampled batch and
    the (note, connects, group) membership edges of its notes.  -> (group_gid int32 [B * cap_g], edges int64 [2, n_nodes], dropped).
    The build's own contract (graphmuse's loader is absent: the CHOICE of nodes is parity unpinned, the layout invariants —
    contiguous ranges, an edge only between a note and a group of its own subgraph, edge slot = note slot — are properties)."""
    B, T = len(win_start), int(n_targets)
    n_nodes = int(node_gid.shape[0])
    assert n_nodes == B * T + sum(B * c for c in cap)
    group_gid = np.full(B * cap_g, -1, dtype=np.int32)
    edges = np.full((2, n_nodes), -1, dtype=np.int64)
    dropped = 0
    rng = []
    for s in range(B):
        w = int(win_start[s])
        gmin, gmax = int(group_of[w]), int(group_of[w + T - 1])
        rng.append((gmin, gmax))
        for j in range(cap_g):
            if gmin + j <= gmax:
                group_gid[s * cap_g + j] = gmin + j
        dropped += max(0, gmax - gmin + 1 - cap_g)
    blocks = [(0, T)]
    base = B * T
    for c in cap:
        blocks.append((base, c))
        base += B * c
    for base, width in blocks:
        for s in range(B):
            gmin, gmax = rng[s]
            for i in range(width):
                slot = base + s * width + i
                g = int(node_gid[slot])
                if g < 0:
                    continue
                gg = int(group_of[g])
                if gmin <= gg <= gmax and gg - gmin < cap_g:
                    edges[0, slot] = slot
                    edges[1, slot] = s * cap_g + (gg - gmin)
    return group_gid, edges, dropped


def compact(node_gid: np.ndarray, edges: Sequence[np.ndarray], n_sub: int, n_targets: int, cap: Sequence[int], pool: Sequence[int]):
    """CPU restatement of `agnn_sample_compact`: the padded hop blocks squeezed into batch-wide pools.
    -> (node_gid_pool int32, edges (remapped copies), batch int64 [n_out], dropped)."""
    B, T = int(n_sub), int(n_targets)
    n_tgt = B * T
    n_out = n_tgt + sum(pool)
    gid = np.full(n_out, -1, dtype=np.int32)
    batch = np.full(n_out, B - 1, dtype=np.int64)
    gid[:n_tgt] = node_gid[:n_tgt]
    batch[:n_tgt] = np.arange(n_tgt) // T
    new_id = np.full(node_gid.shape[0] + 1, -1, dtype=np.int64)       # last entry: the image of -1
    new_id[:n_tgt] = np.arange(n_tgt)
    base, pbase, dropped = n_tgt, n_tgt, 0
    for h, (c, p) in enumerate(zip(cap, pool)):
        run = 0
        for s in range(B):
            blk = node_gid[base + s * c:base + (s + 1) * c]
            k = int((blk >= 0).sum())
            assert (blk[:k] >= 0).all()                               # kept nodes fill a block from the front
            take = max(0, min(k, p - run))
            gid[pbase + run:pbase + run + take] = blk[:take]
            batch[pbase + run:pbase + run + take] = s
            new_id[base + s * c:base + s * c + take] = pbase + run + np.arange(take)
            run += take
            dropped += k - take
        base += B * c
        pbase += p
    out = []
    for e in edges:
        s1, d1 = new_id[e[0]], new_id[e[1]]                           # index -1 -> the sentinel
        ok = (s1 >= 0) & (d1 >= 0)
        out.append(np.stack([np.where(ok, s1, -1), np.where(ok, d1, -1)]))
    return gid, out, batch, dropped
