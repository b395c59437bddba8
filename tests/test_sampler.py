"""The batch sampler's contract on the CPU restatement (oracle/sampler_ref.py): the invariants the reference's consumers
rely on (PyG NeighborLoader layout consumed at models/analysis.py:948-961 and by trim_to_layer, models/cadence.py:165-173).
The choice of neighbours itself is parity-unpinned (graphmuse's loader is not available); these are properties."""
import numpy as np
import pytest

from analysisgnn_amd.synth import make_score_graph
from oracle import sampler_ref as S


def _store(n_scores=3, n_notes=700):
    graphs = [make_score_graph(seed=40 + i, n_notes=n_notes) for i in range(n_scores)]
    start = np.concatenate([[0], np.cumsum([n_notes] * n_scores)])
    ets = graphs[0].edge_types
    rowptr, col = [], []
    for et in ets:
        src = np.concatenate([g.edge_index[et][0] + o for g, o in zip(graphs, start[:-1])])
        dst = np.concatenate([g.edge_index[et][1] + o for g, o in zip(graphs, start[:-1])])
        order = np.lexsort((np.arange(dst.size), dst))                # stable by destination
        rp = np.zeros(start[-1] + 1, dtype=np.int64)
        np.add.at(rp, dst + 1, 1)
        rowptr.append(np.cumsum(rp).astype(np.int32))
        col.append(src[order].astype(np.int32))
    return graphs, start, rowptr, col


def test_philox_known_answer():
    """Random123 known-answer vectors for Philox-4x32-10 (kat_vectors: counter / key all zeros, all ones, and the pi digits)."""
    assert S.philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert S.philox4x32_10((0xffffffff,) * 4, (0xffffffff,) * 2) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert S.philox4x32_10((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
        (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)


@pytest.mark.parametrize("fan,cap", [((5, 5), (64, 64)), ((2, 3), (96, 128)), ((1,), (8,))])
def test_layout_invariants(fan, cap):
    graphs, start, rowptr, col = _store()
    T, wins = 120, [50, 700 + 300, 1400 + 580]
    gid, edges, dropped = S.sample_hops(rowptr, col, wins, T, fan, cap, seed=7, step=3)
    B = len(wins)
    blocks = [B * T] + [B * c for c in cap]
    nb = np.concatenate([[0], np.cumsum(blocks)])
    assert gid.size == nb[-1]
    # targets are the windows; every node belongs to its window's score; no node twice inside a subgraph
    for s, w in enumerate(wins):
        assert (gid[s * T:(s + 1) * T] == np.arange(w, w + T)).all()
        mine = [gid[s * T:(s + 1) * T]] + [gid[nb[h + 1] + s * cap[h]: nb[h + 1] + (s + 1) * cap[h]] for h in range(len(cap))]
        allg = np.concatenate(mine)
        real = allg[allg >= 0]
        assert real.size == np.unique(real).size
        sc = np.searchsorted(start, w, side="right") - 1
        assert ((real >= start[sc]) & (real < start[sc + 1])).all()
        for h in range(len(cap)):                      # a hop's new nodes: ascending global id, padding behind them
            blk = mine[h + 1]
            k = int((blk >= 0).sum())
            assert (blk[:k] >= 0).all() and (blk[k:] == -1).all() and (np.diff(blk[:k]) > 0).all()
    F, e0 = T, 0
    for h in range(len(cap)):
        eh = B * F * fan[h]
        for r in range(len(edges)):
            seg = edges[r][:, e0:e0 + eh]
            live = seg[0] >= 0
            assert ((seg[0] >= 0) == (seg[1] >= 0)).all()
            # hop h+1 edges: target in hop block h, source in hop blocks <= h + 1 (what trim_to_layer assumes)
            assert ((seg[1][live] >= nb[h]) & (seg[1][live] < nb[h + 1])).all()
            assert (seg[0][live] < nb[h + 2]).all()
            assert (gid[seg[0][live]] >= 0).all() and (gid[seg[1][live]] >= 0).all()
            # the edge exists in the score graph, at most `fan` per (target, relation), none across subgraphs
            et = graphs[0].edge_types[r]
            for s_l, d_l in zip(seg[0][live][:200], seg[1][live][:200]):
                gs, gd = int(gid[s_l]), int(gid[d_l])
                st, en = rowptr[r][gd], rowptr[r][gd + 1]
                assert gs in col[r][st:en]
            cnt = np.bincount(seg[1][live], minlength=nb[-1])
            assert cnt.max(initial=0) <= fan[h]
        e0 += eh
        F = cap[h]
    assert e0 == edges[0].shape[1]


def test_reproducible_and_step_dependent():
    _, _, rowptr, col = _store()
    a = S.sample_hops(rowptr, col, [10, 900], 100, (1, 1), (64, 64), seed=5, step=1)      # fan 1 forces the random branch
    b = S.sample_hops(rowptr, col, [10, 900], 100, (1, 1), (64, 64), seed=5, step=1)
    c = S.sample_hops(rowptr, col, [10, 900], 100, (1, 1), (64, 64), seed=5, step=2)
    assert all((x == y).all() for x, y in zip(a[1], b[1])) and (a[0] == b[0]).all()
    assert any((x != y).any() for x, y in zip(a[1], c[1]))


def test_capacity_overflow_is_counted_not_silent():
    _, _, rowptr, col = _store()
    gid, edges, dropped = S.sample_hops(rowptr, col, [300], 100, (5, 5), (4, 4), seed=1, step=1)
    assert dropped > 0
    for e in edges:                                     # edges to dropped nodes are not emitted
        live = e[0] >= 0
        assert (gid[e[0][live]] >= 0).all()


def test_members_oracle_invariants():
    """sample_members: a subgraph's groups are the contiguous range of its target window; every live edge slot i is (i, the slot
    of note i's own group inside ITS subgraph's block); groups beyond the capacity are counted."""
    from analysisgnn_amd.synth import make_score_graph
    graphs = [make_score_graph(seed=20 + i, n_notes=600, add_beats=True) for i in range(3)]
    start = np.concatenate([[0], np.cumsum([g.num_nodes["note"] for g in graphs])])
    goff = np.concatenate([[0], np.cumsum([g.num_nodes["beat"] for g in graphs])])
    group_of = np.concatenate([g.edge_index[("note", "connects", "beat")][1] + o for g, o in zip(graphs, goff[:-1])]).astype(np.int32)
    wins = [int(start[0]) + 50, int(start[2]) + 100]
    T, cap = 120, (16,)
    rng = np.random.default_rng(0)
    node_gid = np.full(2 * T + 2 * 16, -1, dtype=np.int32)
    for s, w in enumerate(wins):
        node_gid[s * T:(s + 1) * T] = np.arange(w, w + T)
        node_gid[2 * T + s * 16:2 * T + s * 16 + 9] = rng.integers(start[[0, 2][s]], start[[0, 2][s] + 1], 9)      # hop nodes of the same score
    for cap_g in (64, 10):
        ggid, edges, dropped = S.sample_members(node_gid, group_of, wins, T, cap, cap_g)
        for s, w in enumerate(wins):
            lo, hi = int(group_of[w]), int(group_of[w + T - 1])
            blk = ggid[s * cap_g:(s + 1) * cap_g]
            n = min(hi - lo + 1, cap_g)
            assert np.array_equal(blk[:n], np.arange(lo, lo + n)) and (blk[n:] == -1).all()
        live = edges[0] >= 0
        assert np.array_equal(edges[0][live], np.nonzero(live)[0])
        assert np.array_equal(ggid[edges[1][live]], group_of[node_gid[edges[0][live]]])
        assert (edges[1][~live] == -1).all() and (node_gid[~live & (np.arange(node_gid.size) < 2 * T)] >= 0).sum() == (0 if cap_g == 64 else (~live[:2 * T]).sum())
        assert (dropped > 0) == (cap_g == 10)


def test_compact_oracle_invariants():
    """sampler_ref.compact: hop order kept, a subgraph's nodes contiguous and in rank order, every live edge maps onto the same
    global ids as before, nothing lost while the pools have room."""
    _, _, rowptr, col = _store()
    wins, T, fan, cap = [30, 700, 1500], 100, (5, 5), (40, 40)
    gid, edges, _ = S.sample_hops(rowptr, col, wins, T, fan, cap, seed=3, step=2)
    B = len(wins)
    real = [int((gid[B * T + h * B * 40:B * T + (h + 1) * B * 40] >= 0).sum()) for h in range(2)]
    pool = (real[0] + 3, real[1] + 3)
    g2, e2, batch, dropped = S.compact(gid, edges, B, T, cap, pool)
    assert dropped == 0 and g2.shape[0] == B * T + sum(pool)
    assert sorted(g2[g2 >= 0].tolist()) == sorted(gid[gid >= 0].tolist())
    for h in range(2):
        blk = slice(B * T + sum(pool[:h]), B * T + sum(pool[:h + 1]))
        b = batch[blk][g2[blk] >= 0]
        assert (np.diff(b) >= 0).all()                          # subgraph order inside a pool
    for e_old, e_new in zip(edges, e2):
        live = e_old[0] >= 0
        assert np.array_equal(live, e_new[0] >= 0)
        assert np.array_equal(gid[e_old[0][live]], g2[e_new[0][live]]) and np.array_equal(gid[e_old[1][live]], g2[e_new[1][live]])
    _, e3, _, dropped3 = S.compact(gid, edges, B, T, cap, (real[0], max(real[1] - 4, 1)))
    assert dropped3 == min(4, real[1] - 1) and all((x[0] >= 0).sum() <= (y[0] >= 0).sum() for x, y in zip(e3, e2))
