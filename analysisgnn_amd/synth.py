"""Synthetic score graphs (numpy only; no torch, no GPU).

The reference's datasets need network access, partitura and graphmuse, none of which exist
offline, so every benchmark / parity input is synthetic.  The generator follows the edge rules
of the reference's in-tree graph builder ``hetero_graph_from_note_array``
(/root/reference/analysisgnn/utils/hgraph.py:232-285) as restated in SURVEY.md Appendix B:

* ``onset``        i -> j  iff  onset_i == onset_j            (self loops kept, see
                                                               models/analysis.py:583-584)
* ``consecutive``  i -> j  iff  onset_i + dur_i == onset_j    (hgraph.py:244-247)
* ``during``       i -> j  iff  onset_i < onset_j < onset_i + dur_i   (hgraph.py:255-259)
* ``rest``         for every note-end time with no onset there: all notes ending then ->
                   all notes of the next onset                (hgraph.py:274-285)

Edges are PyG convention ``edge_index[0] = source i``, ``edge_index[1] = target j``.
Beat nodes = one per 4 divs, measure nodes = one per 16 divs, with membership edges.

Seed 0 / 500 notes gives 1380 / 604 / 793 / 528 edges, 112 beats, 28 measures (checked in
tests/test_synth.py against the numbers quoted in SURVEY.md §8).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

NOTE_RELATIONS = ("onset", "consecutive", "during", "rest")
EdgeType = Tuple[str, str, str]


@dataclass
class ScoreGraph:
    """One (or a block-diagonal batch of) synthetic score graph(s), numpy arrays only."""

    num_nodes: Dict[str, int]
    edge_index: Dict[EdgeType, np.ndarray]          # int64 [2, E]
    batch: Dict[str, np.ndarray]                    # int64 [N_t] subgraph id
    onset_div: np.ndarray                           # int64 [N_note]
    duration_div: np.ndarray                        # int64 [N_note]
    num_graphs: int = 1
    # optional neighbour-sampling bookkeeping (PyG NeighborLoader convention)
    num_sampled_nodes: Optional[Dict[str, List[int]]] = None
    num_sampled_edges: Optional[Dict[EdgeType, List[int]]] = None
    batch_size: Optional[int] = None                # number of target notes (first rows)
    extras: Dict[str, np.ndarray] = field(default_factory=dict)

    @property
    def node_types(self) -> List[str]:
        return list(self.num_nodes.keys())

    @property
    def edge_types(self) -> List[EdgeType]:
        return list(self.edge_index.keys())

    def metadata(self) -> Tuple[List[str], List[EdgeType]]:
        return self.node_types, self.edge_types


def _note_times(seed: int, n_notes: int) -> Tuple[np.ndarray, np.ndarray]:
    rng = np.random.default_rng(seed)
    on: List[int] = []
    du: List[int] = []
    t = 0
    while len(on) < n_notes:
        k = int(rng.choice([1, 2, 3, 4], p=[0.35, 0.3, 0.2, 0.15]))
        ds = [int(rng.choice([1, 2, 4, 8], p=[0.3, 0.4, 0.2, 0.1])) for _ in range(k)]
        adv = int(rng.choice([1, 2, 4], p=[0.4, 0.4, 0.2]))
        for d in ds:
            on.append(t)
            du.append(d)
        t += adv
    return np.asarray(on[:n_notes], dtype=np.int64), np.asarray(du[:n_notes], dtype=np.int64)


def _pairs(mask: np.ndarray) -> np.ndarray:
    """Row-major (i, then j) list of True positions, like the reference's nested loops."""
    src, dst = np.nonzero(mask)
    return np.stack([src, dst]).astype(np.int64)


def make_score_graph(
    seed: int = 0,
    n_notes: int = 500,
    add_beats: bool = False,
    add_measures: bool = False,
    reverse_note_edges: bool = False,
    reverse_metrical_edges: bool = False,
) -> ScoreGraph:
    """One synthetic score graph (reference hgraph.py:232-285)."""
    on, du = _note_times(seed, n_notes)
    end = on + du
    ei: Dict[EdgeType, np.ndarray] = {}
    ei[("note", "onset", "note")] = _pairs(on[:, None] == on[None, :])
    ei[("note", "consecutive", "note")] = _pairs(end[:, None] == on[None, :])
    ei[("note", "during", "note")] = _pairs((on[:, None] < on[None, :]) & (end[:, None] > on[None, :]))
    rs: List[int] = []
    rd: List[int] = []
    onsets = set(on.tolist())
    for et in np.sort(np.unique(end))[:-1]:
        if int(et) in onsets:
            continue
        diffs = on - et
        pos = diffs > 0
        if not pos.any():
            continue
        nxt = diffs[pos].min()
        dst = np.nonzero(diffs == nxt)[0]
        for i in np.nonzero(end == et)[0]:
            for j in dst:
                rs.append(int(i))
                rd.append(int(j))
    ei[("note", "rest", "note")] = np.asarray([rs, rd], dtype=np.int64).reshape(2, -1)
    if reverse_note_edges:
        for rel in ("consecutive", "during", "rest"):
            ei[("note", rel + "_rev", "note")] = ei[("note", rel, "note")][::-1].copy()

    num_nodes = {"note": int(n_notes)}
    batch = {"note": np.zeros(n_notes, dtype=np.int64)}
    note_ids = np.arange(n_notes, dtype=np.int64)
    beat_of = on // 4
    meas_of = on // 16
    if add_beats:
        ub, bidx = np.unique(beat_of, return_inverse=True)
        num_nodes["beat"] = int(ub.size)
        batch["beat"] = np.zeros(ub.size, dtype=np.int64)
        ei[("note", "connects", "beat")] = np.stack([note_ids, bidx.astype(np.int64)])
        if reverse_metrical_edges:
            ei[("beat", "rev_connects", "note")] = np.stack([bidx.astype(np.int64), note_ids])
    if add_measures:
        um, midx = np.unique(meas_of, return_inverse=True)
        num_nodes["measure"] = int(um.size)
        batch["measure"] = np.zeros(um.size, dtype=np.int64)
        ei[("note", "connects", "measure")] = np.stack([note_ids, midx.astype(np.int64)])
        if reverse_metrical_edges:
            ei[("measure", "rev_connects", "note")] = np.stack([midx.astype(np.int64), note_ids])
        if add_beats:
            ub = np.unique(beat_of)
            b2m = np.searchsorted(um, ub // 4)
            ei[("beat", "connects", "measure")] = np.stack(
                [np.arange(ub.size, dtype=np.int64), b2m.astype(np.int64)])
            if reverse_metrical_edges:
                ei[("measure", "rev_connects", "beat")] = ei[("beat", "connects", "measure")][::-1].copy()
    return ScoreGraph(num_nodes=num_nodes, edge_index=ei, batch=batch, onset_div=on,
                      duration_div=du, num_graphs=1, batch_size=int(n_notes))


def collate(graphs: Sequence[ScoreGraph]) -> ScoreGraph:
    """Block-diagonal batch: node ids offset per subgraph, ``batch`` = subgraph id.

    Mirrors what the reference's loader hands to ``TorchAnalysisGNN.encode``
    (models/analysis.py:948-961): disconnected subgraphs, no cross edges.
    All graphs must be un-sampled (every note is a target).
    """
    node_types = graphs[0].node_types
    edge_types = graphs[0].edge_types
    offs = {t: 0 for t in node_types}
    ei: Dict[EdgeType, List[np.ndarray]] = {et: [] for et in edge_types}
    batch: Dict[str, List[np.ndarray]] = {t: [] for t in node_types}
    on: List[np.ndarray] = []
    du: List[np.ndarray] = []
    for gi, g in enumerate(graphs):
        for et in edge_types:
            s, _, d = et
            e = g.edge_index[et].copy()
            e[0] += offs[s]
            e[1] += offs[d]
            ei[et].append(e)
        for t in node_types:
            batch[t].append(np.full(g.num_nodes[t], gi, dtype=np.int64))
        on.append(g.onset_div)
        du.append(g.duration_div)
        for t in node_types:
            offs[t] += g.num_nodes[t]
    return ScoreGraph(
        num_nodes=dict(offs),
        edge_index={et: np.concatenate(v, axis=1) for et, v in ei.items()},
        batch={t: np.concatenate(v) for t, v in batch.items()},
        onset_div=np.concatenate(on), duration_div=np.concatenate(du),
        num_graphs=len(graphs), batch_size=int(offs["note"]))


def make_batch(n_graphs: int, n_notes: int = 500, first_seed: int = 0, seeds: Optional[Sequence[int]] = None, **kw) -> ScoreGraph:
    """Batch of ``n_graphs`` subgraphs with seeds first_seed .. first_seed+n_graphs-1 (or the given ``seeds``: a
    data-parallel rank's share {i : i mod G = rank} of the global batch, dp.shard_units)."""
    seeds = list(seeds) if seeds is not None else [first_seed + i for i in range(n_graphs)]
    return collate([make_score_graph(seed=sd, n_notes=n_notes, **kw) for sd in seeds])


def sample_hops(g: ScoreGraph, n_targets: int, num_neighbors: Sequence[int], seed: int = 0,
                random_targets: bool = False, first_target: int = 0) -> ScoreGraph:
    """Neighbour-sampled view of a single note-only graph in PyG NeighborLoader layout.

    Targets are the ``n_targets`` notes from ``first_target`` on (a window, as MuseNeighborLoader takes —
    reference datamodules/analysis.py:270-278); hop h samples up to ``num_neighbors[h]``
    in-neighbours per relation for every node of hop h.  Nodes are hop-ordered, edges are
    hop-ordered per relation, ``num_sampled_nodes/edges`` hold per-hop counts exactly as
    ``trim_to_layer`` expects.
    """
    assert set(g.num_nodes) == {"note"}, "sample_hops handles note-only graphs"
    rng = np.random.default_rng(seed)
    n = g.num_nodes["note"]
    order = (sorted(rng.choice(n, size=n_targets, replace=False).tolist()) if random_targets
             else list(range(first_target, first_target + n_targets)))
    pos = {v: i for i, v in enumerate(order)}
    nodes_per_hop = [n_targets]
    in_lists = {}
    for et, e in g.edge_index.items():
        lst: Dict[int, List[int]] = {}
        for s, d in zip(e[0].tolist(), e[1].tolist()):
            lst.setdefault(d, []).append(s)
        in_lists[et] = lst
    new_edges = {et: [] for et in g.edge_index}
    edges_per_hop = {et: [] for et in g.edge_index}
    frontier = list(order)
    for fan in num_neighbors:
        nxt: List[int] = []
        for et in g.edge_index:
            cnt = 0
            for d in frontier:
                cand = in_lists[et].get(d, [])
                if len(cand) > fan:
                    cand = rng.choice(cand, size=fan, replace=False).tolist()
                for s in cand:
                    if s not in pos:
                        pos[s] = len(order)
                        order.append(s)
                        nxt.append(s)
                    new_edges[et].append((pos[s], pos[d]))
                    cnt += 1
            edges_per_hop[et].append(cnt)
        nodes_per_hop.append(len(nxt))
        frontier = nxt
    order_arr = np.asarray(order, dtype=np.int64)
    ei = {et: np.asarray(v, dtype=np.int64).reshape(-1, 2).T.copy() for et, v in new_edges.items()}
    out = ScoreGraph(
        num_nodes={"note": int(order_arr.size)}, edge_index=ei,
        batch={"note": np.zeros(order_arr.size, dtype=np.int64)},
        onset_div=g.onset_div[order_arr], duration_div=g.duration_div[order_arr], num_graphs=1,
        num_sampled_nodes={"note": nodes_per_hop},
        num_sampled_edges={et: v for et, v in edges_per_hop.items()},
        batch_size=int(n_targets))
    out.extras["orig_id"] = order_arr
    assert n >= order_arr.size
    return out


def merge_sampled(samples: Sequence[ScoreGraph]) -> ScoreGraph:
    """Joint hop-ordered batch of neighbour-sampled subgraphs, the layout a NeighborLoader-style loader hands over for
    a batch of seeds (reference datamodules/analysis.py:270-278 with transform_to_pyg; consumed at
    models/analysis.py:948-961): nodes = [targets of all subgraphs | hop-1 nodes of all | hop-2 ...], per relation
    edges = [hop-1 edges of all | hop-2 ...], `num_sampled_nodes/edges` = per-hop totals, `batch_size` = all targets.
    Subgraphs stay disconnected (block structure), `batch` holds the subgraph id of every node."""
    hops = len(samples[0].num_sampled_nodes["note"])
    ets = samples[0].edge_types
    # node id maps: subgraph-local (hop-ordered) -> global
    maps = []
    base = 0
    hop_tot = [sum(s.num_sampled_nodes["note"][h] for s in samples) for h in range(hops)]
    hop_base = np.concatenate([[0], np.cumsum(hop_tot)])
    fill = [0] * hops
    for s in samples:
        m = np.empty(s.num_nodes["note"], dtype=np.int64)
        lo = 0
        for h in range(hops):
            k = s.num_sampled_nodes["note"][h]
            m[lo:lo + k] = hop_base[h] + fill[h] + np.arange(k)
            fill[h] += k
            lo += k
        maps.append(m)
    n_total = int(hop_base[-1])
    batch = np.empty(n_total, dtype=np.int64)
    on = np.empty(n_total, dtype=np.int64)
    du = np.empty(n_total, dtype=np.int64)
    for i, (s, m) in enumerate(zip(samples, maps)):
        batch[m] = i
        on[m] = s.onset_div
        du[m] = s.duration_div
    ei = {}
    eph = {}
    for et in ets:
        parts = []
        counts = []
        for h in range(hops - 1):
            c = 0
            for s, m in zip(samples, maps):
                lo = sum(s.num_sampled_edges[et][:h])
                k = s.num_sampled_edges[et][h]
                parts.append(m[s.edge_index[et][:, lo:lo + k]])
                c += k
            counts.append(c)
        ei[et] = (np.concatenate(parts, axis=1) if parts else np.zeros((2, 0), dtype=np.int64)).astype(np.int64)
        eph[et] = counts
    return ScoreGraph(num_nodes={"note": n_total}, edge_index=ei, batch={"note": batch}, onset_div=on, duration_div=du,
                      num_graphs=len(samples), num_sampled_nodes={"note": [int(v) for v in hop_tot]}, num_sampled_edges=eph,
                      batch_size=int(hop_tot[0]))


def make_sampled_batch(n_graphs: int, n_targets: int = 500, num_neighbors: Sequence[int] = (5, 5), first_seed: int = 0,
                       score_notes: int = 800, first_target: int = 150, seeds: Optional[Sequence[int]] = None) -> ScoreGraph:
    """`n_graphs` windows of `n_targets` target notes, each cut out of its own `score_notes`-note synthetic score and
    neighbour-sampled with `num_neighbors` (the reference trains with subgraph_size 500 and [5] * (num_layers - 1):
    train/train_analysisgnn.py:82,154), merged hop-ordered (`merge_sampled`)."""
    seeds = list(seeds) if seeds is not None else [first_seed + i for i in range(n_graphs)]
    return merge_sampled([sample_hops(make_score_graph(seed=sd, n_notes=score_notes), n_targets, num_neighbors,
                                      seed=sd, first_target=first_target) for sd in seeds])


def torch_inputs(g: ScoreGraph, in_channels: int = 25, device="cpu", seed: int = 0):
    """Tensors in the HeteroData layout `TorchAnalysisGNN.encode` consumes:
    x_dict ~ N(0,1) float32, int64 COO edge_index_dict, batch_dict, pitch_spelling ~ U{0..34},
    key_signature ~ U{0..14}, batch_size and the PyG per-hop count dicts (or None)."""
    import torch
    gen = torch.Generator().manual_seed(seed)
    x_dict = {t: torch.randn(n, in_channels, generator=gen).to(device) for t, n in g.num_nodes.items()}
    n_note = g.num_nodes["note"]
    batch_dict = {t: torch.from_numpy(b).to(device) for t, b in g.batch.items()}
    bs = int(g.batch_size if g.batch_size is not None else n_note)
    # the per-subgraph target counts are host knowledge at collation time: carried with the batch tensor so that the
    # sequence branch does not have to read them back from the device (`bincount(batch).tolist()`, models/analysis.py:529-530)
    batch_dict["note"].agnn_target_lengths = np.bincount(g.batch["note"][:bs], minlength=g.num_graphs).tolist()
    return dict(
        x_dict=x_dict,
        edge_index_dict={et: torch.from_numpy(np.ascontiguousarray(e)).to(device) for et, e in g.edge_index.items()},
        batch_dict=batch_dict,
        pitch_spelling=torch.randint(0, 35, (n_note,), generator=gen).to(device),
        key_signature=torch.randint(0, 15, (n_note,), generator=gen).to(device),
        batch_size=int(g.batch_size if g.batch_size is not None else n_note),
        neighbor_mask_node=g.num_sampled_nodes,
        neighbor_mask_edge=g.num_sampled_edges,
    )
