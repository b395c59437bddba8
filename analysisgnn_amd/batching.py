"""Device-side batch assembly: the scores of a corpus stay resident in HBM and every training
batch — windows of target notes plus their sampled hop neighbours, block-diagonal over the subgraphs, hop-ordered as
PyG's NeighborLoader lays it out — is produced by ONE kernel launch (`agnn_sample_hops`) into buffers of static shape.
Replaces graphmuse's `MuseNeighborLoader`, its collation and the host-to-device copy of every batch
(reference analysisgnn/data/datamodules/analysis.py:270-293; consumed at models/analysis.py:948-961).

Static shapes are what lets the WHOLE step (sampling, feature gather, CSR build, forward, backward, optimizer) be captured
once and replayed with a different batch every time: hop blocks are padded to a capacity, padding nodes carry no features
and no edges, padding edge slots hold (-1, -1) and are dropped by the CSR build, and `num_sampled_nodes/edges` are the
capacities, so `trim_to_layer` trims whole padded blocks.  Only the 32 window starts cross the PCIe bus per step.
Neighbour sampling runs over the relations among notes (as `synth.sample_hops`); when the scores carry metrical nodes
(beats, measures — the reference's default, data/datamodules/analysis.py:213-225) a second launch per metrical type
(`agnn_sample_members`) adds each subgraph's beats / measures and the membership edges of its notes.  No CPU path."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .graph import SegSpec, build_csr

EdgeType = Tuple[str, str, str]


class ScoreStore:
    """All scores of a corpus on the device: note features / attributes concatenated over the scores (global note ids) and,
    per relation, one CSR by DESTINATION over all notes (built once with `agnn_csr_build`)."""

    def __init__(self, graphs: Sequence, in_channels: int, device, tasks: Optional[Dict[str, int]] = None, seed: int = 0):
        """`graphs`: synth.ScoreGraph objects (numpy).  Features / spellings / keys / labels are synthetic (seeded), as
        everywhere in this repo: the reference's datasets cannot be fetched offline."""
        dev = torch.device(device)
        self.device = dev
        self.edge_types: List[EdgeType] = [et for et in graphs[0].edge_types if et[0] == "note" and et[2] == "note"]
        sizes = [int(g.num_nodes["note"]) for g in graphs]
        self.score_start = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        n = int(self.score_start[-1])
        self.num_notes = n
        gen = torch.Generator().manual_seed(seed)
        width = (in_channels + 3) & ~3                       # feature rows padded to 16 bytes (zero spare columns)
        x = torch.zeros(n, width)
        x[:, :in_channels] = torch.randn(n, in_channels, generator=gen)
        self.in_channels = in_channels
        self.x = x.to(dev)
        attrs = [torch.randint(0, 35, (n,), generator=gen), torch.randint(0, 15, (n,), generator=gen)]
        self.tasks = dict(tasks) if tasks else {}
        for c in self.tasks.values():
            attrs.append(torch.randint(0, c, (n,), generator=gen))
        self.attrs = torch.stack(attrs).to(dev)              # int64 [2 + T, n]: pitch spelling, key signature, labels
        self.onset_div = torch.from_numpy(np.concatenate([g.onset_div for g in graphs])).to(dev)
        specs, self._keep = [], []
        for et in self.edge_types:
            src = np.concatenate([g.edge_index[et][0] + o for g, o in zip(graphs, self.score_start[:-1])])
            dst = np.concatenate([g.edge_index[et][1] + o for g, o in zip(graphs, self.score_start[:-1])])
            s_t, d_t = torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev)
            self._keep += [s_t, d_t]
            specs.append(SegSpec(row=d_t, col=s_t, n_rows=n))
        self.csr = build_csr(specs)                          # rows = destination note, col = source note
        # metrical node types: per note the GLOBAL id of its beat / measure (scores concatenated), and synthetic features
        self.group_types: List[str] = [t for t in ("beat", "measure") if ("note", "connects", t) in graphs[0].edge_index]
        self.group_of: Dict[str, torch.Tensor] = {}
        self.group_x: Dict[str, torch.Tensor] = {}
        for t in self.group_types:
            ids, off = [], 0
            for g in graphs:
                e = g.edge_index[("note", "connects", t)]
                if not np.array_equal(e[0], np.arange(g.num_nodes["note"])) or np.any(np.diff(e[1]) < 0):
                    raise _lib.AgnnError(f"ScoreStore: ('note', 'connects', '{t}') must list every note once, groups non-decreasing")
                ids.append(e[1] + off)
                off += int(g.num_nodes[t])
            self.group_of[t] = torch.from_numpy(np.concatenate(ids).astype(np.int32)).to(dev)
            xg = torch.zeros(off, width)
            xg[:, :in_channels] = torch.randn(off, in_channels, generator=gen)
            self.group_x[t] = xg.to(dev)

    def random_windows(self, n_sub: int, n_targets: int, rng: np.random.Generator) -> np.ndarray:
        """What the loader's sampler decides on the host: `n_sub` (score, first target) pairs -> global ids, int32."""
        ok = np.nonzero(np.diff(self.score_start) >= n_targets)[0]
        scores = rng.choice(ok, size=n_sub, replace=len(ok) < n_sub)
        lo = self.score_start[scores]
        hi = self.score_start[scores + 1] - n_targets
        return (lo + (rng.random(n_sub) * (hi - lo + 1)).astype(np.int64)).astype(np.int32)


class DeviceSampler:
    """Static-shape sampled batches out of a `ScoreStore`.  `sample()` refills the same tensors every call (a captured
    graph replays it); `batch` is the dict `TorchAnalysisGNN.encode` consumes (the keys of `synth.torch_inputs`)."""

    def __init__(self, store: ScoreStore, n_sub: int, n_targets: int = 500, num_neighbors: Sequence[int] = (5, 5),
                 capacity: Sequence[int] = (64, 64), seed: int = 0, group_capacity: Optional[Dict[str, int]] = None,
                 pool: Optional[Sequence[int]] = None):
        """`capacity[h]`: new nodes a subgraph may add in hop h.  `pool[h]` (optional): the batch-wide number of slots of hop h —
        subgraphs rarely fill their capacity, so the padded blocks (n_sub x capacity[h] rows each) are squeezed into pools by a
        second launch (`agnn_sample_compact`): fewer padding rows through every layer, the shapes stay static."""
        if len(capacity) != len(num_neighbors):
            raise ValueError("one capacity per hop")
        self.store, self.n_sub, self.n_targets = store, int(n_sub), int(n_targets)
        self.fan, self.cap = [int(v) for v in num_neighbors], [int(v) for v in capacity]
        dev = store.device
        lib = _lib.load()
        cfg = _lib.Sampler()
        cfg.n_rel = len(store.edge_types)
        for r, csr in enumerate(store.csr):
            cfg.rowptr[r], cfg.col[r] = csr.rowptr.data_ptr(), csr.col.data_ptr()
        cfg.n_sub, cfg.n_targets, cfg.n_hops = self.n_sub, self.n_targets, len(self.fan)
        for h, (f, c) in enumerate(zip(self.fan, self.cap)):
            cfg.fan[h], cfg.cap[h] = f, c
        self.num_nodes = int(lib.agnn_sampler_num_nodes(cfg))
        self.e_cap = int(lib.agnn_sampler_edge_capacity(cfg))
        if self.num_nodes < 0 or self.e_cap < 0:
            raise _lib.AgnnError("bad sampler configuration")
        self.win_start = torch.zeros(self.n_sub, dtype=torch.int32, device=dev)
        self.rng = torch.tensor([int(seed), 0], dtype=torch.int64, device=dev)       # (seed, step): bump step per batch
        self.pool = [int(v) for v in pool] if pool is not None else None
        if self.pool is not None and len(self.pool) != len(self.cap):
            raise ValueError("one pool size per hop")
        self.node_gid = torch.empty(self.num_nodes, dtype=torch.int32, device=dev)     # what agnn_sample_hops writes (padded blocks)
        self.edges = {et: torch.empty((2, self.e_cap), dtype=torch.int64, device=dev) for et in store.edge_types}
        cfg.win_start, cfg.rng, cfg.node_gid = self.win_start.data_ptr(), self.rng.data_ptr(), self.node_gid.data_ptr()
        for r, et in enumerate(store.edge_types):
            cfg.edges[r] = self.edges[et].data_ptr()
        cfg.e_cap = self.e_cap
        cfg.status = _lib.status_word(dev).data_ptr()
        self.drops = torch.zeros(1, dtype=torch.int32, device=dev)     # statistic, not an error: see dropped()
        cfg.drops = self.drops.data_ptr()
        self._cfg = cfg
        B, T = self.n_sub, self.n_targets
        self.batch_size = B * T
        blocks = [B * c for c in self.cap]
        if self.pool is not None:
            self._kept = torch.zeros(len(self.cap) * B, dtype=torch.int32, device=dev)
            cfg.kept = self._kept.data_ptr()
            self._pool_arr = (_lib.C.c_int32 * len(self.pool))(*self.pool)
            n_pool = int(lib.agnn_sample_compact_nodes(cfg, self._pool_arr))
            if n_pool < 0:
                raise _lib.AgnnError("bad pool configuration")
            self._gid_padded, self.num_nodes = self.node_gid, n_pool
            self.node_gid = torch.empty(n_pool, dtype=torch.int32, device=dev)          # pool layout: what everything downstream reads
            blocks = list(self.pool)
        # static per-hop block sizes = what trim_to_layer is driven by (PyG num_sampled_nodes / num_sampled_edges)
        self.num_sampled_nodes = {"note": [B * T] + blocks}
        e_per_hop, F = [], T
        for f, c in zip(self.fan, self.cap):
            e_per_hop.append(B * F * f)
            F = c
        self.num_sampled_edges = {et: list(e_per_hop) for et in store.edge_types}
        sub = [np.repeat(np.arange(B), T)] + [np.repeat(np.arange(B), c) for c in self.cap]
        if self.pool is not None:                     # rewritten by every sample(): which subgraph a pool slot serves
            sub = [sub[0], np.zeros(sum(self.pool), dtype=np.int64)]
        self.batch_note = torch.from_numpy(np.concatenate(sub).astype(np.int64)).to(dev)
        self.batch_note.agnn_target_lengths = [T] * B
        self.x = torch.empty((self.num_nodes, store.x.shape[1]), dtype=torch.float32, device=dev)
        self.attrs = torch.empty((store.attrs.shape[0], self.num_nodes), dtype=torch.int64, device=dev)
        # metrical node types: one static block of `group_capacity[t]` slots per subgraph (all hop 0: never trimmed), one
        # membership edge slot per batch note (trimmed with its note's hop block)
        self.group_cap = {t: int((group_capacity or {}).get(t, {"beat": 160, "measure": 48}[t])) for t in store.group_types}
        self.group_gid, self.group_x = {}, {}
        x_dict = {"note": self.x[:, :store.in_channels]}
        batch_dict = {"note": self.batch_note}
        self._cap_arr = (_lib.C.c_int32 * max(len(self.cap), 1))(*self.cap)
        for t in store.group_types:
            cg = self.group_cap[t]
            et = ("note", "connects", t)
            self.group_gid[t] = torch.empty(B * cg, dtype=torch.int32, device=dev)
            self.edges[et] = torch.empty((2, self.num_nodes), dtype=torch.int64, device=dev)
            self.group_x[t] = torch.empty((B * cg, store.group_x[t].shape[1]), dtype=torch.float32, device=dev)
            self.num_sampled_nodes[t] = [B * cg] + [0] * len(self.cap)
            self.num_sampled_edges[et] = [B * T + (blocks[0] if blocks else 0)] + blocks[1:]
            x_dict[t] = self.group_x[t][:, :store.in_channels]
            batch_dict[t] = torch.from_numpy(np.repeat(np.arange(B), cg).astype(np.int64)).to(dev)
        self.batch = dict(
            x_dict=x_dict, edge_index_dict=self.edges, batch_dict=batch_dict,
            pitch_spelling=self.attrs[0], key_signature=self.attrs[1], batch_size=self.batch_size,
            neighbor_mask_node=self.num_sampled_nodes, neighbor_mask_edge=self.num_sampled_edges,
            labels={t: self.attrs[2 + i, :self.batch_size] for i, t in enumerate(store.tasks)},
            label_matrix=self.attrs[2:, :self.batch_size])

    def set_windows(self, win_start: np.ndarray) -> None:
        """The only per-batch host-to-device traffic: n_sub int32 (outside the captured graph, like any input refill).
        Staged through a small ring of PINNED buffers: a copy from pageable memory makes the host wait for the stream, i.e.
        for the previous step, and the host then cannot run ahead of the device any more."""
        if not hasattr(self, "_ring"):
            self._ring = [torch.empty(self.n_sub, dtype=torch.int32).pin_memory() for _ in range(16)]
            self._ring_ev = [None] * len(self._ring)       # per slot: the event behind the copy that last read it
            self._ring_pos = 0
        k = self._ring_pos
        self._ring_pos = (k + 1) % len(self._ring)
        if self._ring_ev[k] is not None:
            self._ring_ev[k].synchronize()                 # a host 16 steps ahead waits here instead of overwriting a slot in flight
        buf = self._ring[k]
        buf.numpy()[:] = win_start
        self.win_start.copy_(buf, non_blocking=True)
        ev = self._ring_ev[k] or torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.store.device))
        self._ring_ev[k] = ev

    def metadata(self):
        """(node types, edge types) of the batches this sampler writes — what the model constructors take."""
        return (["note"] + list(self.store.group_types),
                list(self.store.edge_types) + [("note", "connects", t) for t in self.store.group_types])

    def dropped(self) -> int:
        """Sources cut by the hop capacities since this sampler was created (synchronises; a statistic, not an error)."""
        return int(self.drops.item())

    def sample(self) -> dict:
        """Advance the step counter, sample, gather features and attributes — four launches, graph-capturable."""
        dev = self.store.device
        lib = _lib.load()
        st = _lib.stream_ptr(dev)
        self.rng[1:2].add_(1)
        _lib.check(lib.agnn_sample_hops(self._cfg, st), "agnn_sample_hops")
        if self.pool is not None:
            _lib.check(lib.agnn_sample_compact(self._cfg, self._pool_arr, self.node_gid.data_ptr(), self.batch_note.data_ptr(), st),
                       "agnn_sample_compact")
        s = self.store
        _lib.check(lib.agnn_gather_rows_f32(s.x.data_ptr(), s.x.stride(0), self.node_gid.data_ptr(), self.num_nodes, s.x.shape[1],
                                            self.x.data_ptr(), self.x.stride(0), st), "agnn_gather_rows_f32")
        _lib.check(lib.agnn_gather_i64(s.attrs.data_ptr(), s.attrs.stride(0), self.node_gid.data_ptr(), self.num_nodes, s.attrs.shape[0], 0,
                                       self.attrs.data_ptr(), self.attrs.stride(0), st), "agnn_gather_i64")
        for t in s.group_types:          # + two launches per metrical type: members, their feature rows
            cg = self.group_cap[t]
            _lib.check(lib.agnn_sample_members(self.node_gid.data_ptr(), self.num_nodes,
                                               self.batch_note.data_ptr() if self.pool is not None else None,
                                               s.group_of[t].data_ptr(), self.win_start.data_ptr(),
                                               self.n_sub, self.n_targets, len(self.cap), self._cap_arr, cg, self.group_gid[t].data_ptr(),
                                               self.edges[("note", "connects", t)].data_ptr(), self.drops.data_ptr(), st), "agnn_sample_members")
            gx = s.group_x[t]
            _lib.check(lib.agnn_gather_rows_f32(gx.data_ptr(), gx.stride(0), self.group_gid[t].data_ptr(), self.n_sub * cg, gx.shape[1],
                                                self.group_x[t].data_ptr(), self.group_x[t].stride(0), st), "agnn_gather_rows_f32")
        return self.batch
