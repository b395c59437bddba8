"""Device-side graph indices: COO int64 (what PyG / the reference hand over) -> CSR int32, both
directions, all relations of a batch in ONE `agnn_csr_build` call.

The reference never builds an index: every layer re-masks `edge_index[:, edge_type == r]`
(analysisgnn/models/core/hgnn.py:137-139) and every scatter works on the unsorted COO.  Here
the index is built once per batch and shared by all layers and by forward and backward.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib

EdgeType = Tuple[str, str, str]


@dataclass
class Csr:
    """One relation in one direction.  `rowptr` holds offsets into the shared `col`/`perm`."""
    rowptr: torch.Tensor          # int32 [n_rows + 1] (view into the batch's rowstart array)
    col: torch.Tensor             # int32 [E_total]  shared by all segments of the build
    perm: torch.Tensor            # int32 [E_total]  original edge position inside the segment
    n_rows: int
    n_edges: int                  # capacity (edges handed in; masked-out ones excluded from rows)
    _rowend: Dict[int, torch.Tensor] = field(default_factory=dict, repr=False)

    def rowend(self, e_limit: Optional[int]) -> Optional[torch.Tensor]:
        """Row ends when only the COO prefix [0, e_limit) is kept (PyG trim_to_layer,
        reference models/cadence.py:167-173).  None = all edges."""
        if e_limit is None or e_limit >= self.n_edges:
            return None
        t = self._rowend.get(e_limit)
        if t is None:
            t = torch.empty(max(self.n_rows, 1), dtype=torch.int32, device=self.rowptr.device)
            lib = _lib.load()
            _lib.check(lib.agnn_csr_rowend(self.rowptr.data_ptr(), self.perm.data_ptr(), self.n_rows,
                                           int(e_limit), t.data_ptr(), _lib.stream_ptr(t.device)),
                       "agnn_csr_rowend")
            self._rowend[e_limit] = t
        return t


@dataclass
class SegSpec:
    row: torch.Tensor             # int64 [E]
    col: torch.Tensor             # int64 [E]
    n_rows: int
    etype: Optional[torch.Tensor] = None
    code: int = 0


def _i64c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.int64:
        raise _lib.AgnnError(f"edge indices must be int64, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def build_csr(specs: Sequence[SegSpec]) -> List[Csr]:
    """All segments in one pass (one key space, one stable radix sort)."""
    if not specs:
        return []
    if len(specs) > _lib.MAX_SEG:
        out: List[Csr] = []
        for i in range(0, len(specs), _lib.MAX_SEG):
            out += build_csr(specs[i:i + _lib.MAX_SEG])
        return out
    dev = _lib.require_gpu(*[s.row for s in specs], *[s.col for s in specs])
    lib = _lib.load()
    keep = []                      # keep contiguous copies alive until the call is enqueued
    segs = (_lib.CooSeg * len(specs))()
    e_total = 0
    r_total = 0
    for i, s in enumerate(specs):
        row, col = _i64c(s.row), _i64c(s.col)
        if row.numel() != col.numel():
            raise _lib.AgnnError("row/col length mismatch")
        et = _i64c(s.etype) if s.etype is not None else None
        keep += [row, col, et]
        segs[i].row = row.data_ptr() if row.numel() else None
        segs[i].col = col.data_ptr() if col.numel() else None
        segs[i].etype = et.data_ptr() if (et is not None and et.numel()) else None
        segs[i].etype_code = int(s.code)
        segs[i].n_edges = row.numel()
        segs[i].n_rows = int(s.n_rows)
        e_total += row.numel()
        r_total += int(s.n_rows)
    rowstart = torch.empty(r_total + 1, dtype=torch.int32, device=dev)
    col = torch.empty(max(e_total, 1), dtype=torch.int32, device=dev)
    perm = torch.empty(max(e_total, 1), dtype=torch.int32, device=dev)
    ws_bytes = int(lib.agnn_csr_workspace_bytes(e_total, r_total))
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    _lib.check(lib.agnn_csr_build(len(specs), segs, rowstart.data_ptr(), col.data_ptr(), perm.data_ptr(),
                                  ws.data_ptr(), ws_bytes, _lib.status_word(dev).data_ptr(), _lib.stream_ptr(dev)),
               "agnn_csr_build")
    out = []
    base = 0
    for s in specs:
        out.append(Csr(rowptr=rowstart[base:base + s.n_rows + 1], col=col, perm=perm,
                       n_rows=int(s.n_rows), n_edges=int(s.row.numel())))
        base += s.n_rows
    return out


class HeteroIndex:
    """CSR by destination (`fwd`) and by source (`bwd`) for every relation of a PyG-convention
    `edge_index_dict` (row 0 = source, row 1 = target)."""

    def __init__(self, edge_index_dict: Dict[EdgeType, torch.Tensor], num_nodes: Dict[str, int]):
        self.edge_types: List[EdgeType] = list(edge_index_dict.keys())
        self.num_nodes = dict(num_nodes)
        self.num_edges = {et: int(ei.shape[1]) for et, ei in edge_index_dict.items()}
        specs: List[SegSpec] = []
        for et in self.edge_types:
            s, _, d = et
            ei = edge_index_dict[et]
            specs.append(SegSpec(row=ei[1], col=ei[0], n_rows=num_nodes[d]))
        for et in self.edge_types:
            s, _, d = et
            ei = edge_index_dict[et]
            specs.append(SegSpec(row=ei[0], col=ei[1], n_rows=num_nodes[s]))
        csrs = build_csr(specs)
        n = len(self.edge_types)
        self.fwd: Dict[EdgeType, Csr] = {et: csrs[i] for i, et in enumerate(self.edge_types)}
        self.bwd: Dict[EdgeType, Csr] = {et: csrs[n + i] for i, et in enumerate(self.edge_types)}


    def prepare_trim(self, e_keep_layers: Sequence[Dict[EdgeType, Optional[int]]]) -> None:
        """Row ends of every relation, in both directions, for every COO prefix the layers of a trimmed (sampled) batch
        keep (PyG trim_to_layer, reference models/cadence.py:165-173) — ONE launch (`agnn_csr_rowend_batch`) instead of
        one per (relation, direction, layer) on first use.  Fills the `Csr.rowend` caches."""
        todo = []
        for e_keep in e_keep_layers:
            for et, lim in e_keep.items():
                if lim is None or et not in self.fwd:
                    continue
                for csr in (self.fwd[et], self.bwd[et]):
                    if lim < csr.n_edges and lim not in csr._rowend and csr.n_rows > 0 and not any(
                            c is csr and l == lim for c, l in todo):
                        todo.append((csr, int(lim)))
        if not todo:
            return
        lib = _lib.load()
        dev = todo[0][0].rowptr.device
        buf = torch.empty(sum(c.n_rows for c, _ in todo), dtype=torch.int32, device=dev)
        base = 0
        for i in range(0, len(todo), _lib.ROWEND_MAX_ITEMS):
            chunk = todo[i:i + _lib.ROWEND_MAX_ITEMS]
            items = (_lib.RowendItem * len(chunk))()
            for j, (csr, lim) in enumerate(chunk):
                out = buf[base:base + csr.n_rows]
                base += csr.n_rows
                items[j].rowptr, items[j].perm, items[j].rowend = csr.rowptr.data_ptr(), csr.perm.data_ptr(), out.data_ptr()
                items[j].n_rows, items[j].e_limit = csr.n_rows, lim
                csr._rowend[lim] = out
            _lib.check(lib.agnn_csr_rowend_batch(len(chunk), items, _lib.stream_ptr(dev)), "agnn_csr_rowend_batch")


_INDEX_CACHE: "Dict[tuple, HeteroIndex]" = {}
_INDEX_CACHE_MAX = 8
index_cache_enabled = True


def hetero_index(edge_index_dict: Dict[EdgeType, torch.Tensor], num_nodes: Dict[str, int]) -> HeteroIndex:
    """Build (or fetch) the batch index.  Keyed on the identity + version of the COO tensors so the
    L layers of a forward, its backward and repeated calls on one batch share one build."""
    if not index_cache_enabled:
        return HeteroIndex(edge_index_dict, num_nodes)
    key = tuple((et, ei.data_ptr(), tuple(ei.shape), ei._version, num_nodes[et[0]], num_nodes[et[2]])
                for et, ei in edge_index_dict.items())
    hit = _INDEX_CACHE.get(key)
    if hit is None:
        hit = HeteroIndex(edge_index_dict, num_nodes)
        if len(_INDEX_CACHE) >= _INDEX_CACHE_MAX:
            _INDEX_CACHE.pop(next(iter(_INDEX_CACHE)))
        _INDEX_CACHE[key] = hit
        hit._keepalive = list(edge_index_dict.values())   # data_ptr stays unique while cached
    return hit


def clear_index_cache() -> None:
    _INDEX_CACHE.clear()
