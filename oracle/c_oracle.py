"""ctypes/numpy front end of oracle/c/libagnn_oracle.so.  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "c", "libagnn_oracle.so")
_lib = None


class CooSeg(C.Structure):
    _fields_ = [("row", C.c_void_p), ("col", C.c_void_p), ("etype", C.c_void_p),
                ("etype_code", C.c_int64), ("n_edges", C.c_int64), ("n_rows", C.c_int64)]


class Rel(C.Structure):
    _fields_ = [("src", C.c_void_p), ("rowptr", C.c_void_p), ("rowend", C.c_void_p), ("col", C.c_void_p),
                ("ew", C.c_void_p), ("colscale", C.c_void_p), ("ld_src", C.c_int64)]


def build() -> str:
    src = os.path.join(_HERE, "c", "agnn_oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "c/libagnn_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data


def csr_build(segs: Sequence[dict]):
    """segs: dicts with row, col (int64 arrays), n_rows, optional etype/code.
    Returns rowstart[int32 total_rows+1], col[int32 E_total], perm[int32 E_total], n_kept."""
    arr = (CooSeg * len(segs))()
    keep = []
    e_total = 0
    r_total = 0
    for i, s in enumerate(segs):
        row = np.ascontiguousarray(s["row"], dtype=np.int64)
        col = np.ascontiguousarray(s["col"], dtype=np.int64)
        et = np.ascontiguousarray(s["etype"], dtype=np.int64) if s.get("etype") is not None else None
        keep += [row, col, et]
        arr[i].row, arr[i].col, arr[i].etype = _p(row), _p(col), _p(et)
        arr[i].etype_code = int(s.get("code", 0))
        arr[i].n_edges = row.size
        arr[i].n_rows = int(s["n_rows"])
        e_total += row.size
        r_total += int(s["n_rows"])
    rowstart = np.zeros(r_total + 1, dtype=np.int32)
    col_o = np.zeros(max(e_total, 1), dtype=np.int32)
    perm_o = np.zeros(max(e_total, 1), dtype=np.int32)
    rc = lib().oracle_csr_build(len(segs), arr, C.c_void_p(_p(rowstart)), C.c_void_p(_p(col_o)), C.c_void_p(_p(perm_o)))
    assert rc == 0
    return rowstart, col_o, perm_o, int(rowstart[-1])


def csr_rowend(rowptr: np.ndarray, perm: np.ndarray, e_limit: int) -> np.ndarray:
    n = rowptr.size - 1
    out = np.zeros(max(n, 1), dtype=np.int32)
    rp = np.ascontiguousarray(rowptr, dtype=np.int32)
    lib().oracle_csr_rowend(C.c_void_p(_p(rp)), C.c_void_p(_p(perm)), C.c_int64(n), C.c_int64(e_limit), C.c_void_p(_p(out)))
    return out[:n]


def spmm(rels: Sequence[dict], n_rows: int, H: int, rel_stride: int, self_: Optional[np.ndarray] = None,
         mean: bool = True, skip_self: bool = False, col_limit: int = 2 ** 31 - 1, want_inv_cnt: bool = False):
    """rels: dicts with src [n_src, H] float32, rowptr, col, optional rowend/ew/colscale."""
    R = len(rels)
    arr = (Rel * R)()
    keep = []
    for i, r in enumerate(rels):
        src = np.ascontiguousarray(r["src"], dtype=np.float32)
        rp = np.ascontiguousarray(r["rowptr"], dtype=np.int32)
        col = np.ascontiguousarray(r["col"], dtype=np.int32)
        re = np.ascontiguousarray(r["rowend"], dtype=np.int32) if r.get("rowend") is not None else None
        ew = np.ascontiguousarray(r["ew"], dtype=np.float32) if r.get("ew") is not None else None
        cs = np.ascontiguousarray(r["colscale"], dtype=np.float32) if r.get("colscale") is not None else None
        keep += [src, rp, col, re, ew, cs]
        arr[i].src, arr[i].rowptr, arr[i].rowend, arr[i].col = _p(src), _p(rp), _p(re), _p(col)
        arr[i].ew, arr[i].colscale, arr[i].ld_src = _p(ew), _p(cs), src.shape[1] if src.ndim == 2 else H
    width = H if rel_stride == 0 else R * H
    out = np.zeros((n_rows, width), dtype=np.float32)
    inv = np.zeros((R, max(n_rows, 1)), dtype=np.float32) if want_inv_cnt else None
    sf = np.ascontiguousarray(self_, dtype=np.float32) if self_ is not None else None
    flags = (1 if mean else 0) | (2 if skip_self else 0)
    f = lib().oracle_spmm_f32
    rc = f(C.c_int(R), arr, C.c_int64(n_rows), C.c_int32(H), C.c_void_p(_p(out)), C.c_int64(width),
           C.c_int64(rel_stride), C.c_void_p(_p(sf)), C.c_int64(sf.shape[1] if sf is not None else 0),
           C.c_void_p(_p(inv)), C.c_int32(min(col_limit, 2 ** 31 - 1)), C.c_uint32(flags))
    assert rc == 0
    return (out, inv) if want_inv_cnt else out
