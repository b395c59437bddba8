"""The C-ABI library loads without a GPU and exports every symbol include/agnn.h declares;
the Python binding declares a signature for each of them.  No compute call is made here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "agnn.h")
SO = os.path.join(ROOT, "analysisgnn_amd", "libagnn_hip.so")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(agnn_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    for need in ("agnn_csr_build", "agnn_spmm_f32", "agnn_last_error"):
        assert need in syms


@pytest.mark.skipif(not os.path.exists(SO), reason="libagnn_hip.so not built (run __graft_entry__.build())")
def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(SO)
    for s in declared_symbols():
        assert hasattr(lib, s), f"{s} declared in include/agnn.h but not exported"


def test_python_binding_covers_header():
    from analysisgnn_amd import _lib
    assert sorted(_lib.SIGNATURES.keys()) == declared_symbols()


def test_cpu_tensors_are_refused():
    import torch
    from analysisgnn_amd import _lib
    with pytest.raises(_lib.AgnnError):
        _lib.require_gpu(torch.zeros(2, 4))


@pytest.mark.skipif(not os.path.exists(SO), reason="libagnn_hip.so not built")
def test_argument_validation_without_gpu():
    """Bad arguments are rejected before any HIP call (safe on a CPU-only host)."""
    from analysisgnn_amd import _lib
    lib = _lib.load()
    rels = (_lib.Rel * 1)()
    rc = lib.agnn_spmm_f32(1, rels, 4, 6, None, 8, 0, None, 0, None, 0, 0, None)   # H % 4 != 0
    assert rc == -22 and b"H=6" in lib.agnn_last_error()
    rc = lib.agnn_spmm_f32(99, rels, 4, 8, None, 8, 0, None, 0, None, 0, 0, None)
    assert rc == -22
    segs = (_lib.CooSeg * 1)()
    assert lib.agnn_csr_build(0, segs, None, None, None, None, 0, None, None) == -22
    assert lib.agnn_check_status(None, None) == -22
