"""Live float64 check of oracle/intree_ref.py against the reference's own files.

Runs ONLY in the build container (skipped wherever /root/reference is absent, e.g. the GPU
box).  Loads core/gnn.py + core/hgnn.py exactly as oracle/gen_golden.py does and compares in
float64, where rounding noise (fp32 BatchNorm backward) cannot mask a semantic difference.
"""
import os

import numpy as np
import pytest
import torch

from helpers import load_golden

REF = "/root/reference/analysisgnn/models/core"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")


@pytest.fixture(scope="module")
def ref():
    import sys
    sys.dont_write_bytecode = True
    from oracle.gen_golden import load_reference_core
    return load_reference_core()


@pytest.fixture()
def f64():
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)   # the reference allocates `torch.zeros(...)` without dtype
    yield
    torch.set_default_dtype(old)


@pytest.mark.parametrize("tag", ["eq", "ragged"])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_metrical_gnn_f64(ref, f64, tag, mode):
    from oracle import intree_ref as R
    gnn, hgnn = ref
    z = load_golden(f"metrical_{tag}_{mode}")
    rels = [str(r) for r in z["meta.rels"]]
    m = hgnn.MetricalGNN(8, 8, 8, etypes={r: i for i, r in enumerate(rels)}, num_layers=3, dropout=0.0,
                         metrical=True)
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w.")}
    m.load_state_dict(sd)
    m.double().train(mode == "train")
    I = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in.")}
    x = I["x"].double().requires_grad_(True)
    out = m(x, I["edge_index"], I["edge_type"], I["beat_nodes"], I["measure_nodes"], I["beat_edges"],
            I["measure_edges"], beat_lengths=I.get("beat_lengths"), measure_lengths=I.get("measure_lengths"))
    g = torch.from_numpy(z["gout"]).double()
    (out * g).sum().backward()
    P = {k: (v.double().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    x2 = I["x"].double().requires_grad_(True)
    o2 = R.metrical_gnn(P, rels, 3, x2, I["edge_index"], I["edge_type"], I["beat_nodes"].numel(),
                        I["measure_nodes"].numel(), I["beat_edges"], I["measure_edges"],
                        I.get("beat_lengths"), I.get("measure_lengths"), training=(mode == "train"))
    (o2 * g).sum().backward()
    assert float((out - o2).abs().max()) < 1e-10
    assert float((x.grad - x2.grad).abs().max()) < 1e-10
    for n, p in m.named_parameters():
        assert float((p.grad - P[n].grad).abs().max()) < 1e-9, n


def test_hgcn_f64(ref, f64):
    from oracle import intree_ref as R
    gnn, hgnn = ref
    z = load_golden("hgcn3_jk")
    rels = [str(r) for r in z["meta.rels"]]
    m = hgnn.HGCN(8, 16, 8, n_layers=2, etypes={r: i for i, r in enumerate(rels)}, dropout=0.0, jk=True)
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w.")}
    m.load_state_dict(sd)
    m.double()
    I = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in.")}
    x = I["x"].double().requires_grad_(True)
    out = m(x, I["edge_index"], I["edge_type"])
    out.sum().backward()
    P = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    x2 = I["x"].double().requires_grad_(True)
    o2 = R.hgcn(P, rels, 3, x2, I["edge_index"], I["edge_type"], jk=True)
    o2.sum().backward()
    assert float((out - o2).abs().max()) < 1e-11
    assert float((x.grad - x2.grad).abs().max()) < 1e-10
