"""Pin the plain-C kernel-level oracle (oracle/c/agnn_oracle.c) against the torch restatement of
torch_scatter (oracle/scatter_ref.py) and against a golden fixture produced by the reference."""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle.scatter_ref import scatter
from helpers import load_golden


def _rand_graph(rng, n, e):
    return rng.integers(0, n, size=e).astype(np.int64), rng.integers(0, n, size=e).astype(np.int64)


def test_csr_build_is_stable_grouping():
    rng = np.random.default_rng(0)
    row, col = _rand_graph(rng, 17, 200)
    rs, c, p, kept = c_oracle.csr_build([dict(row=row, col=col, n_rows=17)])
    assert kept == 200 and rs[0] == 0 and rs[-1] == 200
    for i in range(17):
        seg = p[rs[i]:rs[i + 1]]
        assert np.all(row[seg] == i)
        assert np.all(np.diff(seg) > 0)            # original order kept
        assert np.all(c[rs[i]:rs[i + 1]] == col[seg])


def test_csr_build_etype_mask_and_multi_segment():
    rng = np.random.default_rng(1)
    row, col = _rand_graph(rng, 9, 60)
    et = rng.integers(0, 3, size=60).astype(np.int64)
    segs = [dict(row=row, col=col, n_rows=9, etype=et, code=k) for k in range(4)]   # code 3: empty
    rs, c, p, kept = c_oracle.csr_build(segs)
    assert kept == 60
    for k in range(4):
        rp = rs[9 * k: 9 * k + 10]
        ids = p[rp[0]:rp[-1]]
        assert np.all(et[ids] == k) if ids.size else k == 3
        assert ids.size == int((et == k).sum())


def test_rowend_prefix():
    rng = np.random.default_rng(2)
    row, col = _rand_graph(rng, 11, 80)
    rs, c, p, _ = c_oracle.csr_build([dict(row=row, col=col, n_rows=11)])
    re = c_oracle.csr_rowend(rs, p, 30)
    for i in range(11):
        assert re[i] - rs[i] == int(((row == i) & (np.arange(80) < 30)).sum())


@pytest.mark.parametrize("mean", [True, False])
@pytest.mark.parametrize("with_self", [True, False])
def test_spmm_matches_scatter_ref(mean, with_self):
    rng = np.random.default_rng(3)
    n, e, H = 23, 90, 12
    row, col = _rand_graph(rng, n, e)
    h = rng.standard_normal((n, H)).astype(np.float32)
    x = rng.standard_normal((n, H)).astype(np.float32)
    rs, c, p, _ = c_oracle.csr_build([dict(row=row, col=col, n_rows=n)])
    got = c_oracle.spmm([dict(src=h, rowptr=rs, col=c)], n, H, 0, self_=x if with_self else None, mean=mean)
    out0 = torch.from_numpy(x).clone() if with_self else torch.zeros(n, H)
    exp = scatter(torch.from_numpy(h)[torch.from_numpy(col)], torch.from_numpy(row), 0, out=out0,
                  reduce="mean" if mean else "sum").numpy()
    np.testing.assert_allclose(got, exp, rtol=2e-6, atol=2e-6)


def test_spmm_onset_pool_filters():
    from oracle.intree_ref import onset_pool
    rng = np.random.default_rng(4)
    n, B, H = 30, 20, 8
    row, col = _rand_graph(rng, n, 150)
    row[:10] = col[:10]                           # self loops
    x = rng.standard_normal((B, H)).astype(np.float32)
    rs, c, p, _ = c_oracle.csr_build([dict(row=row, col=col, n_rows=n)])
    got = c_oracle.spmm([dict(src=x, rowptr=rs, col=c)], B, H, 0, self_=x, mean=True, skip_self=True, col_limit=B)
    exp = onset_pool(torch.from_numpy(x), torch.from_numpy(np.stack([row, col])), B).numpy()[:, H:]
    np.testing.assert_allclose(got, exp, rtol=2e-6, atol=2e-6)


def test_spmm_against_reference_fixture():
    """sage_small fixture (reference SageConvScatter run): s = (x + sum h_j)/max(deg,1) then linear."""
    z = load_golden("sage_small")
    x, ei = z["in.x"], z["in.edge_index"]
    h = x @ z["w.neigh_linear.weight"].T + z["w.neigh_linear.bias"]
    rs, c, p, _ = c_oracle.csr_build([dict(row=ei[0], col=ei[1], n_rows=x.shape[0])])
    s = c_oracle.spmm([dict(src=h.astype(np.float32), rowptr=rs, col=c)], x.shape[0], x.shape[1], 0, self_=x, mean=True)
    out = np.concatenate([x, s], axis=1) @ z["w.linear.weight"].T + z["w.linear.bias"]
    np.testing.assert_allclose(out, z["out"], rtol=1e-5, atol=1e-5)
