"""Hand-computed known-answer tests pinning oracle/scatter_ref.py (fixture F1)
and the onset-pool restatement (models/analysis.py:580-587, fixture F8)."""
import torch

from oracle.scatter_ref import scatter, scatter_add, scatter_mean
from oracle.intree_ref import onset_pool


def test_sum_into_prefilled_out():
    src = torch.tensor([[1., 2.], [3., 4.], [5., 6.], [7., 8.]])
    idx = torch.tensor([0, 0, 2, 4])
    out = torch.tensor([[10., 10.], [20., 20.], [30., 30.], [40., 40.], [50., 50.]])
    got = scatter(src, idx, 0, out=out.clone(), reduce="sum")
    exp = torch.tensor([[14., 16.], [20., 20.], [35., 36.], [40., 40.], [57., 58.]])
    assert torch.equal(got, exp)
    assert torch.equal(scatter_add(src, idx, 0, out=out.clone()), exp)


def test_mean_divides_prefilled_out_by_neighbour_count_only():
    # node 0: (10 + 1 + 3) / 2 ; node 1: 20 / max(0,1) ; node 2: (30 + 5) / 1
    src = torch.tensor([[1.], [3.], [5.]])
    idx = torch.tensor([0, 0, 2])
    out = torch.tensor([[10.], [20.], [30.]])
    got = scatter_mean(src, idx, 0, out=out.clone())
    assert torch.allclose(got, torch.tensor([[7.], [20.], [35.]]))


def test_dim_size_and_default_size():
    src = torch.ones(3, 2)
    idx = torch.tensor([1, 1, 3])
    assert scatter(src, idx, 0, dim_size=6, reduce="sum").shape == (6, 2)
    got = scatter(src, idx, 0, reduce="mean")
    assert got.shape == (4, 2)
    assert torch.equal(got, torch.tensor([[0., 0.], [1., 1.], [0., 0.], [1., 1.]]))


def test_empty_index_keeps_out():
    out = torch.tensor([[2., 3.]])
    got = scatter_mean(torch.zeros(0, 2), torch.zeros(0, dtype=torch.long), 0, out=out.clone())
    assert torch.equal(got, out)


def test_mean_gradients():
    x = torch.tensor([[1.], [2.], [3.]], requires_grad=True)
    h = torch.tensor([[10.], [20.], [30.]], requires_grad=True)
    dst = torch.tensor([0, 0])
    src = torch.tensor([1, 2])
    s = scatter(h[src], dst, 0, out=x.clone(), reduce="mean")
    s.sum().backward()
    assert torch.allclose(x.grad, torch.tensor([[0.5], [1.], [1.]]))      # grad_out / count
    assert torch.allclose(h.grad, torch.tensor([[0.], [0.5], [0.5]]))     # grad_out[index] / count


def test_onset_pool_three_note_chord_divides_by_two():
    # notes 0,1,2 share an onset (all 9 onset edges incl. self loops); note 3 alone; note 4 beyond batch_size
    x = torch.tensor([[3.], [6.], [9.], [5.], [100.]])
    pairs = [(i, j) for i in range(3) for j in range(3)] + [(3, 3), (4, 4), (4, 0), (0, 4)]
    e = torch.tensor(pairs).t().contiguous()
    got = onset_pool(x[:4], e, batch_size=4)
    # (x_i + sum_{j != i} x_j) / 2 = 18 / 2 = 9 for the chord ; lone note stays itself
    exp_pool = torch.tensor([[9.], [9.], [9.], [5.]])
    assert torch.allclose(got, torch.cat([x[:4], exp_pool], dim=-1))
