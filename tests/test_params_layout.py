"""Host logic (no GPU): parameters that the fused schedule consumes concatenated are laid out back to back by
dp.plan_parameters / FlatGradBuffer / FlatAdamW, and params.cat_rows / stack_rows then return views of the flat buffer
with the gradients of torch.cat / torch.stack."""
import torch
import torch.nn as nn

from analysisgnn_amd import dp
from analysisgnn_amd.params import adjacent, cat_rows, stack_rows


class _Heads(nn.Module):
    def __init__(self, sizes=(5, 3, 8)):
        super().__init__()
        self.first = nn.Linear(6, 6)
        self.heads = nn.ModuleList([nn.Linear(4, c) for c in sizes])
        self.rnn = nn.GRU(4, 4, num_layers=1, bidirectional=True, batch_first=True)

    def adjacent_parameter_groups(self):
        return [[h.weight for h in self.heads], [h.bias for h in self.heads],
                [self.rnn.weight_ih_l0, self.rnn.weight_ih_l0_reverse]]


def test_aligned_offsets_with_tight_members():
    assert dp._aligned_offsets([5, 3, 8, 2]) == [0, 8, 12, 20, 24]                       # every slot on a 4-float boundary
    assert dp._aligned_offsets([5, 3, 8, 2], tight=[False, True, True, False]) == [0, 5, 8, 16, 20]
    assert dp._aligned_offsets([4, 4]) == [0, 4, 8]


def test_plan_parameters_orders_groups_first_and_keeps_every_parameter_once():
    m = _Heads()
    params, tight = dp.plan_parameters(m)
    assert len(params) == len(list(m.parameters())) and len({id(p) for p in params}) == len(params)
    assert [id(p) for p in params[:3]] == [id(h.weight) for h in m.heads]
    assert id(params[0]) not in tight and id(params[1]) in tight and id(params[2]) in tight
    assert id(m.rnn.weight_ih_l0_reverse) in tight and id(m.first.weight) not in tight


def test_cat_rows_is_a_view_after_flattening_and_matches_torch_cat():
    torch.manual_seed(0)
    m = _Heads()
    ref = [h.weight.detach().clone() for h in m.heads], [h.bias.detach().clone() for h in m.heads]
    assert not adjacent([h.bias for h in m.heads])                                       # separate allocations: a real cat
    assert torch.equal(cat_rows([h.bias for h in m.heads]), torch.cat(ref[1]))
    params, tight = dp.plan_parameters(m)
    flat = dp.FlatGradBuffer(params, views=False, tight=tight)
    opt = dp.FlatAdamW(params, flat, lr=1e-2)
    assert adjacent([h.weight for h in m.heads]) and adjacent([h.bias for h in m.heads])  # biases of 5, 3, 8: no padding inside
    W = cat_rows([h.weight for h in m.heads])
    b = cat_rows([h.bias for h in m.heads])
    S = stack_rows([m.rnn.weight_ih_l0, m.rnn.weight_ih_l0_reverse])
    assert W.untyped_storage().data_ptr() == opt.flat.untyped_storage().data_ptr()       # views of the flat parameter buffer
    assert torch.equal(W, torch.cat(ref[0])) and torch.equal(b, torch.cat(ref[1]))
    assert S.shape == (2, 12, 4) and torch.equal(S[1], m.rnn.weight_ih_l0_reverse)
    gW, gb, gS = torch.randn_like(W), torch.randn_like(b), torch.randn_like(S)
    flat.zero()
    ((W * gW).sum() + (b * gb).sum() + (S * gS).sum()).backward()
    r0 = 0
    for h in m.heads:
        c = h.weight.shape[0]
        assert torch.equal(h.weight.grad, gW[r0:r0 + c]) and torch.equal(h.bias.grad, gb[r0:r0 + c])
        r0 += c
    assert torch.equal(m.rnn.weight_ih_l0_reverse.grad, gS[1])
    flat.pack()                                                                          # gradients land at the parameters' offsets
    for p, o in zip(flat.params, flat.offsets):
        assert torch.equal(flat.flat[o:o + p.numel()].view_as(p), p.grad if p.grad is not None else torch.zeros_like(p))
    before = m.heads[1].bias.detach().clone()
    opt.step()
    assert not torch.equal(m.heads[1].bias, before)                                      # the update reaches the views
    assert torch.equal(cat_rows([h.bias for h in m.heads])[5:8], m.heads[1].bias)
