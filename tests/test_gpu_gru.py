"""Persistent GRU kernels vs the explicit-recurrence oracle (oracle/rnn_ref.py, itself pinned against
torch.nn.GRU).  fp32; tolerance 1e-4 relative to max(1,|ref|max) for outputs, input gradients and all
weight gradients (observed ~1e-6)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import assert_close  # noqa: E402

DEV = "cuda:0"


def _run(B, T, I, layers, seed, fast_oracle=False, hidden=128):
    from analysisgnn_amd.gru import gru_forward
    from oracle import rnn_ref
    torch.manual_seed(seed)
    m = torch.nn.GRU(I, hidden, num_layers=layers, batch_first=True, bidirectional=True)
    with torch.no_grad():                      # larger recurrent weights: make the chain matter
        for n, p in m.named_parameters():
            if "weight_hh" in n:
                p.mul_(2.0)
    x = torch.randn(B, T, I)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    xc = x.clone().requires_grad_(True)
    ref = (rnn_ref.gru_fast if fast_oracle else rnn_ref._gru_loops)(P, "", xc, layers, True)
    gout = torch.randn(ref.shape, generator=torch.Generator().manual_seed(seed + 1))
    (ref * gout).sum().backward()
    mg = m.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    out = gru_forward(mg, xg, training=False)
    assert_close(out, ref, 1e-4, "y")
    (out * gout.to(DEV)).sum().backward()
    assert_close(xg.grad, xc.grad, 1e-4, "dx")
    for n, p in mg.named_parameters():
        assert_close(p.grad, P[n].grad, 1e-4, f"d{n}")


@pytest.mark.parametrize("B,T,I,layers", [(1, 1, 8, 1), (2, 2, 8, 2), (2, 3, 8, 1), (1, 4, 8, 2), (3, 5, 16, 1), (2, 6, 8, 1), (2, 7, 8, 2),
                                           (3, 37, 64, 1), (2, 19, 256, 2), (5, 64, 32, 2)])
def test_gru_small(B, T, I, layers):
    _run(B, T, I, layers, seed=B * 100 + T)


@pytest.mark.parametrize("B,T,I,layers", [(1, 1, 8, 1), (3, 37, 64, 1), (4, 50, 128, 2)])
def test_gru_hidden_64(B, T, I, layers):
    """The hidden-64 instantiation (HybridGNN / HybridHGT at H = 128): three rows of W_hh per lane, eight columns per wave, the
    gate math on wave 0 only; backward with half the lanes carrying units."""
    from analysisgnn_amd.gru import kernel_applicable
    assert kernel_applicable(torch.nn.GRU(I, 64, num_layers=layers, batch_first=True, bidirectional=True))
    _run(B, T, I, layers, seed=B * 10 + T, hidden=64)


def test_gru_c2_shape():
    """32 sequences x 500 steps, 256 -> 128 x 2, two layers: the hybrid branch of BASELINE config C2."""
    _run(32, 500, 256, 2, seed=7, fast_oracle=True)


def test_other_hidden_sizes_use_library_rnn():
    from analysisgnn_amd.gru import gru_forward, kernel_applicable
    m = torch.nn.GRU(16, 32, num_layers=2, batch_first=True, bidirectional=True).to(DEV)
    assert not kernel_applicable(m)
    x = torch.randn(2, 5, 16, device=DEV)
    assert torch.allclose(gru_forward(m, x, False), m(x)[0])


@pytest.mark.parametrize("B,T,I", [(3, 41, 64), (32, 120, 256)])
def test_inter_layer_dropout_rides_with_the_recurrence_kernels(B, T, I):
    """nn.GRU(dropout=p) between the two layers (models/cadence.py:249-251) as a factor the first layer's kernels apply
    themselves: with a GIVEN 0 / (1 / (1 - p)) pattern, outputs and all gradients equal those of two one-layer GRUs with
    `y0 * pattern` between them (torch.nn.GRU on the GPU as the yardstick); and in training mode about p of the first layer's
    outputs really are dropped."""
    from analysisgnn_amd.gru import _GRULayer, _stack_gru_params, gru_forward
    torch.manual_seed(B + T)
    m = torch.nn.GRU(I, 128, num_layers=2, batch_first=True, bidirectional=True, dropout=0.3).to(DEV)
    x = torch.randn(B, T, I, device=DEV)
    pattern = (torch.rand(B, T, 256, device=DEV) >= 0.3).float() / 0.7
    # reference: the two layers as separate torch GRUs with the pattern applied in between
    l0 = torch.nn.GRU(I, 128, batch_first=True, bidirectional=True).to(DEV)
    l1 = torch.nn.GRU(256, 128, batch_first=True, bidirectional=True).to(DEV)
    sd = m.state_dict()
    l0.load_state_dict({k: sd[k] for k in l0.state_dict()})
    l1.load_state_dict({k: sd[k.replace("_l0", "_l1")] for k in l1.state_dict()})
    xr = x.clone().requires_grad_(True)
    ref = l1(l0(xr)[0] * pattern)[0]
    gout = torch.randn_like(ref)
    (ref * gout).sum().backward()
    xg = x.clone().requires_grad_(True)
    st = _stack_gru_params(m)
    y0 = _GRULayer.apply(xg, st[0], st[1], st[2], st[3], True, pattern)
    out = _GRULayer.apply(y0, st[4], st[5], st[6], st[7], False, None)
    assert_close(out, ref, 1e-4, "y")
    (out * gout).sum().backward()
    assert_close(xg.grad, xr.grad, 1e-4, "dx")
    for n, p in m.named_parameters():
        src = l0 if n.endswith("_l0") or n.endswith("_l0_reverse") else l1
        pr = dict(src.named_parameters())[n.replace("_l1", "_l0")]
        assert_close(p.grad, pr.grad, 1e-4, f"d{n}")
    with torch.no_grad():
        a = gru_forward(m.train(), x, training=True)
        b = gru_forward(m, x, training=True)
        c = gru_forward(m, x, training=False)
    assert not torch.equal(a, b) and not torch.equal(a, c)


@pytest.mark.parametrize("B,T,hidden", [(1, 1, 128), (3, 37, 128), (4, 50, 64)])
def test_backward_walk_leaves_previous_states(B, T, hidden):
    """agnn_gru_bwd_f32's optional `hprev` output = the matrix agnn_gru_hprev_f32 builds from y (bit for bit: both copy y)."""
    from analysisgnn_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)                                 # noqa: E731
    gi, w, b = r(B, T, 2, 3 * hidden), r(2, 3 * hidden, hidden) * 0.1, r(2, 3 * hidden)
    y = torch.empty(B, T, 2 * hidden, device=DEV)
    saved = torch.empty(B, T, 2, 4, hidden, device=DEV)
    st = _lib.stream_ptr(torch.device(DEV))
    _lib.check(lib.agnn_gru_fwd_f32(gi.data_ptr(), w.data_ptr(), b.data_ptr(), B, T, hidden, y.data_ptr(), saved.data_ptr(), None, None, st), "fwd")
    dy = r(B, T, 2 * hidden)
    dgi, dgh = torch.empty_like(gi), torch.empty_like(gi)
    hp = torch.full((B, T, 2, hidden), float("nan"), device=DEV)
    _lib.check(lib.agnn_gru_bwd_f32(dy.data_ptr(), y.data_ptr(), saved.data_ptr(), w.data_ptr(), B, T, hidden, dgi.data_ptr(),
                                    dgh.data_ptr(), None, hp.data_ptr(), st), "bwd")
    want = torch.empty_like(hp)
    _lib.check(lib.agnn_gru_hprev_f32(y.data_ptr(), B, T, hidden, want.data_ptr(), st), "hprev")
    assert torch.equal(hp, want)
    dgi2, dgh2 = torch.empty_like(gi), torch.empty_like(gi)
    _lib.check(lib.agnn_gru_bwd_f32(dy.data_ptr(), y.data_ptr(), saved.data_ptr(), w.data_ptr(), B, T, hidden, dgi2.data_ptr(),
                                    dgh2.data_ptr(), None, None, st), "bwd without hprev")
    assert torch.equal(dgi, dgi2) and torch.equal(dgh, dgh2)


@pytest.mark.parametrize("fill", [float("nan"), float("inf"), -1e38])
@pytest.mark.parametrize("B,T,hidden", [(2, 40, 128), (3, 23, 64)])
def test_forward_without_dropout_ignores_what_the_output_buffer_held(B, T, hidden, fill):
    """Without inter-layer dropout the kernels get the output buffer as a stand-in for the dropout-scale operand (its values
    must not matter).  The loader reads that stand-in AHEAD of the walk, i.e. memory the kernel has not written yet: whatever the
    allocator left there — NaN, Inf — must not reach the result (it did while the stand-in was neutralised by 0 * x instead of a
    select: a training run diverged whenever the recycled block held a non-finite value).  Same result, bit for bit, as into a
    zero-filled buffer; the backward kernel likewise with a poisoned `hprev` / gradient buffers."""
    from analysisgnn_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)                                 # noqa: E731
    gi, w, b = r(B, T, 2, 3 * hidden), r(2, 3 * hidden, hidden) * 0.1, r(2, 3 * hidden)
    st = _lib.stream_ptr(torch.device(DEV))
    outs = []
    for f in (0.0, fill):
        y = torch.full((B, T, 2 * hidden), f, device=DEV)
        saved = torch.full((B, T, 2, 4, hidden), f, device=DEV)
        _lib.check(lib.agnn_gru_fwd_f32(gi.data_ptr(), w.data_ptr(), b.data_ptr(), B, T, hidden, y.data_ptr(), saved.data_ptr(), None, None, st), "fwd")
        dy = torch.ones(B, T, 2 * hidden, device=DEV)
        dgi, dgh = torch.full_like(gi, f), torch.full_like(gi, f)
        hp = torch.full((B, T, 2, hidden), f, device=DEV)
        _lib.check(lib.agnn_gru_bwd_f32(dy.data_ptr(), y.data_ptr(), saved.data_ptr(), w.data_ptr(), B, T, hidden, dgi.data_ptr(),
                                        dgh.data_ptr(), None, hp.data_ptr(), st), "bwd")
        outs.append((y, saved, dgi, dgh, hp))
    for a, c in zip(*outs):
        assert torch.isfinite(c).all()
        assert torch.equal(a, c)
