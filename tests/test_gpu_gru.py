"""Persistent GRU kernels vs the explicit-recurrence oracle (oracle/rnn_ref.py, itself pinned against
torch.nn.GRU).  fp32; tolerance 1e-4 relative to max(1,|ref|max) for outputs, input gradients and all
weight gradients (observed ~1e-6)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import assert_close  # noqa: E402

DEV = "cuda:0"


def _run(B, T, I, layers, seed, fast_oracle=False):
    from analysisgnn_amd.gru import gru_forward
    from oracle import rnn_ref
    torch.manual_seed(seed)
    m = torch.nn.GRU(I, 128, num_layers=layers, batch_first=True, bidirectional=True)
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


@pytest.mark.parametrize("B,T,I,layers", [(1, 1, 8, 1), (3, 37, 64, 1), (2, 19, 256, 2), (5, 64, 32, 2)])
def test_gru_small(B, T, I, layers):
    _run(B, T, I, layers, seed=B * 100 + T)


def test_gru_c2_shape():
    """32 sequences x 500 steps, 256 -> 128 x 2, two layers: the hybrid branch of BASELINE config C2."""
    _run(32, 500, 256, 2, seed=7, fast_oracle=True)


def test_other_hidden_sizes_use_library_rnn():
    from analysisgnn_amd.gru import gru_forward, kernel_applicable
    m = torch.nn.GRU(16, 32, num_layers=2, batch_first=True, bidirectional=True).to(DEV)
    assert not kernel_applicable(m)
    x = torch.randn(2, 5, 16, device=DEV)
    assert torch.allclose(gru_forward(m, x, False), m(x)[0])
