"""Fused [ReLU->]LayerNorm[->ReLU][->dropout] kernels vs plain PyTorch fp32 (the nn.ReLU / nn.LayerNorm / nn.Dropout
chains of models/analysis.py:429-443, :474-485).  Tolerance 1e-4 relative to max(1,|ref|max)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from helpers import assert_close  # noqa: E402

DEV = "cuda:0"


@pytest.mark.parametrize("H", [8, 64, 256, 260, 512, 1024])
@pytest.mark.parametrize("pre,post", [(False, False), (True, False), (False, True)])
def test_norm_act_no_dropout(H, pre, post):
    from analysisgnn_amd.fused import norm_act
    torch.manual_seed(H)
    ln = nn.LayerNorm(H)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5)
        ln.bias.uniform_(-0.5, 0.5)
    x = torch.randn(301, H) * 2 + 0.3
    xr = x.clone().requires_grad_(True)
    ref = ln(F.relu(xr) if pre else xr)
    ref = F.relu(ref) if post else ref
    g = torch.randn(ref.shape)
    (ref * g).sum().backward()
    lg = nn.LayerNorm(H).to(DEV)
    lg.load_state_dict(ln.state_dict())
    xg = x.to(DEV).requires_grad_(True)
    out = norm_act(xg, lg, pre_relu=pre, post_relu=post, p=0.3, training=False)        # eval: dropout is the identity
    assert_close(out, ref, 1e-4, "y")
    (out * g.to(DEV)).sum().backward()
    assert_close(xg.grad, xr.grad, 1e-4, "dx")
    assert_close(lg.weight.grad, ln.weight.grad, 1e-4, "dgamma")
    assert_close(lg.bias.grad, ln.bias.grad, 1e-4, "dbeta")


def test_dropout_mask_statistics_determinism_and_backward():
    from analysisgnn_amd import fused
    torch.manual_seed(0)
    H, N, p = 256, 4000, 0.3
    ln = nn.LayerNorm(H).to(DEV)
    x = torch.randn(N, H, device=DEV, requires_grad=True)
    base = fused._NormAct.apply(x, ln.weight, ln.bias, ln.eps, 0.0, fused.POST_RELU, 7)
    y1 = fused._NormAct.apply(x, ln.weight, ln.bias, ln.eps, p, fused.POST_RELU, 7)
    y2 = fused._NormAct.apply(x, ln.weight, ln.bias, ln.eps, p, fused.POST_RELU, 7)
    assert torch.equal(y1, y2)                                      # same (seed, step, call) -> same mask
    y3 = fused._NormAct.apply(x, ln.weight, ln.bias, ln.eps, p, fused.POST_RELU, 8)
    assert not torch.equal(y1, y3)                                  # another call site -> another mask
    fused.advance_rng(x.device)
    y4 = fused._NormAct.apply(x, ln.weight, ln.bias, ln.eps, p, fused.POST_RELU, 7)
    assert not torch.equal(y1, y4)                                  # next training step -> another mask
    live = base > 0
    kept = (y4 != 0) & live
    rate = kept.sum().item() / live.sum().item()
    assert abs(rate - (1 - p)) < 0.01, rate
    assert torch.allclose(y4[kept], base[kept] / (1 - p), rtol=1e-5, atol=1e-6)
    # backward regenerates the same mask: compare with autograd through the unfused ops using the observed mask
    mask = ((y4 != 0) | ~live).float() / (1 - p)
    xr = x.detach().clone().requires_grad_(True)
    ref = F.relu(F.layer_norm(xr, (H,), ln.weight, ln.bias, ln.eps)) * mask
    g = torch.randn(N, H, device=DEV)
    (ref * g).sum().backward()
    x.grad = None
    (y4 * g).sum().backward()
    assert_close(x.grad, xr.grad, 1e-4, "dx with dropout")


def test_two_training_forwards_before_one_backward():
    """A second training-mode forward (which bumps the device step counter: models.encode -> advance_rng) BEFORE the first
    one's backward — two batches with a joint backward, gradient accumulation, the reference's memory replay
    (models/analysis.py:1064-1066, :1327-1366).  The first call's backward must regenerate the masks of ITS forward: the
    (seed, step) pair travels with the call's saved state, not with the live counter."""
    from analysisgnn_amd import fused
    torch.manual_seed(1)
    H, N, p = 256, 1000, 0.3
    ln = nn.LayerNorm(H).to(DEV)
    x = torch.randn(N, H, device=DEV)
    g = torch.randn(N, H, device=DEV)

    def grads(second_forward_in_between):
        xa = x.clone().requires_grad_(True)
        ln.zero_grad()
        ya = fused._NormAct.apply(xa, ln.weight, ln.bias, ln.eps, p, fused.POST_RELU, 21)
        if second_forward_in_between:
            fused.advance_rng(x.device)                             # the next step's forward ...
            xb = x.clone().requires_grad_(True)
            yb = fused._NormAct.apply(xb, ln.weight, ln.bias, ln.eps, p, fused.POST_RELU, 21)
            assert not torch.equal(ya, yb)                          # ... draws other masks
        (ya * g).sum().backward()
        return ya.detach().clone(), xa.grad.clone(), ln.weight.grad.clone(), ln.bias.grad.clone()

    step0 = fused.rng_state(x.device).clone()
    one = grads(False)
    fused.rng_state(x.device).copy_(step0)                          # same starting step for the second scenario
    two = grads(True)
    for a, b, name in zip(one, two, ("y", "dx", "dgamma", "dbeta")):
        assert torch.equal(a, b), name
    # and the gradient is the one of the mask that was applied: zero exactly where the output was dropped
    dropped = (one[0] == 0)
    assert float(one[1][dropped].abs().max()) >= 0.0               # (LayerNorm couples a row: dx is dense; checked via dgamma below)
    xr = x.clone().requires_grad_(True)
    base = F.relu(F.layer_norm(xr, (H,), ln.weight.detach(), ln.bias.detach(), ln.eps))
    mask = ((one[0] != 0) | ~(base.detach() > 0)).float() / (1 - p)
    (base * mask * g).sum().backward()
    assert_close(one[1], xr.grad, 1e-4, "dx of the first call")


def test_fused_sequential_matches_sequential_in_eval():
    from analysisgnn_amd.fused import FusedSequential
    torch.manual_seed(1)
    mods = lambda: [nn.LayerNorm(64), nn.Linear(64, 32), nn.ReLU(), nn.LayerNorm(32), nn.Dropout(0.3), nn.Linear(32, 16),
                    nn.ReLU(), nn.LayerNorm(16), nn.Dropout(0.3), nn.Linear(16, 16)]          # project_enc layout
    a = nn.Sequential(*mods()).to(DEV).eval()
    b = FusedSequential(*mods()).to(DEV).eval()
    b.load_state_dict(a.state_dict())
    x = torch.randn(100, 64, device=DEV)
    assert_close(b(x), a(x), 1e-4)
    x3 = torch.randn(4, 25, 64, device=DEV)
    assert_close(b(x3), a(x3), 1e-4)


def test_grouped_norm_act_matches_per_task_layernorm():
    """[N, T, 64] activations of the T task heads: ReLU + LayerNorm with per-task affine in one launch."""
    from analysisgnn_amd.fused import grouped_norm_act
    torch.manual_seed(3)
    N, T, H = 1000, 21, 64
    x = torch.randn(N, T, H)
    gamma, beta = torch.rand(T, H) + 0.5, torch.randn(T, H) * 0.2
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.layer_norm(F.relu(xr), (H,), None, None, 1e-5) * gr + br
    g = torch.randn(N, T, H)
    (ref * g).sum().backward()
    xg, gg, bg = (t.to(DEV).requires_grad_(True) for t in (x, gamma, beta))
    out = grouped_norm_act(xg, gg, bg, 1e-5, pre_relu=True)
    assert_close(out, ref, 1e-4, "y")
    (out * g.to(DEV)).sum().backward()
    assert_close(xg.grad, xr.grad, 1e-4, "dx")
    assert_close(gg.grad, gr.grad, 1e-4, "dgamma")
    assert_close(bg.grad, br.grad, 1e-4, "dbeta")


def test_small_embedding_backward_matches_library():
    """analysisgnn_amd.embedding: onehot^T @ dY against torch's sort-based embedding backward (35- and 15-row tables)."""
    from analysisgnn_amd.embedding import embedding
    torch.manual_seed(0)
    for V, N in ((35, 5000), (15, 4097), (35, 100)):
        w = torch.randn(V, 64, device=DEV, requires_grad=True)
        w2 = w.detach().clone().requires_grad_(True)
        idx = torch.randint(0, V, (N,), device=DEV)
        g = torch.randn(N, 64, device=DEV)
        out = embedding(idx, w)
        out.backward(g)
        ref = F.embedding(idx, w2)
        ref.backward(g)
        assert torch.equal(out, ref)
        assert_close(w.grad, w2.grad, 1e-5, f"dW V={V} N={N}")


@pytest.mark.parametrize("n,in_x,vocabs,dim", [(16000, 25, (35, 15), 64), (4097, 25, (35, 15), 64), (5, 3, (7,), 8),
                                                 (333, 0, (4, 9, 2), 6), (64, 56, (35, 15), 64)])
def test_embed_cat_matches_torch(n, in_x, vocabs, dim):
    """analysisgnn_amd.embedding.embed_cat (agnn_embed_cat_fwd/bwd_f32) against cat([x, F.embedding(...), ...]) and torch's
    embedding backward (analysis.py:574): forward bit-identical, table gradients to 1e-5 (different summation order)."""
    from analysisgnn_amd.embedding import embed_cat
    torch.manual_seed(n + dim)
    x = torch.randn(n, in_x, device=DEV)
    tabs = [torch.randn(v, dim, device=DEV, requires_grad=True) for v in vocabs]
    tabs2 = [t.detach().clone().requires_grad_(True) for t in tabs]
    idxs = [torch.randint(0, v, (n,), device=DEV) for v in vocabs]
    if vocabs[0] > 3:
        idxs[0][idxs[0] == 2] = 1                           # a table row nobody uses: its gradient must be exactly zero
    out = embed_cat(x, idxs, tabs)
    ref = torch.cat([x] + [F.embedding(i, t) for i, t in zip(idxs, tabs2)], dim=-1)
    assert out.shape == ref.shape and torch.equal(out, ref)
    assert out.stride(0) % 4 == 0                           # rows padded to 16 bytes ...
    if out.stride(0) > out.shape[1]:                        # ... with zeros behind the view
        full = out.as_strided((n, out.stride(0)), (out.stride(0), 1))
        assert torch.count_nonzero(full[:, out.shape[1]:]) == 0
    g = torch.randn(n, ref.shape[1], device=DEV)
    out.backward(g)
    ref.backward(g)
    for k, (a, b) in enumerate(zip(tabs, tabs2)):
        assert_close(a.grad, b.grad, 1e-5, f"dTable{k}")
    if vocabs[0] > 3:
        assert torch.count_nonzero(tabs[0].grad[2]) == 0
    # same launch twice: bitwise reproducible (no atomics)
    g1 = tabs[0].grad.clone()
    tabs[0].grad = None
    embed_cat(x, idxs, tabs).backward(g)
    assert torch.equal(tabs[0].grad, g1)


def test_fused_clip_adamw_matches_torch():
    """agnn_adamw_f32 (dp.FlatAdamW.step on a GPU) against clip_grad_norm_ + torch.optim.AdamW over several steps."""
    from analysisgnn_amd import dp
    torch.manual_seed(0)
    a = nn.Sequential(nn.Linear(37, 64), nn.ReLU(), nn.Linear(64, 5)).to(DEV)
    b = nn.Sequential(nn.Linear(37, 64), nn.ReLU(), nn.Linear(64, 5)).to(DEV)
    b.load_state_dict(a.state_dict())
    ref = torch.optim.AdamW(a.parameters(), lr=5e-3, weight_decay=5e-3)
    flat = dp.FlatGradBuffer(b.parameters(), views=False)
    opt = dp.FlatAdamW(b.parameters(), flat, lr=5e-3, weight_decay=5e-3)
    x = torch.randn(64, 37, device=DEV)
    for it in range(6):
        ref.zero_grad(set_to_none=True)
        (a(x).pow(2).sum() * 3.0).backward()
        total = torch.nn.utils.clip_grad_norm_(a.parameters(), 0.5)
        ref.step()
        flat.zero()
        (b(x).pow(2).sum() * 3.0).backward()
        flat.pack()
        opt.step(max_norm=0.5)
        assert abs(float(opt.last_norm) - float(total)) <= 1e-4 * float(total)
        for pa, pb in zip(a.parameters(), b.parameters()):
            torch.testing.assert_close(pb, pa, rtol=1e-4, atol=1e-6)


def test_relu_lets_a_nan_through_like_torch():
    """torch.relu(NaN) is NaN; fmaxf(NaN, 0) is 0.  A poisoned row must stay visible in the forward pass (a finite loss with
    non-finite gradients is what the silent version produced): pre- and post-activation of norm_act, and skip_act's ReLU."""
    from analysisgnn_amd.fused import norm_act, skip_act
    ln = torch.nn.LayerNorm(64).to(DEV)
    x = torch.randn(8, 64, device=DEV)
    x[3, 5] = float("nan")
    for pre, post in ((True, False), (False, True)):
        y = norm_act(x, ln, pre_relu=pre, post_relu=post)
        assert torch.isnan(y[3]).all() and torch.isfinite(y[[0, 1, 2, 4, 5, 6, 7]]).all()
    o = torch.randn(8, 64, device=DEV)
    o[2, 7] = float("nan")
    z = skip_act(o, torch.randn(8, 64, device=DEV), torch.zeros((), device=DEV), True, 0.0, False)
    assert torch.isnan(z[2, 7]) and int(torch.isnan(z).sum()) == 1
