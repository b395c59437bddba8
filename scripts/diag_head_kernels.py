"""Relative error (max / Frobenius, against float64) of the head-block kernels taken one by one (diagnosis)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, bench
import torch.nn.functional as F
from analysisgnn_amd.heads import grouped_projection
from analysisgnn_amd.fused import grouped_norm_act
dev = "cuda:0"
classes = list(bench.TASK_DICT.values()); G = len(classes); K = 64
offs = [0]
for c in classes: offs.append(offs[-1] + c)
def rel(a, b): return f"max-rel {float((a.detach().cpu().double() - b).abs().max() / b.abs().max()):.2e} fro-rel {float((a.detach().cpu().double() - b).norm() / b.norm()):.2e}"
for N in (120, 16000):
    torch.manual_seed(0)
    a = torch.randn(N, G * K); w = torch.randn(offs[-1], K) * 0.08; b = torch.randn(offs[-1]) * 0.08; g = torch.randn(N, offs[-1]) * 1e-3
    a64, w64, b64 = a.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = torch.cat([a64[:, t * K:(t + 1) * K] @ w64[offs[t]:offs[t + 1]].t() + b64[offs[t]:offs[t + 1]] for t in range(G)], dim=1)
    ref.backward(g.double())
    ad, wd, bd = a.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    out = grouped_projection(ad, wd, bd, offs, K)
    out.backward(g.to(dev))
    print(f"N={N} gproj fwd {rel(out, ref.detach())} | da {rel(ad.grad, a64.grad)} | dw {rel(wd.grad, w64.grad)} | db {rel(bd.grad, b64.grad)}")
    for t in (0, 1, 9, 11, 16):
        print(f"   da group {t} (C={classes[t]}): {rel(ad.grad[:, t * K:(t + 1) * K], a64.grad[:, t * K:(t + 1) * K])}")
    # segmented ReLU + LayerNorm
    x = torch.randn(N, G, K); gam = 1 + 0.08 * torch.randn(G, K); bet = 0.08 * torch.randn(G, K); gy = torch.randn(N, G, K) * 1e-3
    x64, g64, b64 = x.double().requires_grad_(True), gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    r = F.layer_norm(F.relu(x64), (K,), None, None, 1e-5) * g64 + b64
    r.backward(gy.double())
    xd, gd, bd2 = x.to(dev).requires_grad_(True), gam.to(dev).requires_grad_(True), bet.to(dev).requires_grad_(True)
    y = grouped_norm_act(xd, gd, bd2, 1e-5, pre_relu=True)
    y.backward(gy.to(dev))
    print(f"N={N} seg-LN fwd {rel(y, r.detach())} | dx {rel(xd.grad, x64.grad)} | dgamma {rel(gd.grad, g64.grad)} | dbeta {rel(bd2.grad, b64.grad)}")
