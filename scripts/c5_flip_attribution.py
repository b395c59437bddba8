"""One-off attribution study for tests/test_gpu_c5.py (profiles/r03_parity_notes.md): is the HIP path's gradient deviation
from the float64 oracle, where it exceeds 1e-4, the footprint of flipped ReLU derivatives and nothing else?

For every ReLU input of the float64 run that lies within 3e-6 of its call's largest magnitude (element k of call c), flipping
its derivative changes dL/dtheta by EXACTLY  delta_k * d pre_c[k] / d theta,  delta_k = (1 - 2 m_k) dL/d post_c[k]  (the forward
value does not move: relu(pre) ~ 0 either way).  The script computes those K directions (K backward passes through the
float64 graph), then fits  g_hip - g_64 ~ sum_k b_k direction_k  by least squares over ALL parameters at once and prints the
fitted b_k (expected: 0 = not flipped or 1 = flipped) and the residual left unexplained, against plain fp32 rounding.
usage: python scripts/c5_flip_attribution.py <seed>"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import torch.nn.functional as F
from analysisgnn_amd.models import TorchAnalysisGNN
from analysisgnn_amd.synth import make_sampled_batch, torch_inputs
from oracle import encoders_ref as E
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
DEV = "cuda:0"
C5_TASKS = {"cadence": 4, "localkey": 50, "romanNumeral": 185}
g = make_sampled_batch(4, 500, (5, 5, 5), first_seed=40)
H, L = 512, 4
torch.manual_seed(seed)
m = TorchAnalysisGNN(g.metadata(), in_channels=25, hidden_channels=H, out_channels=128, task_dict=C5_TASKS, num_layers=L, dropout=0.0,
                     use_jk=False, logit_fusion=False, encoder_type="metricalgnn").train()
P = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
m = m.to(DEV)
I = torch_inputs(g, in_channels=25, seed=5)
names = [k for k, v in m.named_parameters()]
P64 = {k: (v.double().requires_grad_(True) if v.is_floating_point() else v) for k, v in P.items()}
pre, post = [], []
orig = F.relu
def relu(x, *a, **k):
    y = orig(x, *a, **k); y.retain_grad(); pre.append(x); post.append(y); return y
F.relu = relu
try:
    x = E.analysis_encode(P64, "metricalgnn", g.metadata(), L, I["pitch_spelling"], I["key_signature"], {k: v.double() for k, v in I["x_dict"].items()},
                          I["edge_index_dict"], I["batch_dict"], I["batch_size"], I["neighbor_mask_node"], I["neighbor_mask_edge"])
    ref = E.analysis_logits(P64, x, list(C5_TASKS))
finally:
    F.relu = orig
loss = sum((v ** 2).mean() for v in ref.values())
plist = [P64[k] for k in names]
g64 = torch.autograd.grad(loss, plist, retain_graph=True, allow_unused=True)
gpost = torch.autograd.grad(loss, post, retain_graph=True, allow_unused=True)
used = [i for i, t in enumerate(g64) if t is not None]
flat64 = torch.cat([g64[i].reshape(-1) for i in used])
risk = []
for c, xin in enumerate(pre):
    idx = (xin.detach().abs() < 3e-6 * xin.detach().abs().max()).reshape(-1).nonzero().reshape(-1)
    for i in idx.tolist():
        risk.append((c, i))
print(f"seed {seed}: {sum(t.numel() for t in pre)} ReLU inputs, {len(risk)} at risk", flush=True)
J = {k: ({kk: vv.to(DEV) for kk, vv in v.items()} if isinstance(v, dict) and v and isinstance(next(iter(v.values())), torch.Tensor)
         else (v.to(DEV) if isinstance(v, torch.Tensor) else v)) for k, v in I.items()}
out = m(J["pitch_spelling"], J["key_signature"], J["x_dict"], J["edge_index_dict"], J["batch_dict"], J["batch_size"], J["neighbor_mask_node"], J["neighbor_mask_edge"])
sum((v ** 2).mean() for v in out.values()).backward()
pg = dict(m.named_parameters())
flat_hip = torch.cat([pg[names[i]].grad.detach().cpu().double().reshape(-1) for i in used])
d = flat_hip - flat64
print(f"|g_hip - g64| / |g64| = {float(d.norm() / flat64.norm()):.3e} over all parameters", flush=True)
t0 = time.time()
dirs = []
for n, (c, i) in enumerate(risk):
    xin = pre[c]
    mk = 1.0 if float(xin.detach().reshape(-1)[i]) > 0 else 0.0
    delta = (1.0 - 2.0 * mk) * float(gpost[c].reshape(-1)[i]) if gpost[c] is not None else 0.0
    gd = torch.autograd.grad(xin.reshape(-1)[i], plist, retain_graph=True, allow_unused=True)
    dirs.append(delta * torch.cat([(gd[j] if gd[j] is not None else torch.zeros_like(plist[j])).reshape(-1) for j in used]))
    if n % 10 == 0:
        print(f"  direction {n + 1}/{len(risk)}  ({time.time() - t0:.0f} s)", flush=True)
A = torch.stack(dirs, dim=1)                       # [n_params, K]
keep = (A.norm(dim=0) > 0)
A2 = A[:, keep]
sol = torch.linalg.lstsq(A2, d.unsqueeze(1)).solution.squeeze(1)
res = d - A2 @ sol
print("fitted flip coefficients b_k (0 = as float64, 1 = flipped):", [round(float(v), 3) for v in sol])
print(f"unexplained residual |r| / |g64| = {float(res.norm() / flat64.norm()):.3e}   (before the fit: {float(d.norm() / flat64.norm()):.3e})")
off = 0
worst = []
for i in used:
    k = g64[i].numel()
    dd, rr, gg = d[off:off + k], res[off:off + k], g64[i].reshape(-1)
    worst.append((float(dd.abs().max() / max(1.0, float(gg.abs().max()))), float(rr.abs().max() / max(1.0, float(gg.abs().max()))), float(rr.norm() / gg.norm()), names[i]))
    off += k
worst.sort(reverse=True)
for w in worst[:8]:
    print(f"  {w[3]}: max err / max(1,|g|max) before {w[0]:.2e} -> after removing the fitted flips {w[1]:.2e}; residual Frobenius {w[2]:.2e}")
