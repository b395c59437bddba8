"""Per-tensor gradient error of the HIP path against a float64 oracle run on the r3 big wrapper case (diagnosis)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from helpers import r3_case
from test_gpu_r3_wrapper import _build, _to_dev
from oracle import encoders_ref as E
import torch.nn.functional as F
name = sys.argv[1] if len(sys.argv) > 1 else "r3_wrapper_c2_h256"
z, cfg, g, I, labels = r3_case(name)
m, clf = _build(z, cfg, g)
tasks = list(cfg["tasks"]); T = len(tasks)
P = {k: v.detach().cpu().double().requires_grad_(True) for k, v in m.state_dict().items() if v.is_floating_point()}
x = E.analysis_encode(P, cfg["enc"], g.metadata(), cfg["L"], I["pitch_spelling"], I["key_signature"], {k: v.double() for k, v in I["x_dict"].items()},
                      I["edge_index_dict"], I["batch_dict"], I["batch_size"], I["neighbor_mask_node"], I["neighbor_mask_edge"], use_jk=cfg["use_jk"])
lg = E.analysis_logits(P, x, tasks)
x.retain_grad()
for t in tasks: lg[t].retain_grad()
p = clf.params.detach().cpu().double()
tot = sum(0.5 / p[i] ** 2 * F.cross_entropy(lg[t], labels[i], ignore_index=-1, label_smoothing=0.1) + torch.log(1 + p[i] ** 2) for i, t in enumerate(tasks)) / T + 0.1 * x.pow(2).mean()
tot.backward()
print("oracle total", float(tot), "fixture", float(z["loss.total"]))
J = _to_dev(I)
xg = m.encode(J["pitch_spelling"], J["key_signature"], J["x_dict"], J["edge_index_dict"], J["batch_dict"], J["batch_size"], J["neighbor_mask_node"], J["neighbor_mask_edge"])
from analysisgnn_amd.heads import training_loss
cat, offs, _ = m.forward_clf_fused(xg)
xg.retain_grad(); cat.retain_grad()
total, per = training_loss(cat, offs, labels.to("cuda:0"), xg, 0.1, 0.1, -1, task_params=clf.weights())
total.backward(); torch.cuda.synchronize()
print("hip total", float(total))
print("x err", float((xg.detach().cpu().double() - x.detach()).abs().max()), "scale", float(x.abs().max()))
def rel(a, b): return float((a.detach().cpu().double() - b).abs().max() / b.abs().max()), float((a.detach().cpu().double() - b).norm() / b.norm())
print("d loss / d encoder output x: max-rel, fro-rel", rel(xg.grad, x.grad))
for i, t in enumerate(tasks):
    print(f"d loss / d logits[{t}] (C={offs[i+1]-offs[i]}): ", rel(cat.grad[:, offs[i]:offs[i + 1]], lg[t].grad), "logits fwd", rel(cat[:, offs[i]:offs[i + 1]], lg[t].detach()))
rows = []
for k, q in m.named_parameters():
    if P[k].grad is None or q.grad is None: continue
    a, b = q.grad.detach().cpu().double(), P[k].grad
    rows.append((float((a - b).abs().max() / b.abs().max()), float((a - b).norm() / b.norm()), float(b.abs().max()), k))
rows.sort(reverse=True)
for r in rows[:25]: print(f"max-rel {r[0]:.2e}  fro-rel {r[1]:.2e}  max|g| {r[2]:.2e}  {r[3]}")
