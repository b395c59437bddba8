import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, torch.nn as nn
from analysisgnn_amd.models import TorchAnalysisGNN
from analysisgnn_amd.synth import make_batch
import analysisgnn_amd.heads as H
tasks = {"cadence": 4, "localkey": 50, "tonkey": 50, "quality": 15, "romanNumeral": 185, "section": 2}
g = make_batch(1, 30)
torch.manual_seed(5)
o=128
m = TorchAnalysisGNN(g.metadata(), 25, 32, o, tasks, 2, dropout=0.0, use_jk=False, logit_fusion=True).train()
N=2500
x = torch.randn(N, o)
W1 = torch.cat([m.clf_dict[t][0].weight for t in tasks]).double(); b1 = torch.cat([m.clf_dict[t][0].bias for t in tasks]).double()
z64 = x.double() @ W1.t() + b1
print("min |z64|", float(z64.abs().min()), "count <1e-6", int((z64.abs()<1e-6).sum()), "<3e-6", int((z64.abs()<3e-6).sum()))
m = m.to("cuda:0"); xg = x.cuda()
mods=[m.clf_dict[t] for t in tasks]
for fused in (True, False):
    H.HEADS_FUSED = fused
    W1g = torch.cat([mm[0].weight for mm in mods]); b1g = torch.cat([mm[0].bias for mm in mods])
    if fused:
        gamma = torch.stack([mm[2].weight for mm in mods]); beta = torch.stack([mm[2].bias for mm in mods])
        W2 = torch.cat([mm[3].weight for mm in mods]); b2 = torch.cat([mm[3].bias for mm in mods])
        offs=[0]
        for mm in mods: offs.append(offs[-1]+mm[3].out_features)
        z = H.heads_forward(xg, W1g, b1g, gamma, beta, 1e-5, W2, b2, offs, 64)[0]
    else:
        z = torch.addmm(b1g, xg, W1g.t())
    zc = z.detach().cpu().double()
    flips = ((zc>0) != (z64>0))
    print("fused", fused, "max |z - z64|", float((zc-z64).abs().max()), "sign flips", int(flips.sum()), [float(v) for v in z64[flips][:5]])
