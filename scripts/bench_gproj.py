#!/usr/bin/env python3
"""The grouped head projection (21 task heads, C2 shapes) forward + backward, 6 rounds — run under rocprofv3
(--kernel-trace --stats, or --pmc ...) to look at k_gproj_fwd / _dx / _dw in isolation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from analysisgnn_amd.heads import grouped_projection
dev = "cuda:0"
classes = list(bench.TASK_DICT.values())
offs = [0]
for c in classes:
    offs.append(offs[-1] + c)
N, K = 16000, 64
a = torch.randn(N, len(classes) * K, device=dev, requires_grad=True)
w = (torch.randn(offs[-1], K, device=dev) * 0.1).requires_grad_(True)
b = torch.zeros(offs[-1], device=dev, requires_grad=True)
g = torch.randn(N, offs[-1], device=dev)
for _ in range(6):
    out = grouped_projection(a, w, b, offs, K)
    out.backward(g)
torch.cuda.synchronize()
print("ok", float(out.abs().mean()))
