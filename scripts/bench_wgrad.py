#!/usr/bin/env python3
"""Kernel vs library timing of the weight-gradient GEMM shapes of the C2 step (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from analysisgnn_amd import linear
dev = "cuda:0"
linear.MAX_OUT_IN = int(os.environ.get("AGNN_WGRAD_MAX", linear.MAX_OUT_IN))
shapes = [(16000, 256, 256), (16000, 256, 1024), (16000, 768, 256), (16000, 384, 128), (16000, 1344, 128), (16000, 690, 1344),
          (16000, 128, 256), (16000, 256, 512), (16000, 128, 128)]
for n, o, i in shapes:
    dy = torch.randn(n, o, device=dev); x = torch.randn(n, i, device=dev)
    for rnd in range(6):
        linear.ENABLED = True
        a, b = linear.weight_grad(dy, x, True)
        linear.ENABLED = False
        c, d = linear.weight_grad(dy, x, True)
    torch.cuda.synchronize()
    print(n, o, i, float((a - c).abs().max()), float((b - d).abs().max()))
