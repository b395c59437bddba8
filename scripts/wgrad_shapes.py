#!/usr/bin/env python3
"""agnn_wgrad_f32 (k_wgrad + slab reduction) per shape of the C2 step: microseconds per call (HIP events around 20 back-to-back
calls), TFLOP/s, and the library GEMM for comparison.  AGNN_WGRAD_WGS overrides the slice-count target (csrc/wgrad.hip)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from analysisgnn_amd import linear
dev = "cuda:0"
shapes = [(16165, 256, 1280, 1280), (16000, 256, 1024, 1024), (16000, 256, 256, 256), (16000, 256, 256, 1280), (16000, 1344, 256, 256),
          (16000, 768, 256, 256), (16000, 384, 128, 256), (16000, 256, 512, 512), (16000, 256, 154, 156)]


def timed(fn, rep=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep * 1e3


for n, o, i, ld in shapes:
    dy = torch.randn(n, o, device=dev)
    xb = torch.randn(n, ld, device=dev)
    x = xb[:, :i]
    linear.ENABLED = True
    us = timed(lambda: linear.weight_grad(dy, x, True))
    linear.ENABLED = False
    us_lib = timed(lambda: linear.weight_grad(dy, x, True))
    linear.ENABLED = True
    fl = 2.0 * n * o * i
    print(f"n={n} out={o} in={i} ld_x={ld}: kernel {us:6.1f} us = {fl / us / 1e6:6.1f} TFLOP/s   library {us_lib:6.1f} us = {fl / us_lib / 1e6:6.1f} TFLOP/s")
