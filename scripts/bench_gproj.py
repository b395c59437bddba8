#!/usr/bin/env python3
"""The grouped head projection (21 task heads, C2 shapes): forward / input gradient / weight gradient timed on their own
(HIP events around 20 back-to-back launches each) and checked against float64; also the thing to run under rocprofv3
(--kernel-trace --stats, or --pmc ...)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from analysisgnn_amd import _lib
from analysisgnn_amd.heads import grouped_projection
dev = "cuda:0"
classes = list(bench.TASK_DICT.values())
offs = [0]
for c in classes:
    offs.append(offs[-1] + c)
N, K = int(os.environ.get("N", "16000")), 64
torch.manual_seed(0)
a = torch.randn(N, len(classes) * K, device=dev, requires_grad=True)
w = (torch.randn(offs[-1], K, device=dev) * 0.1).requires_grad_(True)
b = torch.randn(offs[-1], device=dev, requires_grad=True)
g = torch.randn(N, offs[-1], device=dev)
for _ in range(3):
    out = grouped_projection(a, w, b, offs, K)
    out.backward(g)
ref = torch.cat([a.detach().double()[:, t * K:(t + 1) * K] @ w.detach().double()[offs[t]:offs[t + 1]].t() + b.detach().double()[offs[t]:offs[t + 1]]
                 for t in range(len(classes))], dim=1)
err = float((out.detach().double() - ref).abs().max())
print(f"forward max |err| vs float64: {err:.2e}")
assert err < 1e-4


def timed(fn, rep=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep * 1e3


with torch.no_grad():
    us = timed(lambda: grouped_projection(a, w, b, offs, K))
alg = 4 * (a.numel() + N * offs[-1])
print(f"forward: {us:.1f} us per launch (incl. ~3 us of launch gap) = {alg / us / 1e6:.2f} TB/s of {alg / 1e6:.1f} MB algorithmic")
out = grouped_projection(a, w, b, offs, K)
us_b = timed(lambda: out.backward(g, retain_graph=True))
print(f"backward (dx + dw + slab reduce): {us_b:.1f} us")
