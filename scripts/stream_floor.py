#!/usr/bin/env python3
"""What does this GPU need for the SpMM's traffic pattern with trivial kernels?  (a) write-only 65.5 MB (fill),
(b) read 16.4 MB + write 65.5 MB (repeat the [N,H] matrix four times side by side) — 10 launches per hipGraph replay,
HIP events around the replay, like bench.py's roofline timing."""
import torch
dev = "cuda:0"
N, H, R = 16000, 256, 4
x = torch.randn(N, H, device=dev)
out = torch.empty(N, R * H, device=dev)
def timed(fn, rep=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(rep): fn()
    ts = []
    for _ in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / rep)
    ts.sort()
    return ts[len(ts) // 2]
t_fill = timed(lambda: out.fill_(1.0))
t_rep = timed(lambda: out.view(N, R, H).copy_(x.unsqueeze(1).expand(N, R, H)))
print(f"fill 65.5 MB: {t_fill:.1f} us = {65.536e6 / t_fill / 1e6:.2f} TB/s;  read 16.4 + write 65.5 MB: {t_rep:.1f} us = {81.92e6 / t_rep / 1e6:.2f} TB/s")
