#!/usr/bin/env python3
"""Host cost of replaying a hipGraph of N short kernels on two streams, back to back: does a launch return before the
previous replay of the same graph has finished, and how long does the host spend per node?"""
import sys
import time

import torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
dev = torch.device("cuda:0")
x = torch.randn(2_000_000, device=dev)
y = torch.randn(2_000_000, device=dev)
side = torch.cuda.Stream(device=dev)


def body():
    main = torch.cuda.current_stream(dev)
    x.sin_()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for _ in range(N // 2):
            y.cos_()
    for _ in range(N // 2):
        x.sin_()
    main.wait_stream(side)
    x.add_(1.0)


w = torch.cuda.Stream(device=dev)
with torch.cuda.stream(w):
    body()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
host = []
for _ in range(20):
    h0 = time.perf_counter()
    g.replay()
    host.append(time.perf_counter() - h0)
e1.record()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"N={N}: host time per replay() call: median {sorted(host)[10] * 1e6:.0f} us (min {min(host) * 1e6:.0f}, max {max(host) * 1e6:.0f}); "
      f"20 calls returned after {(t1 - t0) * 1e3:.2f} ms, GPU finished after {(t2 - t0) * 1e3:.2f} ms; GPU time per replay {e0.elapsed_time(e1) / 20 * 1e3:.0f} us")
