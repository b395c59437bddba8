#!/usr/bin/env python3
"""Does a forked branch start when its dependency completes if the main chain keeps the GPU full?  Main chain: 24 big
GEMMs ([16000,1024]x[1024,256], ~75 us, every CU busy); after GEMM #3 a side stream is forked and runs 12 kernels of
kind argv[2] (sin: element-wise, gemm: [16000,256]x[256,256]).  argv[1] = graph | eager.  Run under rocprofv3
--kernel-trace; scripts/fork_probe_report.py prints when the side branch actually started."""
import sys, torch
dev = "cuda:0"
mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
kind = sys.argv[2] if len(sys.argv) > 2 else "sin"
a = torch.randn(16000, 1024, device=dev); w = torch.randn(1024, 1024, device=dev) * 0.03
b = torch.randn(16000, 256, device=dev); v = torch.randn(256, 256, device=dev) * 0.06
big = torch.randn(16 * 1024 * 1024 // 4, device=dev)
side = torch.cuda.Stream()
def work():
    x = a
    for i in range(24):
        x = x @ w
        if i == 3:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                y = big if kind == "sin" else b
                for _ in range(12):
                    y = torch.sin(y) if kind == "sin" else torch.tanh(y @ v)
    torch.cuda.current_stream().wait_stream(side)
    return x, y
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): work()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
if mode == "graph":
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = work()
    torch.cuda.synchronize()
    g.replay()
else:
    work()
torch.cuda.synchronize()
