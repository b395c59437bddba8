#!/usr/bin/env python3
"""How does a replayed hipGraph schedule two independent chains that fork from one node?  Stream M: A (long), m1..mK (short);
stream S forks after A: s1..sJ.  Captured M-chain-first or S-chain-first; the replay's kernel trace (rocprofv3) shows when
s1 and m1 start relative to the end of A.  Kernels are told apart by their element counts (grid sizes) in the trace.
usage: graph_fork_probe.py <order: m_first|s_first> [K] [J] [SZ]"""
import sys

import torch

order = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
J = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
SZ = int(sys.argv[4]) if len(sys.argv) > 4 else 8192      # A = SZ^3 matmul: 8192 -> ~7 ms, 2048 -> ~0.15 ms
a = torch.randn(SZ, SZ, device=dev)
b = torch.randn(SZ, SZ, device=dev)
c = torch.empty(SZ, SZ, device=dev)
xm = torch.randn(3_000_000, device=dev)      # "m" kernels: sin_ on 3.0 M elements
xs = torch.randn(5_000_000, device=dev)      # "s" kernels: cos_ on 5.0 M elements
side = torch.cuda.Stream(device=dev)


def m_chain():
    for _ in range(K):
        xm.sin_()


def s_chain(main):
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for _ in range(J):
            xs.cos_()


def body():
    main = torch.cuda.current_stream(dev)
    torch.mm(a, b, out=c)                     # A
    if order == "m_first":
        ev = torch.cuda.Event()
        ev.record(main)                       # the fork point is the end of A in both orders
        m_chain()
        side.wait_event(ev)
        with torch.cuda.stream(side):
            for _ in range(J):
                xs.cos_()
    else:
        s_chain(main)
        m_chain()
    main.wait_stream(side)
    xm.add_(1.0)                              # join


warm = torch.cuda.Stream(device=dev)
with torch.cuda.stream(warm):
    body()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
for _ in range(12):
    g.replay()
torch.cuda.synchronize()
print("done", order, K, J)
