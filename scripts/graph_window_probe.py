#!/usr/bin/env python3
"""Does a replayed hipGraph let a node start only when the node captured ~W positions before it has finished?
main stream: A, then m1..mK (short element-wise kernels, ~16 us); side stream (forks after A): one LONG spin kernel L
(torch.cuda._sleep), then s1..s3.  The side chain is captured before the m's (`first`), after them (`last`) or after the
first P of them (`mid`).  rocprofv3 --kernel-trace + graph_window_probe_report.py give when L and each m start.
usage: graph_window_probe.py first|last|mid [K] [P] [J]"""
import sys

import torch

where = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
P = int(sys.argv[3]) if len(sys.argv) > 3 else 10
J = int(sys.argv[4]) if len(sys.argv) > 4 else 3         # short kernels behind the long one on the side chain
dev = torch.device("cuda:0")
a = torch.randn(2048, 2048, device=dev)
b = torch.randn(2048, 2048, device=dev)
c = torch.empty(2048, 2048, device=dev)
xm = torch.randn(12_000_000, device=dev)
xs = torch.randn(5_000_000, device=dev)
side = torch.cuda.Stream(device=dev)
SPIN = 700_000          # cycles of torch.cuda._sleep: ~0.3 ms


def side_chain(ev):
    side.wait_event(ev)
    with torch.cuda.stream(side):
        torch.cuda._sleep(SPIN)
        for _ in range(J):
            xs.cos_()


def body():
    main = torch.cuda.current_stream(dev)
    torch.mm(a, b, out=c)
    ev = torch.cuda.Event()
    ev.record(main)
    if where == "first":
        side_chain(ev)
    for i in range(K):
        if where == "mid" and i == P:
            side_chain(ev)
        xm.sin_()
    if where == "last":
        side_chain(ev)
    main.wait_stream(side)
    xm.add_(1.0)


w = torch.cuda.Stream(device=dev)
with torch.cuda.stream(w):
    body()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
for _ in range(8):
    g.replay()
torch.cuda.synchronize()
print("done", where, K, P)
