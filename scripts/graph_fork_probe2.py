#!/usr/bin/env python3
"""Fork probe, second form: does a long-running small-grid kernel (the persistent GRU kernel: 64 workgroups, ~300 us) on the
FIRST-captured chain hold back the other chain of a fork in a replayed hipGraph?  main: A (GEMM), G (GRU layer), m1..m5;
side (forks after A): s1..s8.  order = main_first | side_first (capture order of the two chains after A).
Report with graph_fork_probe_report.py (kernels told apart by name: Cijk = A, k_gru_fwd = G, sin = m, cos = s)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from analysisgnn_amd.gru import _GRULayer  # noqa: E402

order = sys.argv[1]
KM = int(sys.argv[2]) if len(sys.argv) > 2 else 5       # short kernels behind the GRU kernel on the main chain
WITH_GRU = (sys.argv[3] if len(sys.argv) > 3 else "1") == "1"
dev = torch.device("cuda:0")
a = torch.randn(2048, 2048, device=dev)
b = torch.randn(2048, 2048, device=dev)
c = torch.empty(2048, 2048, device=dev)
xm = torch.randn(12_000_000, device=dev)      # ~20 us per sin_: the replay is never host-bound
xs = torch.randn(5_000_000, device=dev)
B, T, H = 32, 500, 128
gi_x = torch.randn(B, T, 256, device=dev)
w_ih = torch.randn(2, 3 * H, 256, device=dev) * 0.05
w_hh = torch.randn(2, 3 * H, H, device=dev) * 0.05
b_ih = torch.zeros(2, 3 * H, device=dev)
b_hh = torch.zeros(2, 3 * H, device=dev)
side = torch.cuda.Stream(device=dev)


def main_chain():
    if WITH_GRU:
        with torch.no_grad():
            _GRULayer.apply(gi_x, w_ih, w_hh, b_ih, b_hh)
    for _ in range(KM):
        xm.sin_()


def side_chain():
    with torch.cuda.stream(side):
        for _ in range(8):
            xs.cos_()


def body():
    main = torch.cuda.current_stream(dev)
    torch.mm(a, b, out=c)
    ev = torch.cuda.Event()
    ev.record(main)
    if order == "main_first":
        main_chain()
        side.wait_event(ev)
        side_chain()
    else:
        side.wait_event(ev)
        side_chain()
        main_chain()
    main.wait_stream(side)
    xm.add_(1.0)


w = torch.cuda.Stream(device=dev)
with torch.cuda.stream(w):
    body()
torch.cuda.synchronize()
if order == "two_graphs":
    # the two chains as separate graphs on two real streams, ordered by events between the launches
    def head():
        torch.mm(a, b, out=c)

    def side_only():
        for _ in range(8):
            xs.cos_()

    def tail():
        xm.add_(1.0)

    run = torch.cuda.Stream(device=dev)        # graphs are captured and replayed on non-default streams
    g_head, g_main, g_side, g_tail = (torch.cuda.CUDAGraph() for _ in range(4))
    with torch.cuda.stream(run):
        for gg, fn in ((g_head, head), (g_main, main_chain), (g_side, side_only), (g_tail, tail)):
            with torch.cuda.graph(gg, stream=run):
                fn()
    torch.cuda.synchronize()
    for _ in range(8):
        with torch.cuda.stream(run):
            g_head.replay()
            side.wait_stream(run)
            with torch.cuda.stream(side):
                g_side.replay()
            g_main.replay()
            run.wait_stream(side)
            g_tail.replay()
    torch.cuda.synchronize()
else:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    for _ in range(8):
        g.replay()
    torch.cuda.synchronize()
print("done", order)
