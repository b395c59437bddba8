#!/usr/bin/env python3
"""How does ROCm replay a FORKED hipGraph?  Two chains captured on two streams: A = n_a spin kernels of t_a us each,
B = one spin kernel of t_b us, joined at the end.  True concurrency: max(n_a * t_a, t_b).  Level-synchronous replay
(node k of every chain starts when all nodes k-1 have finished): t_b + (n_a - 1) * t_a.  Also prints the host time of the
replay call and the same graph captured on ONE stream (sum of all kernels), and what ONE memcpy / memset node in chain A
does to both (the fork is captured before chain A: `side.wait_stream(main)` depends on everything captured so far)."""
import sys, time
import torch

dev = torch.device("cuda:0")
CPU_US = 2100.0          # spin cycles per microsecond (shader clock ~2.1 GHz); calibrated below


def spin(us):
    torch.cuda._sleep(int(us * CPU_US))


SRC = None


def build(n_a, t_a, t_b, forked, b_first, extra=None):
    global SRC
    if SRC is None:
        SRC = (torch.randn(1 << 20, device=dev), torch.empty(1 << 20, device=dev))
    main = torch.cuda.Stream(device=dev)
    side = torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
        with torch.cuda.graph(g, stream=main):
            def chain_a():
                for i in range(n_a):
                    spin(t_a)
                    if extra == "memcpy" and i == n_a // 2:
                        SRC[1].copy_(SRC[0])                     # contiguous D2D copy: a memcpy node
                    if extra == "memset" and i == n_a // 2:
                        torch.cuda.current_stream().synchronize if False else None
                        SRC[1].view(torch.uint8).zero_()
                    if extra == "kernelcopy" and i == n_a // 2:
                        torch.add(SRC[0], 0.0, out=SRC[1])       # the same bytes moved by a kernel node
            def chain_b(fork_event=None):
                if forked:
                    if fork_event is not None:
                        side.wait_event(fork_event)              # depends on the fork point only (what autograd records)
                    else:
                        side.wait_stream(main)                   # depends on everything captured on main so far
                    with torch.cuda.stream(side):
                        spin(t_b)
                else:
                    spin(t_b)
            if b_first == "early_event":                         # B captured AFTER chain A, but waiting only for the fork
                ev = torch.cuda.Event()
                ev.record(main)
                chain_a(); chain_b(ev)
            elif b_first:
                chain_b(); chain_a()
            else:
                chain_a(); chain_b()
            if forked:
                main.wait_stream(side)
            spin(1)
    return g, main


def timed(g, main, reps=10):
    ts, hs = [], []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(main):
            e0.record()
            h0 = time.perf_counter()
            g.replay()
            h1 = time.perf_counter()
            e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
        hs.append((h1 - h0) * 1e6)
    ts.sort(); hs.sort()
    return ts[len(ts) // 2], hs[len(hs) // 2]


def main():
    global CPU_US
    torch.cuda._sleep(1000)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(int(1000 * CPU_US)); e1.record(); e1.synchronize()
    CPU_US *= 1000.0 / (e0.elapsed_time(e1) * 1e3)
    for n_a, t_a, t_b in ((10, 50, 500), (20, 25, 500)):
        for forked, b_first in ((False, False), (True, True), (True, "early_event")):
            for extra in (None, "memcpy"):
                g, m = build(n_a, t_a, t_b, forked, b_first, extra)
                gpu, host = timed(g, m)
                tag = "one stream" if not forked else ("forked, B captured first" if b_first is True else "forked, B captured LAST, waits for the fork event")
                print(f"A = {n_a} x {t_a} us, B = {t_b} us | {tag} | extra node {str(extra):10s} | GPU {gpu:7.1f} us  host {host:7.1f} us | "
                      f"concurrent would be {max(n_a * t_a, t_b)}, serial {n_a * t_a + t_b}")


if __name__ == "__main__":
    main()
