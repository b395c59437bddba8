#!/usr/bin/env python3
"""Forked hipGraph with REAL kernels on the main chain: fork -> side chain {short, LONG (spin), short} and main chain
{n matmuls} -> join.  Does the main chain run beside the long side kernel, for both capture orders?  Prints the GPU time
of a replay against the two bounds (concurrent: max of the chains; serial: their sum)."""
import sys, time
import torch

dev = torch.device("cuda:0")
CYC = 2100.0


def spin(us):
    torch.cuda._sleep(int(us * CYC))


def build(order, n_main, long_us, a, b, cross_wait):
    main, side, third = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph()
    outs = []
    with torch.cuda.stream(main):
        with torch.cuda.graph(g, stream=main):
            outs.append(a @ b)
            ev = torch.cuda.Event(); ev.record(main)

            def side_chain():
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    outs.append(a @ b)
                    spin(long_us)
                    if cross_wait:                       # a third stream waits for the long kernel (like the weight-gradient stream)
                        third.wait_stream(side)
                        with torch.cuda.stream(third):
                            outs.append(a @ b)
                    outs.append(a @ b)

            def main_chain():
                for _ in range(n_main):
                    outs.append(a @ b)
            if order == "side_first":
                side_chain(); main_chain()
            else:
                main_chain(); side_chain()
            main.wait_stream(side)
            if cross_wait:
                main.wait_stream(third)
            outs.append(a @ b)
    return g, main, outs


def timed(g, main, reps=8):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(main):
            e0.record(); g.replay(); e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    global CYC
    torch.cuda._sleep(1000); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(int(1000 * CYC)); e1.record(); e1.synchronize()
    CYC *= 1000.0 / (e0.elapsed_time(e1) * 1e3)
    a = torch.randn(2048, 1024, device=dev); b = torch.randn(1024, 1024, device=dev)
    for _ in range(3):
        a @ b
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        a @ b
    e1.record(); e1.synchronize()
    mm = e0.elapsed_time(e1) * 1e3 / 20
    print(f"one matmul ~{mm:.1f} us")
    for n_main, long_us in ((20, 300), (40, 300)):
        for cross_wait in (False, True):
            for order in ("side_first", "main_first"):
                g, m, keep = build(order, n_main, long_us, a, b, cross_wait)
                t = timed(g, m)
                print(f"main chain {n_main} matmuls, side chain matmul + {long_us} us spin + matmul, third-stream wait {cross_wait!s:5} | captured {order:10} | "
                      f"GPU {t:7.1f} us | concurrent ~{max(n_main * mm, long_us + 2 * mm) + 2 * mm:.0f}, serial ~{(n_main + 4) * mm + long_us:.0f}")


if __name__ == "__main__":
    main()
