#!/usr/bin/env python3
"""Forked hipGraph: when does the branch that is captured SECOND start?  After a fork point the main stream captures N
element-wise kernels (~8 us each on 32 MB), then the side stream captures one long kernel chain (3 x sin on 256 MB).
Prints, from HIP events recorded inside the graph, the start delay of the side branch after the fork.  argv: N."""
import sys, torch
dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
small = torch.randn(8 * 1024 * 1024, device=dev)
big = torch.randn(64 * 1024 * 1024, device=dev)
side = torch.cuda.Stream()
ev_fork = torch.cuda.Event(enable_timing=True)
ev_side0 = torch.cuda.Event(enable_timing=True)
ev_main_end = torch.cuda.Event(enable_timing=True)
ev_side_end = torch.cuda.Event(enable_timing=True)
def work(record):
    x = small
    for _ in range(5):
        x = x * 1.0001
    main = torch.cuda.current_stream()
    if record: ev_fork.record(main)
    side.wait_stream(main)
    for _ in range(N):                      # main branch first in capture order
        x = x * 1.0001
    if record: ev_main_end.record(main)
    with torch.cuda.stream(side):           # side branch second
        if record: ev_side0.record(side)
        y = torch.sin(big); y = torch.sin(y); y = torch.sin(y)
        if record: ev_side_end.record(side)
    main.wait_stream(side)
    return x, y
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): work(False)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
# events cannot be timed inside a captured graph portably: time eagerly AND via kernel trace; here: eager reference
work(True); torch.cuda.synchronize()
print(f"eager : side starts {ev_fork.elapsed_time(ev_side0)*1e3:.0f} us after the fork, main branch ends at {ev_fork.elapsed_time(ev_main_end)*1e3:.0f} us, side ends at {ev_fork.elapsed_time(ev_side_end)*1e3:.0f} us")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = work(False)
torch.cuda.synchronize()
import time
for _ in range(3): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"graph : N={N} replay host {1e6*(t1-t0):.0f} us, total {1e6*(t2-t0):.0f} us")
