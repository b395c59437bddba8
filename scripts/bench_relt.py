#!/usr/bin/env python3
"""Timing of the HGT relation-transform kernels (agnn_relt_fwd / _bwd / _dw) at the C3 shape: N = 16 000 source rows,
6 relations, 4 heads, D = 64, K and V in one launch = 6.3 GFLOP per call.  HIP events around hipGraph replays of 10
back-to-back launches (as bench.py times its roofline kernel).  usage: bench_relt.py [N] [R] [heads]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from analysisgnn_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 6
heads = int(sys.argv[3]) if len(sys.argv) > 3 else 4
D, H = 64, heads * 64
dev = torch.device("cuda:0")
lib = _lib.load()
kqv = torch.randn(N, 3 * H, device=dev)
k, v = kqv[:, :H], kqv[:, 2 * H:]
Wk, Wv = torch.randn(R * heads, D, D, device=dev) * 0.1, torch.randn(R * heads, D, D, device=dev) * 0.1
yk, yv = torch.empty(N, R * H, device=dev), torch.empty(N, R * H, device=dev)
dk, dv = torch.empty(N, H, device=dev), torch.empty(N, H, device=dev)
dWk, dWv = torch.empty_like(Wk), torch.empty_like(Wv)
nws = int(lib.agnn_relt_dw_workspace_bytes(2, R, heads, D, N))
ws = torch.empty(nws, dtype=torch.uint8, device=dev)


def items(triples):
    arr = (_lib.ReltItem * len(triples))()
    for it, (x, w, y, ldx, ldy) in zip(arr, triples):
        it.x, it.w, it.y, it.ld_x, it.ld_y = x.data_ptr(), w.data_ptr(), y.data_ptr(), ldx, ldy
    return arr


I_f = items([(k, Wk, yk, k.stride(0), yk.stride(0)), (v, Wv, yv, v.stride(0), yv.stride(0))])
I_b = items([(yk, Wk, dk, yk.stride(0), dk.stride(0)), (yv, Wv, dv, yv.stride(0), dv.stride(0))])
I_w = items([(k, yk, dWk, k.stride(0), yk.stride(0)), (v, yv, dWv, v.stride(0), yv.stride(0))])
st = lambda: _lib.stream_ptr(dev)
fns = {"fwd": lambda: _lib.check(lib.agnn_relt_fwd_f32(2, I_f, R, heads, D, N, st()), "f"),
       "bwd": lambda: _lib.check(lib.agnn_relt_bwd_f32(2, I_b, R, heads, D, N, st()), "b"),
       "dw": lambda: _lib.check(lib.agnn_relt_dw_f32(2, I_w, R, heads, D, N, ws.data_ptr(), nws, st()), "w")}
flops = 2.0 * N * D * D * R * heads * 2
for name, fn in fns.items():
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    ts = []
    for _ in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 100)
    med = statistics.median(ts[2:])
    print(f"{name}: {med:7.1f} us per launch  {flops / med / 1e6:6.1f} TFLOP/s ({flops / med / 1e6 / 157.3 * 100:.0f}% of the fp32-MFMA peak)  N={N} R={R} heads={heads}")
