#!/usr/bin/env python3
"""agnn_gemm_nt_f32 (hand-written fp32 MFMA) against the library (torch.mm -> hipBLASLt) on the step's projection shapes:
correctness vs float64 and time per call (HIP events around hipGraph replays of 10 back-to-back calls)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from analysisgnn_amd import _lib
dev = torch.device("cuda:0")
lib = _lib.load()
def timed(fn, rep=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(rep): fn()
    ts = []
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / rep * 1e3)
    return sorted(ts)[len(ts) // 2]
shapes = [(16335, 256, 1280, "SAGE layer, c2s (R + 1 blocks)"), (16000, 256, 1280, "SAGE layer, c2"), (16335, 1280, 256, "its input gradient (W^T as w)"),
          (16335, 256, 256, "layer without edges / MLPs"), (16000, 256, 512, "cat_proj / project_enc.1"), (16000, 768, 256, "GRU input projections"),
          (16000, 128, 256, "project_enc.5"), (16000, 128, 128, "project_enc.9"), (16000, 1280, 128, "heads first layer (1344 -> 1280 here)"),
          (16000, 512, 2048, "C5 SAGE layer"), (16000, 2048, 512, "C5 input gradient")]
print(f"{'M':>6} {'N':>5} {'K':>5}  {'hand us':>8} {'TF':>6}  {'lib us':>8} {'TF':>6}  max-rel-err  what")
for M, N, K, what in shapes:
    torch.manual_seed(0)
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    c = torch.empty(M, N, device=dev)
    run = lambda: _lib.check(lib.agnn_gemm_nt_f32(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), b.data_ptr(), M, N, K, c.data_ptr(), c.stride(0), _lib.stream_ptr(dev)), "gemm")
    run(); torch.cuda.synchronize()
    ref = (a[:2048].double() @ w.double().t() + b.double())
    err = float((c[:2048].double() - ref).abs().max() / ref.abs().max())
    refl = (a[-300:].double() @ w.double().t() + b.double())
    err = max(err, float((c[-300:].double() - refl).abs().max() / refl.abs().max()))
    th = timed(run)
    out = torch.empty(M, N, device=dev)
    tl = timed(lambda: torch.addmm(b, a, w.t(), out=out))
    fl = 2.0 * M * N * K
    print(f"{M:6d} {N:5d} {K:5d}  {th:8.1f} {fl / th / 1e6:6.1f}  {tl:8.1f} {fl / tl / 1e6:6.1f}  {err:.1e}  {what}")

print("input-gradient products dX = dY W (agnn_gemm_nn_f32, the weight [out, in] as it lies) against torch.mm:")
for M, O, I, what in [(16335, 256, 1280, "SAGE layer dX"), (16000, 256, 256, "MLP dX"), (16000, 256, 512, "cat_proj dX"), (16000, 1344, 128, "heads first layer dX"),
                      (16000, 512, 2048, "C5 SAGE layer dX")]:
    torch.manual_seed(0)
    dy = torch.randn(M, O, device=dev); w = torch.randn(O, I, device=dev) * 0.05
    c = torch.empty(M, I, device=dev)
    run = lambda: _lib.check(lib.agnn_gemm_nn_f32(dy.data_ptr(), dy.stride(0), w.data_ptr(), w.stride(0), None, M, I, O, c.data_ptr(), c.stride(0), _lib.stream_ptr(dev)), "gemm_nn")
    run(); torch.cuda.synchronize()
    ref = dy[:2048].double() @ w.double()
    err = float((c[:2048].double() - ref).abs().max() / ref.abs().max())
    th = timed(run)
    out = torch.empty(M, I, device=dev)
    tl = timed(lambda: torch.mm(dy, w, out=out))
    fl = 2.0 * M * O * I
    print(f"{M:6d} {I:5d} {O:5d}  {th:8.1f} {fl / th / 1e6:6.1f}  {tl:8.1f} {fl / tl / 1e6:6.1f}  {err:.1e}  {what}")
