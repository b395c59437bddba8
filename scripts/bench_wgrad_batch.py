"""The main flush point's batch of weight-gradient products at C2 (3 GNN layers 256 x 1280, the heads' 1344 x 128, five 256 x 256
/ 256 x 512 projections) in one agnn_wgrad_batch_f32 call: time and check against the library.  usage: python scripts/bench_wgrad_batch.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from analysisgnn_amd import linear
dev = torch.device("cuda:0")
torch.manual_seed(0)
shapes = [(16335, 256, 1280)] * 3 + [(16000, 1344, 128), (16000, 256, 512), (16000, 128, 256), (16000, 256, 256), (16000, 256, 256), (16000, 128, 128)]
items, refs = [], []
for n, o, i in shapes:
    dy, x = torch.randn(n, o, device=dev), torch.randn(n, i, device=dev)
    dw, db = torch.empty(o, i, device=dev), torch.empty(o, device=dev)
    items.append(linear.WgItem(dy, x, True, dw, db))
linear.weight_grad_batch(items)
torch.cuda.synchronize()
err = max(float((it.dw_out - it.dy.t() @ it.x).abs().max() / (it.dy.t() @ it.x).abs().max()) for it in items)
errb = max(float((it.db_out - it.dy.sum(0)).abs().max()) for it in items)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    linear.weight_grad_batch(items)
e0.record()
for _ in range(10):
    linear.weight_grad_batch(items)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 10 * 1e3
fl = sum(2.0 * n * o * i for n, o, i in shapes)
print(f"batch of {len(shapes)}: {us:.1f} us incl. slab reduction, {fl / us / 1e6:.1f} TFLOP/s; max rel err dW {err:.2e}, max abs err db {errb:.2e}")
