#!/usr/bin/env python3
"""The persistent GRU kernels alone at the C2 shape (B = 32 sequences x T = 500 steps x 2 directions, hidden 128): forward
and backward launch time (HIP events around 10 launches each), microseconds per time step, and a float64 check of one
layer (forward y, backward dx and weight gradients through analysisgnn_amd.gru)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from analysisgnn_amd import _lib
from analysisgnn_amd.gru import gru_forward
dev = torch.device("cuda:0")
B, T, I, Hh = int(os.environ.get("B", 32)), int(os.environ.get("T", 500)), 256, 128
torch.manual_seed(0)
rnn = torch.nn.GRU(I, Hh, num_layers=1, batch_first=True, bidirectional=True).to(dev)
x = torch.randn(B, T, I, device=dev, requires_grad=True)
y = gru_forward(rnn, x, True)
g = torch.randn_like(y)
y.backward(g)
ref = torch.nn.GRU(I, Hh, num_layers=1, batch_first=True, bidirectional=True).double()
ref.load_state_dict({k: v.detach().cpu().double() for k, v in rnn.state_dict().items()})
x64 = x.detach().cpu().double().requires_grad_(True)
y64 = ref(x64)[0]
y64.backward(g.cpu().double())
rel = lambda a, b: float((a.detach().cpu().double() - b).abs().max() / b.abs().max())
print(f"y max-rel {rel(y, y64.detach()):.2e}  dx {rel(x.grad, x64.grad):.2e}  " + "  ".join(f"d{k} {rel(p.grad, q.grad):.2e}" for (k, p), q in zip(rnn.named_parameters(), ref.parameters())))
lib = _lib.load()
gi = torch.randn(B, T, 2, 3 * Hh, device=dev)
w = torch.randn(2, 3 * Hh, Hh, device=dev) * 0.05
bh = torch.randn(2, 3 * Hh, device=dev) * 0.05
yy = torch.empty(B, T, 2 * Hh, device=dev)
saved = torch.empty(B, T, 2, 4, Hh, device=dev)
dgi = torch.empty(B, T, 2, 3 * Hh, device=dev)
dgh = torch.empty_like(dgi)
st = _lib.stream_ptr(dev)
def timed(fn, rep=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep * 1e3
f = timed(lambda: _lib.check(lib.agnn_gru_fwd_f32(gi.data_ptr(), w.data_ptr(), bh.data_ptr(), B, T, Hh, yy.data_ptr(), saved.data_ptr(), None, None, st), "fwd"))
dyv = torch.randn_like(yy)
bw = timed(lambda: _lib.check(lib.agnn_gru_bwd_f32(dyv.data_ptr(), yy.data_ptr(), saved.data_ptr(), w.data_ptr(), B, T, Hh, dgi.data_ptr(), dgh.data_ptr(), None, None, st), "bwd"))
print(f"B={B} T={T}: k_gru_fwd {f:.1f} us = {f / T:.3f} us/step ({f / T * 2400:.0f} cycles @2.4 GHz); k_gru_bwd {bw:.1f} us = {bw / T:.3f} us/step")
hp = torch.empty(B, T, 2, Hh, device=dev)
drop = torch.ones(B, T, 2 * Hh, device=dev)
y2 = torch.empty_like(yy)
f2 = timed(lambda: _lib.check(lib.agnn_gru_fwd_f32(gi.data_ptr(), w.data_ptr(), bh.data_ptr(), B, T, Hh, yy.data_ptr(), saved.data_ptr(), drop.data_ptr(), y2.data_ptr(), st), "fwd"))
b2 = timed(lambda: _lib.check(lib.agnn_gru_bwd_f32(dyv.data_ptr(), yy.data_ptr(), saved.data_ptr(), w.data_ptr(), B, T, Hh, dgi.data_ptr(), dgh.data_ptr(), None, hp.data_ptr(), st), "bwd"))
b3 = timed(lambda: _lib.check(lib.agnn_gru_bwd_f32(dyv.data_ptr(), yy.data_ptr(), saved.data_ptr(), w.data_ptr(), B, T, Hh, dgi.data_ptr(), dgh.data_ptr(), drop.data_ptr(), hp.data_ptr(), st), "bwd"))
print(f"as in the training step: k_gru_fwd with dropout scale {f2:.1f} us; k_gru_bwd with hprev output {b2:.1f} us, + dropout scale {b3:.1f} us")
