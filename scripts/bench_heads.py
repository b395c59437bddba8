"""Head block forward (and forward + backward) at the C2 shape: one launch (agnn_heads_fwd_f32) vs three (library GEMM, segmented
ReLU + LayerNorm, grouped projection).  usage: python scripts/bench_heads.py [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
import analysisgnn_amd.heads as H
from bench import TASK_DICT

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
dev = torch.device("cuda:0")
torch.manual_seed(0)
clf = nn.ModuleDict({t: nn.Sequential(nn.Linear(128, 64), nn.ReLU(), nn.LayerNorm(64), nn.Linear(64, c)) for t, c in TASK_DICT.items()}).to(dev)
x = torch.randn(N, 128, device=dev, requires_grad=True)
tasks = list(TASK_DICT)
gout = torch.randn(N, sum(TASK_DICT.values()), device=dev)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def fwd():
    with torch.no_grad():
        return H.fused_head_logits(clf, x, tasks)[0]


def fwd_bwd():
    for p in clf.parameters():
        p.grad = None
    x.grad = None
    logits, _ = H.fused_head_logits(clf, x, tasks)
    logits.backward(gout)


flops = 2 * N * (128 * 64 * len(tasks) + 64 * sum(TASK_DICT.values()))
for fused in (True, False, True, False):
    H.HEADS_FUSED = fused
    tf, tb = timed(fwd), timed(fwd_bwd)
    print(f"fused={fused}: forward {tf:7.1f} us ({flops / tf / 1e6:6.1f} TFLOP/s useful)   forward+backward {tb:7.1f} us")
