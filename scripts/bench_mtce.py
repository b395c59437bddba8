"""agnn_multitask_ce_f32 at the C2 shape ([16000, 658] logits, 21 tasks): kernel time.  usage: python scripts/bench_mtce.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from analysisgnn_amd.heads import multitask_cross_entropy
from bench import TASK_DICT
dev = torch.device("cuda:0")
N = 16000
cs = list(TASK_DICT.values())
offs = [0]
for c in cs:
    offs.append(offs[-1] + c)
logits = torch.randn(N, offs[-1], device=dev, requires_grad=True)
lab = torch.stack([torch.randint(0, c, (N,), device=dev) for c in cs])
def run():
    return multitask_cross_entropy(logits, offs, lab, 0.1, -1)
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
print(f"multitask CE forward (+ gradient image): {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call incl. the per-task reduction launch")
