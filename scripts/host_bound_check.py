#!/usr/bin/env python3
"""Is the bench step host-bound?  Time how long the host needs to ENQUEUE K steps vs. how long the GPU needs to finish them."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from analysisgnn_amd import dp, graph
from analysisgnn_amd.heads import multitask_cross_entropy
from analysisgnn_amd.models import TorchAnalysisGNN
from analysisgnn_amd.synth import make_batch, torch_inputs
dev = torch.device("cuda:0")
g = make_batch(bench.N_SUB, bench.N_NOTES)
I = torch_inputs(g, bench.IN_CH, dev, 0)
labels = bench.make_labels(I["batch_size"], dev, 1)
label_mat = torch.stack([labels[t] for t in bench.TASK_DICT])
torch.manual_seed(0)
model = TorchAnalysisGNN(g.metadata(), bench.IN_CH, bench.H, bench.OUT, bench.TASK_DICT, bench.LAYERS, dropout=0.3, use_jk=False, logit_fusion=False).to(dev).train()
flat = dp.FlatGradBuffer(model.parameters())
opt = torch.optim.AdamW(model.parameters(), lr=5e-3, weight_decay=5e-3, foreach=True)
graph.index_cache_enabled = False
def step():
    flat.zero()
    x = model.encode(I["pitch_spelling"], I["key_signature"], I["x_dict"], I["edge_index_dict"], I["batch_dict"], I["batch_size"], None, None)
    logits, offs, _ = model.forward_clf_fused(x)
    loss = 0.1 * x.pow(2).mean() + multitask_cross_entropy(logits, offs, label_mat, 0.1, -1).sum()
    loss.backward(); flat.clip_norm_(1.0); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/K:.2f} ms/step, GPU drained {1e3*(t2-t0)/K:.2f} ms/step, tail wait {1e3*(t2-t1):.2f} ms total")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
