#!/usr/bin/env python3
"""aten-op inventory of one eager training step (which host-level ops produce the small launches)."""
import os, sys
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
model = TorchAnalysisGNN(g.metadata(), bench.IN_CH, bench.H, bench.OUT, bench.TASK_DICT, bench.LAYERS, dropout=0.3, use_jk=False, logit_fusion=False).to(dev).train()
flat = dp.FlatGradBuffer(model.parameters(), views=False)
opt = dp.FlatAdamW(model.parameters(), flat, lr=5e-3, weight_decay=5e-3)
graph.index_cache_enabled = False
def step():
    flat.zero()
    x = model.encode(I["pitch_spelling"], I["key_signature"], I["x_dict"], I["edge_index_dict"], I["batch_dict"], I["batch_size"], None, None)
    logits, offs, _ = model.forward_clf_fused(x)
    loss = 0.1 * x.pow(2).mean() + multitask_cross_entropy(logits, offs, label_mat, 0.1, -1).sum()
    loss.backward(); flat.pack(); flat.clip_norm_(1.0); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=300, max_name_column_width=40, max_shapes_column_width=70))
