#!/usr/bin/env python3
"""Which Python call sites issue the small torch launches of one training step (copies, fills, cats, adds)?  One eager
step of the default bench workload under torch.profiler with stacks; prints, per (op, enclosing autograd nodes / ops), the count."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
from analysisgnn_amd import dp, graph  # noqa: E402
from analysisgnn_amd.heads import MultiTaskLoss, training_loss  # noqa: E402
from analysisgnn_amd.models import TorchAnalysisGNN  # noqa: E402
from analysisgnn_amd.synth import torch_inputs  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c2s"
dev = torch.device("cuda:0")
g, enc, hid, layers, tasks = bench.build_workload(wl, 0, 1)
I = torch_inputs(g, bench.IN_CH, dev, 0)
labels = bench.make_labels(I["batch_size"], dev, 100, tasks)
torch.manual_seed(0)
model = TorchAnalysisGNN(g.metadata(), bench.IN_CH, hid, bench.OUT, tasks, layers, dropout=0.3, use_jk=False, logit_fusion=False,
                         encoder_type=enc).to(dev).train()
clf = MultiTaskLoss(list(tasks)).to(dev)
params, tight = dp.plan_parameters(torch.nn.ModuleDict({"m": model, "c": clf}))
flat = dp.FlatGradBuffer(params, views=False, tight=tight)
dp.enable_wgrad_overlap(wl in ("c2", "c2s"), "sequence")
dp.defer_weight_grads(True)
opt = dp.FlatAdamW(params, flat, lr=5e-3, weight_decay=5e-3)
graph.index_cache_enabled = False
lm = torch.stack([labels[t] for t in tasks])
one = torch.ones((), device=dev)


def step():
    flat.zero()
    x = model.encode(I["pitch_spelling"], I["key_signature"], I["x_dict"], I["edge_index_dict"], I["batch_dict"], I["batch_size"],
                     I["neighbor_mask_node"], I["neighbor_mask_edge"])
    logits, offs, _ = model.forward_clf_fused(x)
    loss, _ = training_loss(logits, offs, lm, x, 0.1, 0.1, -1, task_params=clf.weights())
    loss.backward(gradient=one)
    flat.pack()
    opt.step(max_norm=1.0)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
want = ("aten::copy_", "aten::fill_", "aten::zero_", "aten::cat", "aten::add", "aten::add_", "aten::mul", "aten::clone", "aten::contiguous",
        "aten::zeros", "aten::sum", "aten::stack", "aten::index_select", "aten::div", "aten::where", "aten::_to_copy", "aten::mul_",
        "aten::masked_fill", "aten::select_backward", "aten::slice_backward", "aten::new_zeros", "aten::zeros_like")
acc = collections.Counter()
for ev in prof.events():
    if ev.name in want and ev.device_time_total > 0:
        chain, q = [], ev.cpu_parent
        while q is not None and len(chain) < 4:
            chain.append(q.name)
            q = q.cpu_parent
        shape = "x".join(str(d) for d in (ev.input_shapes[0] if ev.input_shapes else []))
        acc[(ev.name, " < ".join(chain) + "  [" + shape + "]")] += 1
for (name, site), n in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d}  {name:24s} {site}")
