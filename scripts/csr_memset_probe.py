#!/usr/bin/env python3
"""Does clearing the CSR build's counters with a hipMemsetAsync NODE leave them stale when the step is replayed from a
ONE-stream captured graph (round 1's k_scatter out-of-range write)?  Runs the 40-replay scenario of
tests/test_gpu_step.py ONCE with libagnn_hip_memsetprobe.so (make -C analysisgnn_amd/csrc memset-probe: the counters
cleared by hipMemsetAsync, everything else identical) and prints the device status word (edge positions that fell
outside their row: nothing is written out of range either way) and whether the replayed gradients equal the eager ones.
Usage: AGNN_LIB=analysisgnn_amd/libagnn_hip_memsetprobe.so python scripts/csr_memset_probe.py   (or without AGNN_LIB
for the shipped library as the control)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from analysisgnn_amd import _lib, dp, graph  # noqa: E402
from analysisgnn_amd.encoders import _HybridMixin  # noqa: E402
from analysisgnn_amd.heads import training_loss  # noqa: E402
from analysisgnn_amd.models import TorchAnalysisGNN  # noqa: E402
from analysisgnn_amd.synth import make_batch, torch_inputs  # noqa: E402

dev = torch.device("cuda", 0)
tasks = {"cadence": 4, "localkey": 50, "romanNumeral": 185}
g = make_batch(4, 500)
I = torch_inputs(g, 25, dev, seed=0)
labels = torch.stack([torch.randint(0, c, (I["batch_size"],), generator=torch.Generator().manual_seed(i)).to(dev)
                      for i, c in enumerate(tasks.values())])
torch.manual_seed(0)
model = TorchAnalysisGNN(g.metadata(), 25, 256, 128, tasks, 3, dropout=0.0, use_jk=False, logit_fusion=False).to(dev).train()
params, tight = dp.plan_parameters(model)
flat = dp.FlatGradBuffer(params, views=False, tight=tight)
graph.index_cache_enabled = False
_HybridMixin.overlap_sequence_branch = False
dp.enable_wgrad_overlap(False)


def fwd_bwd():
    flat.zero()
    x = model.encode(I["pitch_spelling"], I["key_signature"], I["x_dict"], I["edge_index_dict"], I["batch_dict"],
                     I["batch_size"], None, None)
    logits, offs, _ = model.forward_clf_fused(x)
    loss, _ = training_loss(logits, offs, labels, x, 0.1, 0.1, -1)
    loss.backward()
    flat.pack()
    return loss


side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    for _ in range(2):
        fwd_bwd()
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize()
g_eager = flat.flat.clone()
cg = torch.cuda.CUDAGraph()
with torch.cuda.graph(cg):
    fwd_bwd()
n = int(os.environ.get("REPLAYS", "60"))
for i in range(n):                                   # back to back, no host sync in between (as the failing runs were)
    cg.replay()
torch.cuda.synchronize()
print(json.dumps({"library": os.path.basename(_lib.LIB_PATH), "replays": n, "status_word": int(_lib.status_word(dev).item()),
                  "grads_equal_eager": bool(torch.equal(flat.flat, g_eager)),
                  "max_abs_diff": float((flat.flat - g_eager).abs().max())}))
