#!/usr/bin/env python3
"""The aggregation launches of one training step on the SAMPLED batch (bench.py workload c2s), 40 times each back to back,
for `rocprofv3 --kernel-trace --stats` / `--pmc`: forward of layer 0 (untrimmed), forward and backward of layer 1 and 2
(trimmed: row ends from rowend; backward with the column limit and 1/deg scales -> FILT instantiation) and the onset
pooling forward / backward (self numerator, self loops skipped, column limit).  Prints the algorithmic bytes per launch
 so that the kernel-trace averages can be turned into fractions of 8 TB/s."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from analysisgnn_amd import ops  # noqa: E402
from analysisgnn_amd.encoders import TrimPlan  # noqa: E402
from analysisgnn_amd.graph import HeteroIndex  # noqa: E402
from analysisgnn_amd.models import onset_pool  # noqa: E402
from analysisgnn_amd.synth import make_sampled_batch, torch_inputs  # noqa: E402

REP = int(os.environ.get("REP", "40"))
WITH_ROOT = os.environ.get("ROOT", "0") == "1"      # the SAGE layer's root operand rides along (agnn_spmm_root_f32)
TIMES = os.environ.get("TIMES", "0") == "1"         # HIP events around every launch: average microseconds per layer and direction
H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
g = make_sampled_batch(32, 500, (5, 5))
I = torch_inputs(g, 25, dev, 0)
n_nodes = {"note": g.num_nodes["note"]}
hix = HeteroIndex(I["edge_index_dict"], n_nodes)
ets = list(hix.edge_types)
plan = TrimPlan(3, I["x_dict"], I["edge_index_dict"], I["neighbor_mask_node"], I["neighbor_mask_edge"])
hix.prepare_trim(plan.e_keep)
report = []
for layer in range(3):
    n_dst = plan.n_keep[layer]["note"]
    n_src = plan.n_keep[layer - 1]["note"] if layer else n_nodes["note"]
    e_keep = [plan.e_keep[layer][et] for et in ets]
    spec = ops.AggSpec(fwd=[hix.fwd[e] for e in ets], bwd=[hix.bwd[e] for e in ets], src_id=[0] * len(ets), n_rows=n_dst, mean=True,
                       shared_slot=False, e_limit=e_keep if any(k is not None for k in e_keep) else None, root=WITH_ROOT)
    x = torch.randn(n_src, H, device=dev, requires_grad=True)
    gout = torch.randn(n_dst, (len(ets) + WITH_ROOT) * H, device=dev)
    if TIMES:
        ops.SPMM_TRACE = []
    for _ in range(REP):
        out = ops.aggregate(spec, [x], self_t=x if WITH_ROOT else None)
        out.backward(gout)
        x.grad = None
    if TIMES:
        torch.cuda.synchronize()
        tr, ops.SPMM_TRACE = ops.SPMM_TRACE, None
        for tag in ("fwd", "bwd"):
            us = sorted(e0.elapsed_time(e1) * 1e3 for t, e0, e1, *_ in tr if t == tag)
            print(f"layer {layer} {tag} root={int(WITH_ROOT)}: median {us[len(us) // 2]:.1f} us  p10 {us[len(us) // 10]:.1f}", file=sys.stderr)
    idx, e_kept, srcs = 0, 0, []
    for et, k in zip(ets, e_keep):
        ei = g.edge_index[et][:, :k] if k is not None else g.edge_index[et]
        ei = ei[:, ei[1] < n_dst]
        idx += 4 * (n_dst + 1) + 4 * ei.shape[1]
        e_kept += int(ei.shape[1])
        srcs.append(ei[0])
    nsu = int(np.unique(np.concatenate(srcs)).size)
    b_fwd = idx + 4 * H * (nsu + len(ets) * n_dst)
    # backward: indices of the transposed CSR once, 1/deg scales once, every dout slot row once, one gradient row per source
    b_bwd = sum(4 * (n_src + 1) for _ in ets) + 4 * e_kept + 4 * len(ets) * n_dst + 4 * H * (len(ets) * n_dst + n_src)
    report.append({"layer": layer, "n_dst": n_dst, "n_src": n_src, "edges": e_kept, "n_src_unique": nsu, "alg_bytes_fwd": b_fwd,
                   "alg_bytes_bwd": b_bwd, "trimmed": any(k is not None for k in e_keep)})
# onset pooling (models/analysis.py:580-587) on the encoder output [batch_size, H]
xb = torch.randn(I["batch_size"], H, device=dev, requires_grad=True)
gp = torch.randn(I["batch_size"], 2 * H, device=dev)
for _ in range(REP):
    y = onset_pool(xb, I["edge_index_dict"][("note", "onset", "note")], I["batch_size"], hix)
    y.backward(gp)
    xb.grad = None
torch.cuda.synchronize()
print(json.dumps({"H": H, "rep": REP, "launches": report}))
