#!/usr/bin/env python3
"""A/B timing of the hetero-SpMM kernel variants the way bench.py times the roofline kernel: HIP events around
hipGraph replays of 10 back-to-back launches (single launches bracketed by events carry ~8 us of event overhead).
usage: spmm_ab.py [H] [n_subgraphs]"""
import os
import statistics
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from analysisgnn_amd import ops  # noqa: E402
from analysisgnn_amd.graph import HeteroIndex  # noqa: E402
from analysisgnn_amd.synth import make_batch  # noqa: E402

VARIANTS = {"generic": 1024, "fast_v4": 2048, "fast": 0}
REP = 10


def replay_time(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP):
            fn()
    ts = []
    for _ in range(14):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / REP)
    return ts[2:]


def main():
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n_sub = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    dev = torch.device("cuda:0")
    b = make_batch(n_sub, 500)
    N = b.num_nodes["note"]
    eid = {et: torch.from_numpy(e).to(dev) for et, e in b.edge_index.items()}
    hix = HeteroIndex(eid, {"note": N})
    ets = list(eid)
    R = len(ets)
    x = torch.randn(N, H, device=dev, requires_grad=True)
    spec = ops.AggSpec(fwd=[hix.fwd[e] for e in ets], bwd=[hix.bwd[e] for e in ets], src_id=[0] * R, n_rows=N,
                       mean=True, shared_slot=False)
    gout = torch.randn(N, R * H, device=dev)
    b_fwd = sum(4 * (N + 1) + 4 * b.edge_index[e].shape[1] for e in ets) + 4 * H * (N + R * N)
    b_bwd = sum(4 * (N + 1) + 4 * b.edge_index[e].shape[1] + 4 * N for e in ets) + 4 * H * (R * N + N)
    res = {}
    ref = {}
    for rnd in range(3):
        for name, flag in VARIANTS.items():
            ops.SPMM_VARIANT = flag
            with torch.no_grad():
                res.setdefault(("fwd", name), []).extend(replay_time(lambda: ops.aggregate(spec, [x])))
            out = ops.aggregate(spec, [x])
            inv_cnt = out.grad_fn.saved_tensors[0]
            ctx = types.SimpleNamespace(spec=spec, H=H, src_rows=[N], has_self=False, self_rows=0,
                                        saved_tensors=(inv_cnt,), needs_input_grad=(False, False, True))
            res.setdefault(("bwd", name), []).extend(replay_time(lambda: ops._Aggregate.backward(ctx, gout)))
            if rnd == 0:
                ref[name] = (out.detach().clone(), ops._Aggregate.backward(ctx, gout)[2].clone())
    ops.SPMM_VARIANT = 0
    for (tag, name), v in sorted(res.items()):
        med, mn = statistics.median(v), min(v)
        bts = b_fwd if tag == "fwd" else b_bwd
        print(f"{tag} {name:8s} median {med:7.2f} us  min {mn:7.2f} us  alg {bts/1e6:.1f} MB -> {bts/med/1e6:6.2f} TB/s "
              f"({bts/med/1e6/8.0*100:.1f}% of 8 TB/s)  n={len(v)}")
    for name in VARIANTS:
        if name != "generic":
            print(f"{name} vs generic: fwd max|d| {float((ref[name][0]-ref['generic'][0]).abs().max()):.3e}  "
                  f"bwd max|d| {float((ref[name][1]-ref['generic'][1]).abs().max()):.3e}")
    print(f"N={N} R={R} H={H}")


if __name__ == "__main__":
    main()
