#!/usr/bin/env python3
"""A/B timing of the hetero-SpMM kernels on the C2 layer shape (32 x 500 notes, R=4, H=256), one
process, interleaved rounds (cdna_hip_programming.md §5.4 rule 24).  Prints median/min per variant and
the algorithmic-bytes roofline fraction."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from analysisgnn_amd import ops  # noqa: E402
from analysisgnn_amd.graph import HeteroIndex  # noqa: E402
from analysisgnn_amd.synth import make_batch  # noqa: E402


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
    E = sum(e.shape[1] for e in b.edge_index.values())
    x = torch.randn(N, H, device=dev, requires_grad=True)
    spec = ops.AggSpec(fwd=[hix.fwd[e] for e in ets], bwd=[hix.bwd[e] for e in ets], src_id=[0] * R, n_rows=N,
                       mean=True, shared_slot=False)
    gout = torch.randn(N, R * H, device=dev)
    b_fwd = sum(4 * (N + 1) + 4 * b.edge_index[e].shape[1] for e in ets) + 4 * H * (N + R * N)
    b_bwd = sum(4 * (N + 1) + 4 * b.edge_index[e].shape[1] + 4 * N for e in ets) + 4 * H * (R * N + N)
    res = {}
    for rnd in range(12):
        for legacy in ("generic", "fast_v4", "fast"):
            ops.SPMM_VARIANT = {"generic": 1024, "fast_v4": 2048, "fast": 0}[legacy]
            ops.SPMM_TRACE = []
            for _ in range(10):
                x.grad = None
                out = ops.aggregate(spec, [x])
                out.backward(gout)
            torch.cuda.synchronize()
            tr, ops.SPMM_TRACE = ops.SPMM_TRACE, None
            if rnd < 2:
                continue
            for tag, e0, e1, *_ in tr:
                res.setdefault((tag, legacy), []).append(e0.elapsed_time(e1) * 1e3)
    for (tag, legacy), v in sorted(res.items()):
        med, mn = statistics.median(v), min(v)
        bts = b_fwd if tag == "fwd" else b_bwd
        print(f"{tag} {legacy:7s} median {med:7.2f} us  min {mn:7.2f} us  "
              f"alg {bts/1e6:.1f} MB -> {bts/med/1e6:7.2f} TB/s ({bts/med/1e6/8.0*100:.1f}% of 8 TB/s)  n={len(v)}")
    print(f"N={N} R={R} E={E} H={H}")


if __name__ == "__main__":
    main()
