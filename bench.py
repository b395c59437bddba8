#!/usr/bin/env python3
"""bench.py — subgraph-nodes/sec, forward+backward, HybridGNN L=3 H=256 (BASELINE.json).

One "step" = one training pass of the hot path over one batch of synthetic input already resident in
HBM, as a training step of the reference runs it (analysisgnn-train CLI defaults, train/train_analysisgnn.py:
52-97): TorchAnalysisGNN(encoder=HybridGNN, L=3, H=256, out=128, 21 task heads, dropout 0.3, use_jk off,
logit_fusion off) forward on a neighbour-SAMPLED batch with the per-hop counts passed (models/analysis.py:
960-961 — every layer is trimmed), the default objective (--mt_strategy wloss: learned task weights,
models/chord.py:39-49, / number of tasks, + 0.1 * feature norm: models/analysis.py:1034-1036, :1072), backward,
gradient all-reduce (N>1), gradient clipping (1.0) and AdamW.  The COO->CSR index is rebuilt every step (a
fresh sampled batch arrives every step in the reference's loader).  Default workload "c2s": 32 sampled
subgraphs (500 target notes + [5,5] hops each) per GPU (weak scaling: per-GPU work fixed).

Launch:  python bench.py [--gpus N --steps K --warmup W]
N > 1 without WORLD_SIZE in the environment: this process only spawns `python -m torch.distributed.run
--nproc-per-node N bench.py ...` (before anything touches the GPU) and relays rank 0's JSON line and the exit
code; under torch.distributed.run it is one rank.  Prints ONE JSON line on rank 0 with `roofline` (the
workload's dominant hand-written aggregation kernel, HBM bound, live HIP-event timing) and `cpu_baseline`
(oracle port on host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TASK_DICT = {  # train/train_analysisgnn.py:22-45 (duplicate key "organ_point" collapses, as in the reference)
    "cadence": 4, "localkey": 50, "tonkey": 50, "quality": 15, "inversion": 4, "root": 38, "bass": 38,
    "degree1": 22, "degree2": 22, "hrythm": 2, "pcset": 94, "romanNumeral": 185, "section": 2, "phrase": 2,
    "organ_point": 2, "tpc_in_label": 2, "tpc_is_root": 2, "tpc_is_bass": 2, "downbeat": 45, "note_degree": 49,
    "staff": 4,
}
C5_TASKS = {"cadence": 4, "localkey": 50, "romanNumeral": 185}
HBM_PEAK = 8.0e12          # B/s, MI355X spec (MI355X_MICROARCH.md)
MFMA_F32_PEAK = 157.3e12   # FLOP/s, dense fp32 MFMA (v_mfma_f32_32x32x2_f32; MI355X_MICROARCH.md)
N_SUB, N_NOTES, IN_CH, H, OUT, LAYERS = 32, 500, 25, 256, 128, 3

WORKLOADS = {
    "c2s": "C2 (sampled, the batch a training step sees): HybridGNN L=3 H=256 out=128, 21 task heads, 32 neighbour-sampled "
           "subgraphs x (500 target notes + [5,5] hops) per GPU, per-hop counts passed (every layer trimmed)",
    "c2": "C2 (whole graphs, no trimming): HybridGNN L=3 H=256 out=128, 21 task heads, 32 subgraphs x 500 notes per GPU",
    "c2d": "C2 with the batch assembled ON THE DEVICE inside the timed step: 64 scores x 1500 notes resident in HBM, every step "
           "samples 32 fresh windows x (500 target notes + [5,5] hops) into static-shape buffers (agnn_sample_hops), gathers their "
           "features, rebuilds the CSR and trains on them; HybridGNN L=3 H=256 out=128, 21 task heads",
    "c3d": "C3 with the batch assembled ON THE DEVICE inside the timed step: 64 scores x 1500 notes with their beats and measures "
           "resident in HBM, every step samples 32 fresh windows x (500 target notes + [5,5] hops), adds each window's beats / measures "
           "and the notes' membership edges (agnn_sample_hops + agnn_sample_members), gathers features, rebuilds the CSR and trains; "
           "HGT L=3 H=256 heads=4, 6 relation types, 21 task heads",
    "c3": "C3: HGT L=3 H=256 heads=4, note+beat+measure nodes, 6 relation types, 21 task heads, 32 subgraphs x 500 notes per GPU "
          "(whole graphs)",
    "c5": "C5: MetricalGNN L=4 H=512, heads cadence/localkey/romanNumeral, 32 subgraphs x 500 notes per GPU (whole graphs)",
}
METRICS = {"c2s": "subgraph-nodes/sec fwd+bwd, HybridGNN L=3 H=256", "c2": "subgraph-nodes/sec fwd+bwd, HybridGNN L=3 H=256",
           "c2d": "subgraph-nodes/sec fwd+bwd, HybridGNN L=3 H=256",
           "c3": "subgraph-nodes/sec fwd+bwd, HGT L=3 H=256", "c3d": "subgraph-nodes/sec fwd+bwd, HGT L=3 H=256", "c5": "subgraph-nodes/sec fwd+bwd, MetricalGNN L=4 H=512"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="c2s", choices=list(WORKLOADS),
                    help="c2s (default, the BASELINE metric on the batch a training step of the reference sees): " + WORKLOADS["c2s"] +
                         "; c2: " + WORKLOADS["c2"] + "; c3: " + WORKLOADS["c3"] + "; c5: " + WORKLOADS["c5"])
    ap.add_argument("--mt-strategy", default="wloss", choices=["wloss", "sum"],
                    help="wloss (reference CLI default): learned uncertainty weights per task; sum: plain sum of the task losses")
    ap.add_argument("--no-graph", action="store_true", help="issue every launch eagerly instead of replaying hipGraphs")
    ap.add_argument("--schedule", default="auto", choices=["auto", "late", "plain"],
                    help="backward schedule of the hybrid encoders: capture both and keep the faster (auto), or force one")
    ap.add_argument("--no-defer", action="store_true", help="A/B only: weight gradients where they are computed, not deferred")
    ap.add_argument("--defer-all", action="store_true", help="A/B only: deferred (batched) weight gradients for MetricalGNN too")
    ap.add_argument("--no-wgrad-overlap", action="store_true", help="A/B only: weight gradients on the main stream")
    ap.add_argument("--force-wgrad-overlap", action="store_true", help="A/B only: the weight-gradient stream for any workload (default: c2 / c2s)")
    ap.add_argument("--items-home", default="auto", choices=["auto", "on", "off"],
                    help="a branch stream's deferred weight-gradient products: with the main chain's flush (on) / on the branch's own (off) / measured (auto)")
    ap.add_argument("--fused-heads", action="store_true", help="A/B only: the head block's forward in one launch (agnn_heads_fwd_f32)")
    ap.add_argument("--set", action="append", default=[], metavar="MODULE.NAME=VALUE",
                    help="A/B runs: set a module-level switch of the package, e.g. --set encoders.JOIN_ONE_LAUNCH=False (value: a Python literal)")
    ap.add_argument("--lr", type=float, default=5e-4, help="AdamW rate of the run (the reference's warm-up schedule at step 50 of 500: 5e-3 * 50 / 500)")
    ap.add_argument("--flush-keep", type=float, default=None,
                    help="share of the main flush point's weight-gradient FLOPs that runs there (rest: the sequence branch's flush); default: measured")
    ap.add_argument("--no-yield-gemm", action="store_true", help="A/B only: the graph stack's second layer does not wait for the GRU's inner input projection")
    ap.add_argument("--no-tune-gemm", action="store_true",
                    help="library GEMMs as hipBLASLt's heuristic picks them (default: PyTorch TunableOp times the candidates during the eager warm-up steps)")
    ap.add_argument("--side-priority", type=int, default=0, help="A/B only: HIP priority of the sequence branch's stream (-1 = high)")
    ap.add_argument("--wgrad-scope", default="sequence", help="A/B only: kinds of weight-gradient work on the side stream")
    ap.add_argument("--library-wgrad", action="store_true", help="A/B only: weight gradients through the library GEMM")
    ap.add_argument("--blas", default=None, choices=[None, "hipblaslt", "rocblas"], help="A/B only: torch's preferred BLAS library")
    ap.add_argument("--graph-dot", default=None, help="debugging: write the captured step's dependency graph as DOT to this path")
    ap.add_argument("--host-times", action="store_true", help="debugging: report the host time inside the replay calls")
    ap.add_argument("--one-bucket", action="store_true", help="N > 1, A/B only: one gradient message between the two graph replays "
                    "instead of two buckets with the first all-reduce beside the input layers' backward")
    ap.add_argument("--no-other", action="store_true", help="skip the secondary (whole-graph C2) measurement of the default run")
    return ap.parse_args(argv)


def self_launch(args) -> int:
    """`python bench.py --gpus N` as ONE command (the reference gets its ranks from one command too:
    train/train_analysisgnn.py:138-146).  The parent never touches the GPU; it relays the children's output."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:                      # rank 0 prints the one JSON line; anything else goes to stderr
        if ln.lstrip().startswith("{") and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        sys.stdout.write(line)
        sys.stdout.flush()
    return rc if rc != 0 else (0 if line is not None else 1)


def make_labels(n, device, seed, tasks):
    import torch
    g = torch.Generator().manual_seed(seed)
    return {t: torch.randint(0, c, (n,), generator=g).to(device) for t, c in tasks.items()}


def _dump_capture_dot(path: str, dev) -> None:
    """Inside a stream capture: write the graph captured so far as DOT (hipStreamGetCaptureInfo_v2 + hipGraphDebugDotPrint)."""
    import ctypes as C
    import torch
    hip = C.CDLL("libamdhip64.so")
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    status, cid, graph, deps, ndeps = C.c_int(0), C.c_ulonglong(0), C.c_void_p(0), C.c_void_p(0), C.c_size_t(0)
    rc = hip.hipStreamGetCaptureInfo_v2(stream, C.byref(status), C.byref(cid), C.byref(graph), C.byref(deps), C.byref(ndeps))
    rc2 = hip.hipGraphDebugDotPrint(graph, path.encode(), C.c_uint(0)) if rc == 0 and graph.value else -1
    print(f"[bench] capture graph -> {path}: hipStreamGetCaptureInfo_v2 rc={rc} status={status.value}, hipGraphDebugDotPrint rc={rc2}", file=sys.stderr)


def build_workload(name: str, rank: int, world: int, n_sub: int = N_SUB):
    """(graph, encoder_type, hidden, layers, tasks).  Rank r of G takes subgraphs {i : i mod G = r} of the global batch of
    32 * G (dp.shard_units; DistributedSampler semantics): independent units, no data-path collective."""
    from analysisgnn_amd.dp import shard_units
    from analysisgnn_amd.synth import make_batch, make_sampled_batch
    seeds = shard_units(n_sub * world, rank, world)
    if name in ("c2s", "c2d"):                     # c2d's CPU baseline / shapes: the same kind of batch, assembled on the host
        g = make_sampled_batch(n_sub, N_NOTES, (5,) * (LAYERS - 1), seeds=seeds)      # train_analysisgnn.py:82,154
        return g, "hybridgnn", H, LAYERS, TASK_DICT
    if name == "c2":
        return make_batch(n_sub, N_NOTES, seeds=seeds), "hybridgnn", H, LAYERS, TASK_DICT
    if name in ("c3", "c3d"):                      # c3d's shapes for the host-side helpers: the same kind of graphs
        g = make_batch(n_sub, N_NOTES, seeds=seeds, add_beats=True, add_measures=True)
        keep = [et for et in g.edge_types if et[0] == "note"]       # 4 note-note + note->beat + note->measure
        g.edge_index = {et: g.edge_index[et] for et in keep}
        return g, "hgt", H, LAYERS, TASK_DICT
    if name == "c5":
        return make_batch(n_sub, N_NOTES, seeds=seeds), "metricalgnn", 512, 4, C5_TASKS
    raise ValueError(name)


def algorithmic_flops(enc, I, hid, layers, tasks):
    """ALGORITHMIC matrix FLOPs of one training step (what a dense-algebra implementation of the reference's layers cannot
    avoid: 2 m k n per product, every product once forward and twice backward — dX and dW — except the input layers' dX,
    their inputs being data).  Aggregations, norms, activations, the objective and the optimizer are byte work and are not
    counted.  SAGE layers follow the build's formulation (R neighbour products + ONE root product per layer,
    the per-relation root weights pre-summed; a trimmed layer that keeps no edge is its root product alone); rows per layer
    follow PyG's trim_to_layer.  -> (total, breakdown dict)."""
    from analysisgnn_amd.encoders import TrimPlan
    mm = lambda m, k, n: 2.0 * m * k * n                                           # noqa: E731
    n_all = {t: int(v.shape[0]) for t, v in I["x_dict"].items()}
    B = int(I["batch_size"])
    plan = TrimPlan(layers, I["x_dict"], I["edge_index_dict"], I["neighbor_mask_node"], I["neighbor_mask_edge"])
    ets = list(I["edge_index_dict"].keys())
    T = len(tasks)
    fwd, no_dx = {}, 0.0
    first = sum(mm(n, IN_CH + (128 if t == "note" else 0), hid) for t, n in n_all.items())
    fwd["input MLPs (project_dict)"] = first + sum(mm(n, hid, hid) for n in n_all.values())
    no_dx += first
    stack = 0.0
    for layer in range(layers):
        nk, ek = plan.n_keep[layer], plan.e_keep[layer]
        live = [et for et in ets if ek[et] is None or ek[et] > 0]
        if enc == "hgt":
            D, heads = hid // 4, 4
            n_src = plan.n_keep[layer - 1] if layer > 0 else n_all
            for t in n_all:
                stack += mm(n_src[t] if any(et[0] == t for et in live) else nk[t], hid, 3 * hid) + mm(nk[t], hid, hid)   # kqv, out_lin
            for et in live:
                stack += 2 * heads * mm(n_src[et[0]], D, D)                          # k_rel and v_rel: one D x D matrix per head
        else:
            for d in n_all:
                r = sum(1 for et in live if et[2] == d)
                if r or any(et[2] == d for et in ets):
                    stack += mm(nk[d], (r + 1) * hid, hid)
    fwd["GNN stack projections"] = stack
    if enc in ("hybridgnn", "hgt"):
        hh = hid // 2
        fwd["sequence branch (GRU input projections, recurrences, MLP, cat_proj)"] = (
            2 * mm(B, hid, 6 * hh) + 2 * 2 * B * 2.0 * 3 * hh * hh + 2 * mm(B, hid, hid) + mm(B, 2 * hid, hid))
    else:
        fwd["encoder output MLP"] = 2 * mm(B, hid, hid)
    fwd["project_enc"] = mm(B, 2 * hid, hid) + mm(B, hid, OUT) + mm(B, OUT, OUT)
    fwd["task heads"] = mm(B, OUT, T * (OUT // 2)) + sum(mm(B, OUT // 2, c) for c in tasks.values())
    total = 3.0 * sum(fwd.values()) - no_dx
    return total, {k: 3.0 * v for k, v in fwd.items()}


def reference_objective(logits, labels, feat, params):
    """The objective exactly as the reference composes it (models/analysis.py:1034-1036, :1072; models/chord.py:39-49),
    on torch ops — the CPU baseline's loss."""
    import torch
    import torch.nn.functional as F
    loss_sum = 0
    for i, (t, y) in enumerate(labels.items()):
        ce = F.cross_entropy(logits[t], y, ignore_index=-1, label_smoothing=0.1)
        loss_sum = loss_sum + ((0.5 / (params[i] ** 2) * ce + torch.log(1 + params[i] ** 2)) if params is not None else ce)
    return loss_sum / len(labels) + 0.1 * feat.pow(2).mean()


def _percentiles(ts):
    ts = sorted(ts)
    k = len(ts)
    return ts[k // 2], ts[max(0, int(0.1 * (k - 1)))], ts[min(k - 1, int(round(0.9 * (k - 1))))]


def cpu_baseline(workload: str, mt_strategy: str):
    """Oracle port (oracle/encoders_ref.py, pure PyTorch CPU: per relation index_select -> index_add_ -> divide -> Linear,
    Python loops over relations and layers) of the same model on the same workload, fwd + objective + bwd.
    BASELINE.md §3 protocol, bounded to ~30 s of CPU work: all cores of this job's CPU share on the full per-GPU batch
    (32 subgraphs; 3 warm-up + 10 timed), and one thread on C1 (one subgraph; 2 + 8)."""
    import torch
    from analysisgnn_amd.models import TorchAnalysisGNN
    from analysisgnn_amd.synth import torch_inputs
    from oracle import encoders_ref as E, rnn_ref
    rnn_ref.USE_FAST = True

    def run(n_sub, threads, warm, iters):
        g, enc, hid, layers, tasks = build_workload(workload, 0, 1, n_sub)
        torch.manual_seed(0)
        m = TorchAnalysisGNN(g.metadata(), IN_CH, hid, OUT, tasks, layers, dropout=0.3, use_jk=False, logit_fusion=False,
                             encoder_type=enc)
        P = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
        params = torch.ones(len(tasks), requires_grad=True) if mt_strategy == "wloss" else None
        I = torch_inputs(g, IN_CH, "cpu", 0)
        labels = make_labels(I["batch_size"], "cpu", 1, tasks)
        torch.set_num_threads(threads)

        def step():
            for p in P.values():
                p.grad = None
            x = E.analysis_encode(P, enc, g.metadata(), layers, I["pitch_spelling"], I["key_signature"], I["x_dict"],
                                  I["edge_index_dict"], I["batch_dict"], I["batch_size"], I["neighbor_mask_node"],
                                  I["neighbor_mask_edge"])
            reference_objective(E.analysis_logits(P, x, list(tasks)), labels, x, params).backward()
        for _ in range(warm):
            step()
        ts = []
        for _ in range(iters):
            t0 = time.perf_counter()
            step()
            ts.append(time.perf_counter() - t0)
        med, p10, p90 = _percentiles(ts)
        return I["batch_size"] / med, med, p10, p90

    # the GPU box gives one-GPU jobs a 16-CPU share while torch sees every host core: oversubscribing made this
    # oracle 6x slower, so the thread count is the share actually available
    cores = max(1, min(16, len(os.sched_getaffinity(0)), os.cpu_count() or 1))
    v, med, p10, p90 = run(N_SUB, cores, 3, 10)
    v1, med1, p101, p901 = run(1, 1, 2, 8)
    return {"value": v, "unit": "subgraph-nodes/s", "cores": cores, "kind": "port",
            "sample": f"workload {workload}: {N_SUB} subgraphs, fwd + objective ({mt_strategy}) + bwd (no optimizer), eval-mode oracle, "
                      f"3 warm-up + 10 timed iters, median {med*1e3:.0f} ms/iter (p10 {p10*1e3:.0f}, p90 {p90*1e3:.0f}), {cores} threads "
                      f"(host has {os.cpu_count()} CPUs; this job's share is {len(os.sched_getaffinity(0))})",
            "single_thread": {"value": v1, "unit": "subgraph-nodes/s", "cores": 1,
                              "sample": f"C1 shape (1 subgraph of workload {workload}), 2 warm-up + 8 timed iters, median {med1*1e3:.0f} ms/iter "
                                        f"(p10 {p101*1e3:.0f}, p90 {p901*1e3:.0f})"}}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    from analysisgnn_amd import dp, graph
    from analysisgnn_amd.heads import MultiTaskLoss, training_loss
    from analysisgnn_amd.models import TorchAnalysisGNN
    from analysisgnn_amd.synth import torch_inputs

    if args.library_wgrad:
        from analysisgnn_amd import linear as _lin
        _lin.ENABLED = False
    if args.blas:
        torch.backends.cuda.preferred_blas_library("cublaslt" if args.blas == "hipblaslt" else "cublas")
    rank, local, world = dp.init_distributed()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    local = local % torch.cuda.device_count()          # ranks may share a device only in the gloo rehearsal
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    g, enc, hid, layers, tasks = build_workload(args.workload, rank, world)
    sampler = None
    if args.workload in ("c2d", "c3d"):
        import numpy as np
        from analysisgnn_amd.batching import DeviceSampler, ScoreStore
        from analysisgnn_amd.synth import make_score_graph
        # the corpus (replicated on every rank, as a dataset would be): 64 synthetic scores of 1500 notes; ranks draw
        # different windows (their own host generator), i.e. disjoint subgraphs of the global batch
        hetero = args.workload == "c3d"
        scores = [make_score_graph(seed=1000 + i, n_notes=1500, add_beats=hetero, add_measures=hetero) for i in range(64)]
        for sg in scores:                                           # C3's relation set: the types whose source is a note
            sg.edge_index = {et: e for et, e in sg.edge_index.items() if et[0] == "note"}
        store = ScoreStore(scores, IN_CH, dev, tasks=tasks, seed=0)
        # per-subgraph capacity 32 new notes per hop, batch-wide pools of 8 per subgraph (a window of consecutive notes adds a few:
        # measured ~2.7 per subgraph and hop on this corpus; overflow is counted in sampler.dropped(), reported below)
        sampler = DeviceSampler(store, N_SUB, N_NOTES, (5,) * (layers - 1), (32,) * (layers - 1), seed=1 + rank,
                                pool=(8 * N_SUB,) * (layers - 1))
        g_meta = sampler.metadata()
        win_rng = np.random.default_rng(100 + rank)
        sampler.set_windows(store.random_windows(N_SUB, N_NOTES, win_rng))
        I = sampler.sample()
        labels = I["labels"]
    else:
        I = torch_inputs(g, IN_CH, dev, seed=rank)
        labels = make_labels(I["batch_size"], dev, 100 + rank, tasks)
    torch.manual_seed(0)                                            # identical replicas
    model = TorchAnalysisGNN(g_meta if sampler is not None else g.metadata(), IN_CH, hid, OUT, tasks, layers, dropout=0.3, use_jk=False,
                             logit_fusion=False, encoder_type=enc).to(dev).train()
    clf_loss = MultiTaskLoss(list(tasks), requires_grad=(args.mt_strategy == "wloss")).to(dev)     # analysis.py:899-908
    trainable = torch.nn.ModuleDict({"model": model, "clf_loss": clf_loss})
    # parameters consumed concatenated (task-head layers, GRU direction pairs) sit back to back: their cats are views
    # N > 1: the input layers' parameters (their gradients come last) form a second, small bucket at the end of the flat buffer;
    # the first bucket (everything else, ~98 % of the message) is all-reduced while their backward still runs
    buckets = world > 1 and not args.one_bucket
    late = model.late_parameters() if buckets else []
    model.split_backward = buckets
    params, tight = dp.plan_parameters(trainable, late=late)
    flat = dp.FlatGradBuffer(params, views=False, tight=tight, late=late)
    # the GRU layers' weight-gradient work on its own stream, joined in flat.pack() — only where the sequence branch is
    # the longer one (C2; with HGT / MetricalGNN the graph branch is, and the extra stream only adds contention)
    # (round 3, measured and not kept: + "embed" — the embedding tables' gradient, the backward pass's last node, beside the input
    # layers' deferred weight gradients: 3.376 vs 3.346 ms, two alternating pairs on one box; `--wgrad-scope sequence,embed`)
    dp.enable_wgrad_overlap(not args.no_wgrad_overlap and (args.workload in ("c2", "c2s") or args.force_wgrad_overlap),
                            "all" if args.wgrad_scope == "all" else args.wgrad_scope.split(","))
    # dW / db of the projections on the main stream wait until that stream has slack (the GNN stack's backward is done, the
    # sequence branch's is not): the hybrid encoders only
    dp.defer_weight_grads(not args.no_defer and (enc in ("hybridgnn", "hgt") or args.defer_all))
    from analysisgnn_amd import encoders as _enc0
    _enc0.SIDE_STREAM_PRIORITY = args.side_priority
    for kv in args.set:
        import ast, importlib
        name, val = kv.split("=", 1)
        mod, attr = name.rsplit(".", 1)
        m = importlib.import_module("analysisgnn_amd." + mod)
        if not hasattr(m, attr):
            raise SystemExit(f"--set: analysisgnn_amd.{mod} has no switch {attr}")
        setattr(m, attr, ast.literal_eval(val))
    from analysisgnn_amd import linear as _lin
    if args.fused_heads:
        from analysisgnn_amd import heads as _heads
        _heads.HEADS_FUSED = True
    if args.no_yield_gemm:
        from analysisgnn_amd import gru as _gru
        _gru.YIELD_TO_PROJECTIONS = False
    if args.flush_keep is not None:
        _lin.FLUSH_KEEP = args.flush_keep
    if args.items_home != "auto":
        _lin.ITEMS_HOME = args.items_home == "on"
    # analysis.py:1380-1381 + train_analysisgnn.py:58-59: AdamW(lr 5e-3, weight decay 5e-3) behind a LINEAR WARM-UP over the first
    # 500 steps (analysis.py:1390-1399).  A bench run is the first ~50 steps of that schedule, where the reference's rate is
    # <= 5e-4; the step's launches take their rate as a constant, so the run uses that value.  (At the peak rate from step 0, on
    # random labels, the un-normalised HGT stack's activations grow ~40 x per layer within 25 steps and overflow fp32: round 3.)
    opt = dp.FlatAdamW(params, flat, lr=args.lr, weight_decay=5e-3)
    graph.index_cache_enabled = False                               # fresh batch every step: rebuild the CSR

    label_mat = I["label_matrix"] if sampler is not None else torch.stack([labels[t] for t in tasks])      # [T, N]
    from analysisgnn_amd.heads import unit_gradient
    one = unit_gradient(dev)                                        # THE resident 1.0: the objective's backward then has no launch

    from analysisgnn_amd import _lib as _agnn_lib

    def _lib_stamp(name):                                           # AGNN_STAMPS=1: device time stamps inside the captured step
        _agnn_lib.stamp(name, dev)

    def fwd_bwd():
        _lib_stamp("step start")
        flat.zero()
        if sampler is not None:
            sampler.sample()                                        # this step's batch: sampled and gathered on the device
        x = model.encode(I["pitch_spelling"], I["key_signature"], I["x_dict"], I["edge_index_dict"], I["batch_dict"],
                         I["batch_size"], I["neighbor_mask_node"], I["neighbor_mask_edge"])      # analysis.py:953-961
        logits, offs, _ = model.forward_clf_fused(x)
        loss, _ = training_loss(logits, offs, label_mat, x, 0.1, 0.1, -1, task_params=clf_loss.weights())   # analysis.py:1034-1036, :1072
        _lib_stamp("forward + objective issued (main)")
        loss.backward(gradient=one)                                # THE resident 1.0: no fill launch, and the objective's backward launches nothing
        flat.pack("early" if buckets else None)                    # two buckets: backward stopped behind the input layers
        _lib_stamp("step end (gradients gathered)")
        return loss

    def bwd_tail():                                                 # two buckets only: the input layers' backward + their bucket's gather
        model.finish_backward()
        flat.pack("late")

    def reduce_between(run_tail):
        """What sits between the step's launches and the optimizer's: the gradient exchange.  Two buckets: ship the first,
        run the input layers' backward beside it, ship theirs, wait, divide."""
        if not buckets:
            flat.all_reduce_mean()
            return
        work = flat.all_reduce_early_async()
        run_tail()
        flat.all_reduce_late_and_finish(work)

    def update():
        opt.step(max_norm=1.0)                                      # clip + AdamW: agnn_adamw_f32 (two launches)

    # The whole step is a few hundred launches; issued one by one from Python they cost more host time than GPU time, so
    # the two launch sequences (forward+backward+gradient gather; clip+AdamW) are captured ONCE into hipGraphs and
    # replayed, with the gradient all-reduce between them.  The batch tensors are static buffers that a loader would
    # refill; the graphs still rebuild the CSR from the COO edge lists on every replay.
    graphs = None
    loss_ref = [None]
    graph_mode = "eager (--no-graph)"
    # With a process group in the process the NCCL watchdog thread issues HIP calls of its own; in the default (global) capture
    # error mode any such call from another thread invalidates a capture in progress.  thread_local confines the check to the
    # capturing thread (what torch's own DDP + CUDA-graph recipes use).  One-rank runs keep the stricter default.
    cap_mode = "thread_local" if (world > 1 or torch.distributed.is_initialized()) else "global"
    # The forward / input-gradient projections are library GEMMs.  hipBLASLt's heuristic pick for these tall shapes is not its
    # fastest kernel (round 2: 66 vs 119 us for [16000, 1024] x [1024, 256]); PyTorch's TunableOp times the candidates the first
    # time a shape is seen — here during the eager warm-up steps — and is frozen before the step is captured.  C2 3.08 -> 3.02 ms,
    # C5 6.33 -> 6.20 (round 3; in round 2 the tuned kernels shared the chip worse with the recurrence kernels, which now keep
    # their CUs to themselves).  A library setting any user of the package can make; `--no-tune-gemm` measures without it.
    gemm_selection = "hipBLASLt heuristic"
    if not args.no_tune_gemm:
        try:
            import tempfile
            _tn = torch.cuda.tunable
            _tn.set_filename(os.path.join(tempfile.gettempdir(), f"agnn_tunableop_{args.workload}_r{rank}.csv"))
            _tn.enable(True)
            _tn.tuning_enable(True)
            gemm_selection = "hipBLASLt via PyTorch TunableOp (candidates timed during the eager warm-up steps, frozen before capture)"
        except Exception as e:                                   # an older torch: measured as the heuristic picks
            print(f"[bench] TunableOp unavailable ({type(e).__name__}: {e})", file=sys.stderr)

    def _freeze_gemm_selection():
        if gemm_selection.startswith("hipBLASLt via"):
            torch.cuda.tunable.tuning_enable(False)              # keep what was found; never tune inside a capture

    if not args.no_graph:
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(3):
                    fwd_bwd(); reduce_between(bwd_tail); update()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            _freeze_gemm_selection()
            dp.barrier_and_sync()                               # no collective in flight on any rank while capturing
            g2 = torch.cuda.CUDAGraph()
            dot = args.graph_dot                                 # the captured step's dependency graph as DOT (debugging)
            # Which of the two backward schedules of the hybrid encoders replays faster depends on how many nodes each
            # branch of the captured graph has (profiles/r02_step_timeline.md: c2s 3.39 vs 3.45 ms, c2 3.61 vs 3.54 ms), so
            # both are captured and the faster one (40 replays each, slowest rank decides) is kept — TunableOp's way.  Same for
            # where the sequence branch's deferred weight-gradient products run (linear.ITEMS_HOME: c2s 3.19 vs 3.22 ms with
            # them in the main chain's flush, c2d 3.38 vs 3.35 — round 3, alternating pairs on one box).
            from analysisgnn_amd import encoders as _enc
            hybrid = enc in ("hybridgnn", "hgt") and not args.no_defer and not dot
            lates = [True, False] if args.schedule == "auto" and hybrid else [{"late": True, "plain": False, "auto": _enc.LATE_SEQUENCE_BACKWARD}[args.schedule]]
            homes = [True, False] if args.items_home == "auto" and hybrid else [_lin.ITEMS_HOME]
            # ... and for how much of the main flush point's weight-gradient work runs there: since the recurrence kernels got
            # shorter (round 3) the main chain's flush outlasts the sequence branch, whose stream then has room for the rest
            keeps = [1.0, 0.75, 0.5] if args.flush_keep is None and hybrid else [_lin.FLUSH_KEEP]
            variants = [(a, b, k) for a in lates for b in homes for k in keeps]
            best = None
            # every captured variant stays alive to the end of the run: letting the losers be destroyed (hipGraphExecDestroy + their
            # private memory pools) crashed a LATER replay of the kept graph about every second run when a process group exists
            # (capture_error_mode=thread_local, NCCL's watchdog thread alive; scripts/nccl_capture_check.py: 0 of 6 runs since)
            keep_alive = []
            for late, home, keep in variants:
                _enc.LATE_SEQUENCE_BACKWARD, _lin.ITEMS_HOME, _lin.FLUSH_KEEP = late, home, keep
                g1 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g1, capture_error_mode=cap_mode):
                    loss_v = fwd_bwd()
                    if dot:
                        _dump_capture_dot(dot, dev)
                gt = None
                if buckets:                                      # the tail belongs to THIS capture's autograd graph (model._cut)
                    gt = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gt, capture_error_mode=cap_mode):
                        bwd_tail()
                t_v = 0.0
                if len(variants) > 1:
                    for _ in range(5):
                        g1.replay()
                    dp.barrier_and_sync()
                    t0 = time.perf_counter()
                    for _ in range(40):
                        g1.replay()
                    dp.barrier_and_sync()
                    t_v = dp.max_over_ranks(time.perf_counter() - t0)
                keep_alive.append((g1, gt))
                if best is None or t_v < best[0]:
                    best = (t_v, g1, loss_v, late, gt, home, keep)
            _, g1, loss_ref[0], schedule_late, g_tail, schedule_home, schedule_keep = best
            _enc.LATE_SEQUENCE_BACKWARD, _lin.ITEMS_HOME, _lin.FLUSH_KEEP = schedule_late, schedule_home, schedule_keep
            if rank == 0 and len(variants) > 1:
                print(f"[bench] backward schedule: sequence branch {'behind a late node' if schedule_late else 'in autograd order'}, "
                      f"its deferred products {'with the main flush' if schedule_home else 'on its own flush'}, "
                      f"{schedule_keep:g} of the main flush's products there", file=sys.stderr)
            with torch.cuda.graph(g2, capture_error_mode=cap_mode):
                update()
            graphs = (g1, g2, g_tail)
            graph_mode = (f"hipGraph replay (capture_error_mode={cap_mode}; sequence branch {'late node' if schedule_late else 'autograd order'}, "
                          f"its deferred products {'in the main flush' if schedule_home else 'in its own flush'}, "
                          f"{schedule_keep:g} of the main flush's weight-gradient FLOPs there)")
        except Exception as e:                                  # capture refused: run eagerly — a HOST-BOUND number, flagged at top level
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graphs = None
            graph_mode = f"eager-fallback: {type(e).__name__}: {str(e)[:200]}"
            torch.cuda.synchronize(dev)

    def step():
        if sampler is not None:                                     # the loader's only host-side decision: which windows (32 int32 H2D)
            sampler.set_windows(store.random_windows(N_SUB, N_NOTES, win_rng))
        if graphs is not None:
            if HOST_T is not None:
                h0 = time.perf_counter()
                graphs[0].replay()
                h1 = time.perf_counter()
                reduce_between(graphs[2].replay if buckets else None)
                graphs[1].replay()
                HOST_T.append((h1 - h0, time.perf_counter() - h1))
                return loss_ref[0]
            graphs[0].replay()
            reduce_between(graphs[2].replay if buckets else None)
            graphs[1].replay()
            return loss_ref[0]
        loss = fwd_bwd()
        reduce_between(bwd_tail)
        if os.environ.get("AGNN_BENCH_TRACE"):                     # debugging (eager runs): the loss and any non-finite gradient, per step
            torch.cuda.synchronize(dev)
            bad = [n for n, p in trainable.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
            print(f"[bench] loss {float(loss.detach()):.5f} non-finite gradients: {bad[:6]}", file=sys.stderr)
            if bad:
                raise SystemExit(f"[bench] step with non-finite gradients: {bad}")
        update()
        return loss

    HOST_T = [] if args.host_times else None           # debugging: host time inside the two replay calls of a step
    for _ in range(args.warmup):
        step()
    dp.barrier_and_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    dp.barrier_and_sync()
    dt = time.perf_counter() - t0
    dt = dp.max_over_ranks(dt)
    if HOST_T:
        a = sorted(t[0] for t in HOST_T[args.warmup:])
        b = sorted(t[1] for t in HOST_T[args.warmup:])
        print(f"[bench] host time in replay calls: step graph median {a[len(a) // 2] * 1e6:.0f} us (max {a[-1] * 1e6:.0f}), "
              f"update graph median {b[len(b) // 2] * 1e6:.0f} us; wall per step {dt / args.steps * 1e6:.0f} us", file=sys.stderr)
    if _agnn_lib.STAMPS["on"] and _agnn_lib.STAMPS["buf"] is not None and rank == 0:
        # the LAST replay's stamps (100 MHz device counter), relative to the step's first one
        v = _agnn_lib.STAMPS["buf"].cpu().tolist()
        names = _agnn_lib.STAMPS["names"]
        t_first = min(v[k] for k in range(len(names)) if v[k])
        print(f"[bench] in-graph time stamps of the last replay (us since '{names[0]}'; wall per step {dt / args.steps * 1e6:.0f} us):", file=sys.stderr)
        for k in sorted(range(len(names)), key=lambda k: v[k]):
            print(f"[bench]   {(v[k] - t_first) / 100.0:9.1f}  {names[k]}", file=sys.stderr)
    assert torch.isfinite(loss).item(), "loss diverged"
    from analysisgnn_amd import _lib
    _lib.check_device_status(dev)                      # no CSR build of the run flagged an inconsistent index (outside the timed region)

    if sampler is not None:                                         # the roofline launch runs on the LAST sampled batch
        g.edge_index = {et: e.cpu().numpy() for et, e in I["edge_index_dict"].items()}
        g.num_nodes = {t: int(v.shape[0]) for t, v in I["x_dict"].items()}
    roof = roofline(args.workload, g, I, hid, layers, dev) if rank == 0 else None
    if rank == 0:
        nodes = I["batch_size"] * world * args.steps
        e_tot = sum(int((e[0] >= 0).sum()) for e in g.edge_index.values())
        out = {
            "metric": METRICS[args.workload], "value": nodes / dt, "unit": "subgraph-nodes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # how the step's launches were issued: "hipGraph replay ..." or "eager-fallback: <why the capture failed>" (then the
            # figure is bound by the host's launch rate, not by the GPU — do not read it as the step's speed)
            "graph": graph_mode,
            "gemm_selection": gemm_selection,
            "config": {"workload": WORKLOADS[args.workload] + f"; {g.num_nodes['note']} notes, {e_tot} edges per GPU; train step = fwd + "
                                   f"objective ({args.mt_strategy}) + bwd + allreduce + clip + AdamW, CSR rebuilt every step; "
                                   + ("hipGraph replay" if graphs is not None else "EAGER launches (see \"graph\")"),
                       "workload_id": args.workload, "per_gpu_subgraphs": N_SUB, "target_notes_per_subgraph": N_NOTES,
                       "objective": args.mt_strategy,
                       "optimizer": f"clip 1.0 + AdamW(lr {args.lr:g}: the reference's 500-step linear warm-up to 5e-3 at step 50; weight decay 5e-3)",
                       **({"sampler": {"capacity_per_subgraph": sampler.cap, "pool": sampler.pool, "rows": sampler.num_nodes,
                                       "sources_dropped_by_capacity_or_pool": sampler.dropped()}} if sampler is not None else {}),
                       "sharding": ("every rank draws its own windows from the replicated corpus" if sampler is not None
                                    else "rank r takes subgraphs {i : i mod G = r}"),
                       "parallelism": f"dp{world}",
                       "allreduce": ("none (one rank)" if world == 1 else
                                     f"2 buckets: {flat.cut} floats shipped asynchronously beside the input layers' backward, then "
                                     f"{flat.flat.numel() - flat.cut}; SUM / world" if buckets else "one message between the two graph replays")},
            "roofline": roof,
        }
        # whole-step efficiency as a reported number: algorithmic matrix FLOPs of the step / step time / fp32-MFMA peak
        fl, parts = algorithmic_flops(enc, I, hid, layers, tasks)
        out["step_flops_alg"] = fl
        out["mfma_frac"] = fl / (dt / args.steps) / MFMA_F32_PEAK
        out["step_flops_breakdown"] = {k: round(v / 1e9, 3) for k, v in parts.items()}     # GFLOP per step (fwd + bwd)
        if world == 1 and args.workload == "c2s" and not args.no_other:
            # secondary figures, each a child process with the same steps: "c2" = continuity with round 1's line (whole graphs,
            # nothing trimmed); "c2d" = the same training step with the batch sampled and gathered ON THE DEVICE inside it
            out["other_workloads"] = {}
            for wl in ("c2", "c2d", "c3", "c3d", "c5"):
                r = None
                try:
                    cmd = [sys.executable, "-X", "faulthandler", os.path.abspath(__file__), "--workload", wl, "--no-cpu-baseline", "--steps",
                           str(args.steps), "--warmup", str(args.warmup), "--mt-strategy", args.mt_strategy]
                    if args.no_tune_gemm:
                        cmd.append("--no-tune-gemm")
                    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
                    o = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
                    out["other_workloads"][wl] = {"metric": o["metric"], "value": o["value"], "ms_per_step": o["ms_per_step"],
                                                  "workload": o["config"]["workload"], "graph": o["graph"], "roofline": o["roofline"],
                                                  "step_flops_alg": o["step_flops_alg"], "mfma_frac": o["mfma_frac"]}
                except Exception as e:                              # secondary figures only; say why (the child's last lines)
                    why = " | ".join((r.stderr or "").strip().splitlines()[-6:]) if r is not None else ""
                    out["other_workloads"][wl] = f"not measured ({type(e).__name__}; exit code {getattr(r, 'returncode', None)}; {why[-600:]})"
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, args.mt_strategy)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


def roofline(workload, g, I, hid, layers, dev):
    """Live timing of the workload's dominant hand-written aggregation kernel.  Inside the timed region the launches are
    replayed from a hipGraph, where single kernels cannot be bracketed by events; and an event pair around ONE eager launch
    mostly measures the ~8 us of event / dispatch overhead.  So the same launch (same CSR, same shapes as in the step) is
    captured REP times back to back into a small graph and HIP events bracket each replay on the launch stream: average
    launch duration = replay time / REP (kernel + the ~1.5 us kernel-to-kernel boundary).
    ALGORITHMIC bytes: B_alg = sum_r [4 (N_dst + 1) + 4 E_r] + 4 H (N_src_unique + R N_dst) per launch."""
    import numpy as np
    import torch
    from analysisgnn_amd import ops
    from analysisgnn_amd.encoders import TrimPlan
    from analysisgnn_amd.graph import HeteroIndex
    REP = 10

    def timed(fn):
        fn(); torch.cuda.synchronize(dev)
        run, per = fn, 1
        try:
            sgraph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(sgraph):
                for _ in range(REP):
                    fn()
            run, per = sgraph.replay, REP
        except Exception:
            run, per = fn, 1
        ts = []
        for _ in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e-3 / per)
        ts = ts[2:]
        return sum(ts) / len(ts), len(ts) * per

    if workload in ("c3", "c3d"):
        from analysisgnn_amd.hgt import attention_roofline_case
        return attention_roofline_case(g, I, hid, dev, timed, HBM_PEAK)

    n_nodes = {k: int(v.shape[0]) for k, v in I["x_dict"].items()}
    hix = HeteroIndex(I["edge_index_dict"], n_nodes)
    ets = [et for et in hix.edge_types if et[0] == "note" and et[2] == "note"]
    plan = TrimPlan(layers, I["x_dict"], I["edge_index_dict"], I["neighbor_mask_node"], I["neighbor_mask_edge"])
    layer = 1 if workload in ("c2s", "c2d") else 0         # sampled batches: the first TRIMMED layer (rowend path), otherwise layer 0
    hix.prepare_trim(plan.e_keep)
    n_dst = plan.n_keep[layer]["note"]
    e_keep = [plan.e_keep[layer][et] for et in ets]
    spec = ops.AggSpec(fwd=[hix.fwd[e] for e in ets], bwd=[hix.bwd[e] for e in ets], src_id=[0] * len(ets), n_rows=n_dst, mean=True,
                       shared_slot=False, e_limit=e_keep if any(k is not None for k in e_keep) else None)
    n_src = plan.n_keep[layer - 1]["note"] if layer > 0 else n_nodes["note"]
    xs = torch.randn(n_src, hid, device=dev)
    with torch.no_grad():
        t_k, launches = timed(lambda: ops.aggregate(spec, [xs]))
    # algorithmic bytes of exactly this launch
    idx_bytes, e_kept, srcs = 0, 0, []
    for et, k in zip(ets, e_keep):
        ei = g.edge_index[et][:, :k] if k is not None else g.edge_index[et]
        ei = ei[:, (ei[1] < n_dst) & (ei[0] >= 0)]         # (-1, -1): padding slots of a device-sampled batch
        idx_bytes += 4 * (n_dst + 1) + 4 * ei.shape[1]
        e_kept += int(ei.shape[1])
        srcs.append(ei[0])
    n_src_unique = int(np.unique(np.concatenate(srcs)).size)
    b_alg = idx_bytes + 4 * hid * (n_src_unique + len(ets) * n_dst)
    trimmed = any(k is not None for k in e_keep)
    ch = hid // 256
    return {"bound": "hbm",
            "kernel": f"k_spmm_fast7<{ch},false,false,false,false> forward hetero-SpMM, layer {layer} of the step "
                      f"({'trimmed: row ends from rowend' if trimmed else 'untrimmed'}; R={len(ets)}, N_dst={n_dst}, E={e_kept}, H={hid} -> [N,{len(ets)}H])",
            "achieved": b_alg / t_k / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": (b_alg / t_k) / HBM_PEAK,
            # HBM bytes per launch come from separate rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), not
            # from this run: see the file named below; null here rather than a constant
            "traffic": None, "traffic_source": "profiles/r02_spmm_pmc.md (rocprofv3 --pmc passes of this launch)",
            "alg_bytes_per_launch": b_alg, "n_src_unique": n_src_unique, "avg_us": t_k * 1e6, "launches": launches,
            "timing": "HIP events around hipGraph replays of 10 back-to-back launches"}


if __name__ == "__main__":
    main()
