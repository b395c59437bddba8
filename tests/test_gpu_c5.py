"""BASELINE config C5 exercised AT SIZE (MetricalGNN L=4 H=512, heads cadence / localkey / romanNumeral;
reference models/analysis.py:463-473): the H=512 instantiations of the aggregation kernel on a full per-GPU batch, the
weight-gradient kernel / library switch at the 512 x 2048 projection, and the whole model (forward logits and every
gradient) on four sampled 500-note subgraphs against the CPU restatement.  fp32, 1e-4 relative to max(1, |ref|max)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import assert_close  # noqa: E402
from oracle import c_oracle  # noqa: E402
from test_gpu_encoders import _cmp_grads, _cpu_params  # noqa: E402

DEV = "cuda:0"
C5_TASKS = {"cadence": 4, "localkey": 50, "romanNumeral": 185}


def test_spmm_h512_full_batch_against_c_oracle():
    """32 x 500 notes, 4 relations, H = 512 (2 KiB rows): forward vs the C oracle, backward by the adjoint identity."""
    from analysisgnn_amd import ops
    from analysisgnn_amd.graph import HeteroIndex
    from analysisgnn_amd.synth import make_batch
    b = make_batch(32, 500)
    N = b.num_nodes["note"]
    eid = {et: torch.from_numpy(e).to(DEV) for et, e in b.edge_index.items()}
    hix = HeteroIndex(eid, {"note": N})
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, 512, generator=g)
    xd = x.to(DEV).requires_grad_(True)
    ets = list(eid.keys())
    spec = ops.AggSpec(fwd=[hix.fwd[et] for et in ets], bwd=[hix.bwd[et] for et in ets], src_id=[0] * 4, n_rows=N, mean=True,
                       shared_slot=False)
    out = ops.aggregate(spec, [xd])
    segs = [dict(row=b.edge_index[et][1], col=b.edge_index[et][0], n_rows=N) for et in ets]
    rs, c, p, _ = c_oracle.csr_build(segs)
    exp = c_oracle.spmm([dict(src=x.numpy(), rowptr=rs[r * N:(r + 1) * N + 1], col=c) for r in range(4)], N, 512, 512)
    np.testing.assert_allclose(out.detach().cpu().numpy(), exp, rtol=1e-6, atol=1e-6)
    gout = torch.randn(N, 2048, generator=g).to(DEV)
    out.backward(gout)
    lhs = float((out.detach().double() * gout.double()).sum())
    rhs = float((xd.detach().double() * xd.grad.double()).sum())
    # <A x, g> == <x, A^T g> up to fp32 rounding of the 33 M products: measured against the size of the terms, not of their sum
    scale = float(out.detach().double().norm() * gout.double().norm())
    assert abs(lhs - rhs) <= 1e-6 * scale, (lhs, rhs, scale)


@pytest.mark.parametrize("out_f,in_f", [(512, 2048), (512, 1024), (512, 512), (185, 64)])
def test_weight_gradient_at_c5_projection_sizes(out_f, in_f):
    """dW = dY^T X, db at N = 16 000: 512 x 2048 (the [N, 4H] x [4H, H] SAGE projection of C5) exceeds the kernel's
    output-size switch (linear.MAX_OUT_IN) and goes to the library; 512 x 1024 is the largest the kernel takes."""
    from analysisgnn_amd import linear
    g = torch.Generator().manual_seed(out_f + in_f)
    n = 16000
    dy = torch.randn(n, out_f, generator=g)
    x = torch.randn(n, in_f, generator=g)
    ref_w = (dy.double().t() @ x.double()).float()
    ref_b = dy.double().sum(0).float()
    dw, db = linear.weight_grad(dy.to(DEV), x.to(DEV), True)
    assert (out_f * in_f <= linear.MAX_OUT_IN) == (out_f * in_f <= 512 * 1024)
    assert_close(dw, ref_w, 1e-4, "dW")
    assert_close(db, ref_b, 1e-4, "db")


def _oracle_run(P, g, I, L, dtype, keep_graph=False):
    from oracle import encoders_ref as E
    from oracle.testing import ReluTap
    Pd = {k: (v.detach().to(dtype).requires_grad_(True) if v.is_floating_point() else v) for k, v in P.items()}
    xd = {k: v.to(dtype) for k, v in I["x_dict"].items()}
    with ReluTap(keep_graph=keep_graph) as tap:
        x = E.analysis_encode(Pd, "metricalgnn", g.metadata(), L, I["pitch_spelling"], I["key_signature"], xd, I["edge_index_dict"],
                              I["batch_dict"], I["batch_size"], I["neighbor_mask_node"], I["neighbor_mask_edge"])
        ref = E.analysis_logits(Pd, x, list(C5_TASKS))
    loss = sum((v ** 2).mean() for v in ref.values())
    return ref, Pd, tap, loss


@pytest.mark.parametrize("seed", [7, 1, 2, 3])
def test_c5_model_on_sampled_subgraphs_matches_float64_cpu_path(seed):
    """TorchAnalysisGNN(MetricalGNN, L=4, H=512, 3 heads) on four neighbour-sampled 500-note subgraphs ([5,5,5] hops, the
    per-hop counts passed: every layer trimmed): forward logits and all parameter gradients against a FLOAT64 run of
    oracle/encoders_ref.py, over four weight seeds (round 2 ran ONE seed, picked because no ReLU input sat within fp32
    rounding of zero — VERDICT r2 weak #2).
    A ReLU input that close to the kink gets its derivative from whichever side the fp32 rounding of THIS evaluation order
    puts it on; one flipped derivative moves a whole row of an activation gradient and every weight gradient upstream by
    1e-4 ... 1e-3 relative (profiles/r03_parity_notes.md).  Here that is ATTRIBUTED, not avoided and not tolerated blindly:
      * logits: always within 1e-4 of float64;
      * gradients: within 1e-4 * max(1, |ref|max) of float64 — directly, or after removing the exactly computed effect of
        flipped ReLU derivatives (oracle.testing.explain_by_relu_flips: one float64 backward pass per ReLU input near the
        kink gives the direction its flip moves the gradients in; the deviation over ALL parameters at once must be a 0/1
        combination of those directions, every residual back inside the 1e-4 bound).  The flips found are printed."""
    from analysisgnn_amd.models import TorchAnalysisGNN
    from analysisgnn_amd.synth import make_sampled_batch, torch_inputs
    from oracle.testing import explain_by_relu_flips
    g = make_sampled_batch(4, 500, (5, 5, 5), first_seed=40)
    H, L = 512, 4
    torch.manual_seed(seed)
    m = TorchAnalysisGNN(g.metadata(), in_channels=25, hidden_channels=H, out_channels=128, task_dict=C5_TASKS, num_layers=L,
                         dropout=0.0, use_jk=False, logit_fusion=False, encoder_type="metricalgnn").train()
    P = _cpu_params(m)
    m = m.to(DEV)
    I = torch_inputs(g, in_channels=25, seed=5)
    assert I["batch_size"] == 2000 and len(I["neighbor_mask_node"]["note"]) == 4
    ref64, P64, tap64, loss64 = _oracle_run(P, g, I, L, torch.float64, keep_graph=True)
    J = {k: ({kk: vv.to(DEV) for kk, vv in v.items()} if isinstance(v, dict) and v and isinstance(next(iter(v.values())), torch.Tensor)
             else (v.to(DEV) if isinstance(v, torch.Tensor) else v)) for k, v in I.items()}
    out = m(J["pitch_spelling"], J["key_signature"], J["x_dict"], J["edge_index_dict"], J["batch_dict"], J["batch_size"],
            J["neighbor_mask_node"], J["neighbor_mask_edge"])
    for t in C5_TASKS:
        assert out[t].shape == (2000, C5_TASKS[t])
        assert_close(out[t], ref64[t], 1e-4, f"logits[{t}]")
    sum((v ** 2).mean() for v in out.values()).backward()
    names = [k for k, _ in m.named_parameters()]
    hip = [p.grad if p.grad is not None else torch.zeros_like(p) for _, p in m.named_parameters()]
    plist = [P64[k] for k in names]
    g64 = torch.autograd.grad(loss64, plist, retain_graph=True, allow_unused=True)
    bound = lambda ref: 1e-4 * max(1.0, float(ref.abs().max()))                      # noqa: E731
    beyond = []
    for k, gh, gr in zip(names, hip, g64):
        if gr is None:
            assert float(gh.abs().max()) == 0.0, k
            continue
        err = float((gh.detach().cpu().double() - gr).abs().max())
        if err > bound(gr):
            beyond.append((k, f"{err:.1e}"))
    n_relu = sum(x.numel() for x in tap64.inputs)
    if not beyond:
        print(f"[c5 seed {seed}] all {sum(t is not None for t in g64)} gradient tensors within 1e-4 of float64 "
              f"({n_relu} ReLU inputs, {tap64.at_risk(3e-6)} within 3e-6 of the kink, none flipped on the HIP path)")
        return
    rep = explain_by_relu_flips(tap64, loss64, plist, hip, bound)
    print(f"[c5 seed {seed}] {len(beyond)} gradient tensors beyond 1e-4 of float64 (e.g. {beyond[:3]}); {n_relu} ReLU inputs, "
          f"{rep['n_candidates']} candidates near the kink examined, flipped on the HIP path: {rep['flips']} "
          f"(fitted coefficients other than 0: {[c for c in rep['coeffs'] if abs(c) > 0.02]}); worst residual after removing them: "
          f"{rep['worst'][1]:.1e} (bound {rep['worst'][2]:.1e}) in {names[rep['worst'][0]]}")
    assert rep["ok"], f"gradient deviation is not explained by flipped ReLU derivatives: {rep['worst']}, coefficients {rep['coeffs']}"
    assert 0 < len(rep["flips"]) <= 8, rep["flips"]
