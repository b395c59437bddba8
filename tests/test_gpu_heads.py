"""Fused task heads and fused multi-task cross entropy vs plain PyTorch fp32 on CPU (the per-task modules and
`F.cross_entropy(ignore_index=-1, label_smoothing=0.1)` the reference uses, analysis.py:486-496, :881-888).
Tolerance 1e-4 relative to max(1,|ref|max); observed ~1e-6."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from helpers import assert_close  # noqa: E402

DEV = "cuda:0"
TASKS = {"cadence": 4, "localkey": 50, "hrythm": 2, "romanNumeral": 185, "pcset": 94}


def _clf(o=32):
    import torch.nn as nn
    torch.manual_seed(0)
    return nn.ModuleDict({t: nn.Sequential(nn.Linear(o, o // 2), nn.ReLU(), nn.LayerNorm(o // 2), nn.Linear(o // 2, c))
                          for t, c in TASKS.items()})


def test_fused_heads_and_loss_match_per_task_modules():
    import copy
    from analysisgnn_amd.heads import fused_head_logits, multitask_cross_entropy
    N = 301
    clf = _clf()
    with torch.no_grad():
        for m in clf.values():
            m[2].weight.uniform_(0.5, 1.5)
            m[2].bias.uniform_(-0.3, 0.3)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, 32, generator=g)
    labels = {t: torch.randint(0, c, (N,), generator=g) for t, c in TASKS.items()}
    labels["localkey"][::7] = -1                         # ignored rows
    labels["hrythm"][:] = -1                             # a task with no valid row at all
    # reference: per-task modules + F.cross_entropy on CPU
    xr = x.clone().requires_grad_(True)
    ref_losses = []
    for t in TASKS:
        ref_losses.append(F.cross_entropy(clf[t](xr), labels[t], ignore_index=-1, label_smoothing=0.1))
    ref_losses = torch.stack([torch.nan_to_num(l, nan=0.0) for l in ref_losses])
    w = torch.tensor([1.0, 0.5, 2.0, 1.5, 0.25])
    (ref_losses * w).sum().backward()
    # fused on the GPU
    clg = copy.deepcopy(clf).to(DEV)
    for p in clg.parameters():
        p.grad = None
    xg = x.to(DEV).requires_grad_(True)
    logits, offs = fused_head_logits(clg, xg, list(TASKS))
    assert logits.shape == (N, sum(TASKS.values())) and offs[-1] == sum(TASKS.values())
    for i, t in enumerate(TASKS):
        assert_close(logits[:, offs[i]:offs[i + 1]], clf[t](x), 1e-4, f"logits[{t}]")
    lab = torch.stack([labels[t] for t in TASKS]).to(DEV)
    losses = multitask_cross_entropy(logits, offs, lab, 0.1, -1)
    assert_close(losses, ref_losses, 1e-4, "losses")
    (losses * w.to(DEV)).sum().backward()
    assert_close(xg.grad, xr.grad, 1e-4, "dx")
    for (n, p), (_, q) in zip(clg.named_parameters(), clf.named_parameters()):
        if q.grad is None:
            continue
        assert_close(p.grad, q.grad, 1e-4, f"d{n}")


def test_model_forward_clf_is_dict_of_views():
    from analysisgnn_amd.models import TorchAnalysisGNN
    from analysisgnn_amd.synth import make_batch
    g = make_batch(1, 30)
    m = TorchAnalysisGNN(g.metadata(), 25, 32, 16, {"a": 3, "b": 7}, 2, dropout=0.0, use_jk=False, logit_fusion=False).to(DEV)
    x = torch.randn(11, 16, device=DEV)
    out = m.forward_clf(x)
    assert list(out) == ["a", "b"] and out["a"].shape == (11, 3) and out["b"].shape == (11, 7)
    assert_close(out["b"], m.clf_dict["b"](x), 1e-5)
    assert list(m.forward_clf(x, tasks=["b"])) == ["b"]


@pytest.mark.parametrize("K,classes,N", [(64, [2, 185, 33, 32, 1, 64, 7], 1003), (32, [5, 40], 70), (128, [3, 97, 31], 257),
                                         (64, [12] * 21, 4100),
                                         (64, [4, 50, 50, 15, 4, 38, 38, 22, 22, 2, 94, 185, 2, 2, 2, 2, 2, 2, 45, 49, 4], 16001),
                                         (64, [3, 400, 17, 16, 5], 130),        # a four-group chunk wider than the LDS logits image
                                         (64, [1] * 32, 65), (64, [7], 5), (64, [40] * 31, 200),
                                         (64, [5] * 17, 97), (64, [9, 30, 2] * 6, 20011), (64, [100] * 9, 300), (64, [33, 1] * 3, 31),
                                         (64, [6, 17, 40, 3, 30], 70001)])
def test_grouped_projection_matches_per_task_linear(K, classes, N):
    """agnn_gproj_* against T separate nn.Linear(K, C_t) evaluated in float64 (forward, da, dw, db).  K = 64 takes the
    persistent whole-row forward kernel (32-row blocks, one workgroup per CU, ceil(G / 16) equal chunks of groups through
    LDS): the 21 heads at a row count that is not a multiple of 32 (two blocks per workgroup), 32 groups of one class (four
    256-column quarters per chunk), a single group, fewer rows than one row tile, 31 x 40 classes (48 tiles in a chunk: six
    tile slots per wave), 17 groups (chunks of 9 + 8) and three row blocks per workgroup.  The input gradient takes the
    persistent whole-row kernel when the class total is even (8-byte row loads) and the padded image fits (the other cases
    keep the one-wave-per-tile kernel): 900 classes are eight 128-column pieces per row, [33, 1] * 3 has groups of one step
    and of three with a padded tail, 31 rows are less than one block; 70 001 rows are nine row blocks per workgroup."""
    from analysisgnn_amd.heads import grouped_projection
    torch.manual_seed(0)
    G = len(classes)
    offs = [0]
    for c in classes:
        offs.append(offs[-1] + c)
    a = torch.randn(N, G * K)
    w = torch.randn(offs[-1], K) * 0.2
    b = torch.randn(offs[-1])
    gout = torch.randn(N, offs[-1])
    a64, w64, b64 = (t.double().requires_grad_(True) for t in (a, w, b))
    ref = torch.cat([a64[:, g * K:(g + 1) * K] @ w64[offs[g]:offs[g + 1]].t() + b64[offs[g]:offs[g + 1]] for g in range(G)], dim=1)
    ref.backward(gout.double())
    ag, wg, bg = (t.to(DEV).requires_grad_(True) for t in (a, w, b))
    out = grouped_projection(ag, wg, bg, offs, K)
    out.backward(gout.to(DEV))
    assert_close(out, ref.float(), 1e-5, "out")
    assert_close(ag.grad, a64.grad.float(), 1e-5, "da")
    assert_close(wg.grad, w64.grad.float(), 1e-5, "dw")
    assert_close(bg.grad, b64.grad.float(), 1e-5, "db")


def test_grouped_projection_on_strided_operands():
    """The persistent forward / input-gradient kernels take row strides from the caller: `a` as a column window of a wider
    buffer (16-byte aligned rows, as the C-ABI asks), the incoming gradient as a window that starts 4 bytes into rows of an
    odd stride (the dx movers load single floats: no alignment asked), N not a multiple of the 32-row block."""
    from analysisgnn_amd.heads import grouped_projection
    torch.manual_seed(1)
    classes, K, N = [4, 50, 50, 15, 4, 38, 38, 22, 22, 2, 94, 185, 2, 2, 2, 2, 2, 2, 45, 49, 4], 64, 8237
    G = len(classes)
    offs = [0]
    for c in classes:
        offs.append(offs[-1] + c)
    big = torch.randn(N, G * K + 8, device=DEV)
    a = big[:, 4:4 + G * K].detach().requires_grad_(True)
    w = (torch.randn(offs[-1], K, device=DEV) * 0.2).requires_grad_(True)
    b = torch.randn(offs[-1], device=DEV).requires_grad_(True)
    wide = torch.randn(N, offs[-1] + 3, device=DEV)
    gout = wide[:, 1:1 + offs[-1]]
    assert not a.is_contiguous() and not gout.is_contiguous()
    out = grouped_projection(a, w, b, offs, K)
    out.backward(gout)
    a64, w64, b64 = (t.detach().double().requires_grad_(True) for t in (a, w, b))
    ref = torch.cat([a64[:, g * K:(g + 1) * K] @ w64[offs[g]:offs[g + 1]].t() + b64[offs[g]:offs[g + 1]] for g in range(G)], dim=1)
    ref.backward(gout.double())
    assert_close(out, ref.float(), 1e-5, "out")
    assert_close(a.grad, a64.grad.float(), 1e-5, "da")
    assert_close(w.grad, w64.grad.float(), 1e-5, "dw")
    assert_close(b.grad, b64.grad.float(), 1e-5, "db")


@pytest.mark.parametrize("wloss", [False, True])
@pytest.mark.parametrize("extra", [(), (5,), (3,)])      # 335 logit columns (scalar rows) / 340 (float4 backward) / 338 (float2)
@pytest.mark.parametrize("N,lam,gscale", [(301, 0.1, 1.0), (16000, 0.1, 1.0), (37, 0.5, 3.0)])
def test_training_loss_matches_torch(N, lam, gscale, extra, wloss):
    """heads.training_loss (agnn_train_loss_f32 / _bwd_f32) against the objective built exactly as the reference builds it
    (models/analysis.py:1034-1036: MultiTaskLoss total / number of tasks; :984, :1072: + lambda * feat.pow(2).mean();
    models/chord.py:39-49: 0.5 / p_i^2 * CE_i + log(1 + p_i^2) when the weights are learned) with F.cross_entropy on the
    CPU; tolerance 1e-4 relative (fp32 sums in a different order).  Called twice: the ticket workspace must be left
    clean, and the result must be bitwise reproducible."""
    from analysisgnn_amd.heads import training_loss
    g = torch.Generator().manual_seed(N)
    C = list(TASKS.values()) + list(extra)
    T = len(C)
    offs = [0]
    for c in C:
        offs.append(offs[-1] + c)
    logits = torch.randn(N, offs[-1], generator=g) * 2
    feat = torch.randn(N, 128, generator=g)
    labels = torch.stack([torch.randint(0, c, (N,), generator=g) for c in C])
    labels[1, ::5] = -1
    labels[3, :] = -1                                    # a task with no valid row (0 here; torch gives NaN: documented)
    params = (0.5 + torch.rand(T, generator=g) * 1.5) if wloss else None
    lr, fr = logits.clone().requires_grad_(True), feat.clone().requires_grad_(True)
    pr = params.clone().requires_grad_(True) if wloss else None
    per = torch.stack([torch.nan_to_num(F.cross_entropy(lr[:, offs[i]:offs[i + 1]], labels[i], ignore_index=-1, label_smoothing=0.1), nan=0.0)
                       for i in range(T)])
    loss_sum = 0
    for i in range(T):                                   # models/chord.py:41-46
        loss_sum = loss_sum + ((0.5 / (pr[i] ** 2) * per[i] + torch.log(1 + pr[i] ** 2)) if wloss else per[i])
    ref = loss_sum / T + lam * fr.pow(2).mean()
    (ref * gscale).backward()
    lg, fg = logits.to(DEV).requires_grad_(True), feat.to(DEV).requires_grad_(True)
    pg = params.to(DEV).requires_grad_(True) if wloss else None
    outs = []
    for _ in range(2):
        lg.grad = fg.grad = None
        if wloss:
            pg.grad = None
        total, per_task = training_loss(lg, offs, labels.to(DEV), fg, lam, 0.1, -1, task_params=pg)
        (total * gscale).backward()
        outs.append((total.detach().clone(), lg.grad.clone(), fg.grad.clone()) + ((pg.grad.clone(),) if wloss else ()))
    assert_close(outs[0][0], ref.detach(), 1e-4, "total")
    assert_close(per_task, per.detach(), 1e-4, "per-task losses")
    assert_close(outs[0][1], lr.grad, 1e-4, "dlogits")
    assert_close(outs[0][2], fr.grad, 1e-4, "dfeat")
    if wloss:
        assert_close(outs[0][3], pr.grad, 1e-4, "dparams")
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("wloss", [False, True])
def test_training_loss_unit_gradient_launches_nothing_backward_and_matches_both_other_paths(wloss):
    """The forward launches leave the FINISHED gradients (agnn_train_loss_final_f32); `backward(gradient=heads.unit_gradient(dev))`
    hands them on without a launch.  Bitwise equal to the general path (a fresh ones tensor: three multiplies by 1.0) and to
    the round-2 flow (unscaled gradient + agnn_train_loss_bwd_f32: the same two-factor products), and no kernel runs in the
    backward pass (torch profiler)."""
    from analysisgnn_amd import heads
    from analysisgnn_amd.heads import training_loss, unit_gradient
    g = torch.Generator().manual_seed(3)
    C = list(TASKS.values())
    T = len(C)
    offs = [0]
    for c in C:
        offs.append(offs[-1] + c)
    N = 2000
    logits = (torch.randn(N, offs[-1], generator=g) * 2).to(DEV)
    feat = torch.randn(N, 128, generator=g).to(DEV)
    labels = torch.stack([torch.randint(0, c, (N,), generator=g) for c in C])
    labels[2, ::3] = -1
    labels = labels.to(DEV)
    params = (0.5 + torch.rand(T, generator=g) * 1.5).to(DEV) if wloss else None

    def run(final, unit):
        saved = heads.FINAL_GRADIENTS
        heads.FINAL_GRADIENTS = final
        try:
            lg, fg = logits.clone().requires_grad_(True), feat.clone().requires_grad_(True)
            pg = params.clone().requires_grad_(True) if wloss else None
            total, _ = training_loss(lg, offs, labels, fg, 0.1, 0.1, -1, task_params=pg)
            torch.cuda.synchronize()
            with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
                total.backward(gradient=unit_gradient(DEV)) if unit else total.backward()
                torch.cuda.synchronize()
            kernels = [e.key for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA]
            return total.detach(), lg.grad, fg.grad, (pg.grad if wloss else None), kernels
        finally:
            heads.FINAL_GRADIENTS = saved
    fast = run(True, True)
    general = run(True, False)
    old = run(False, True)
    assert fast[4] == [], f"kernels in the backward pass: {fast[4]}"
    assert general[4] != [] and old[4] != []
    for a, b, c in zip(fast[:4], general[:4], old[:4]):
        if a is not None:
            assert torch.equal(a, b) and torch.equal(a, c)


def test_training_loss_label_checks():
    """Labels of the wrong dtype / shape are refused on the host; a label outside [0, C) that is not ignore_index makes the
    loss NaN instead of being read out of range (torch: device assert)."""
    from analysisgnn_amd import _lib
    from analysisgnn_amd.heads import training_loss, MultiTaskLoss
    offs = [0, 4, 10]
    logits = torch.randn(33, 10, device=DEV)
    feat = torch.randn(33, 8, device=DEV)
    labels = torch.zeros(2, 33, dtype=torch.long, device=DEV)
    with pytest.raises(_lib.AgnnError):
        training_loss(logits, offs, labels.int(), feat)
    with pytest.raises(_lib.AgnnError):
        training_loss(logits, offs, labels.t().contiguous(), feat)
    total, _ = training_loss(logits, offs, labels, feat)
    assert torch.isfinite(total)
    bad = labels.clone()
    bad[0, 5] = 4                                          # task 0 has 4 classes
    total, _ = training_loss(logits, offs, bad, feat)
    assert torch.isnan(total)
    bad[0, 5] = -100
    assert torch.isnan(training_loss(logits, offs, bad, feat)[0])
    # the module form (dict in, dict out: models/chord.py:39-49) agrees with the fused objective
    m = MultiTaskLoss(["a", "b"]).to(DEV)
    with torch.no_grad():
        m.params.copy_(torch.tensor([0.7, 1.3]))
    out = m({"a": logits[:, :4], "b": logits[:, 4:]}, {"a": labels[0], "b": labels[1]})
    total, per = training_loss(logits, offs, labels, feat, 0.0, task_params=m.params, ce_scale=1.0)
    assert_close(out["total"].detach(), total.detach(), 1e-5, "module total")
    assert_close(torch.stack([out["a"], out["b"]]).detach(), per, 1e-6, "module per-task")


@pytest.mark.parametrize("K,classes,N", [(64, [2, 185, 33, 32, 1, 64, 7], 1003), (32, [5, 40], 70), (64, [12] * 21, 4100)])
def test_grouped_in_projection_matches_per_task_linear(K, classes, N):
    """The mirrored grouped projection (group g reads its own C_g columns, writes K: the 21 `clf_proj_layers[task][0]` of the
    logit-fusion path, models/analysis.py:499-505) against per-task float64 matmuls: forward, dx, dw."""
    from analysisgnn_amd.heads import grouped_in_projection
    torch.manual_seed(1)
    G = len(classes)
    offs = [0]
    for c in classes:
        offs.append(offs[-1] + c)
    x = torch.randn(N, offs[-1])
    w = torch.randn(offs[-1], K) * 0.2
    gout = torch.randn(N, G * K)
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = torch.cat([x64[:, offs[g]:offs[g + 1]] @ w64[offs[g]:offs[g + 1]] for g in range(G)], dim=1)
    ref.backward(gout.double())
    xg, wg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    out = grouped_in_projection(xg, wg, offs, K)
    out.backward(gout.to(DEV))
    assert_close(out, ref.float(), 1e-5, "out")
    assert_close(xg.grad, x64.grad.float(), 1e-5, "dx")
    assert_close(wg.grad, w64.grad.float(), 1e-5, "dw")


class _RefCrossTaskTransformer(torch.nn.Module):
    """The reference's module verbatim in behaviour (models/analysis.py:408-418): nn.MultiheadAttention + residual + LayerNorm."""

    def __init__(self, proj_dim, num_heads=4, dropout=0.1):
        super().__init__()
        self.multihead_attn = torch.nn.MultiheadAttention(proj_dim, num_heads, dropout=dropout, batch_first=True)
        self.norm = torch.nn.LayerNorm(proj_dim)

    def forward(self, x):
        attended, _ = self.multihead_attn(x, x, x)
        return self.norm(x + attended)


@pytest.mark.parametrize("subset", [None, ["localkey", "cadence", "romanNumeral"]])
def test_logit_fusion_matches_reference_wiring(subset):
    """forward_clf with logit_fusion=True (the reference constructor's default, models/analysis.py:422) against the same
    computation written as the reference writes it (:550-565): per-task Linear -> ReLU -> LayerNorm projections of the raw
    logits, stack, nn.MultiheadAttention over the task tokens + residual + LayerNorm, per-task fusion Linear.  Float64 CPU
    reference with the model's own state_dict; logits and every parameter gradient within 1e-4."""
    from analysisgnn_amd.models import TorchAnalysisGNN
    from analysisgnn_amd.synth import make_batch
    import torch.nn as nn
    tasks = {"cadence": 4, "localkey": 50, "tonkey": 50, "quality": 15, "romanNumeral": 185, "section": 2}
    g = make_batch(1, 30)
    torch.manual_seed(5)
    o = 128
    m = TorchAnalysisGNN(g.metadata(), 25, 32, o, tasks, 2, dropout=0.0, use_jk=False, logit_fusion=True).train()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    # reference-side modules, float64
    clf = nn.ModuleDict({t: nn.Sequential(nn.Linear(o, o // 2), nn.ReLU(), nn.LayerNorm(o // 2), nn.Linear(o // 2, c)) for t, c in tasks.items()})
    proj = nn.ModuleDict({t: nn.Sequential(nn.Linear(c, o // 2), nn.ReLU(), nn.LayerNorm(o // 2)) for t, c in tasks.items()})
    ctt = _RefCrossTaskTransformer(o // 2, 4, 0.0)
    fus = nn.ModuleDict({t: nn.Linear(o // 2, c) for t, c in tasks.items()})
    for name, mod in (("clf_dict", clf), ("clf_proj_layers", proj), ("cross_task_transformer", ctt), ("fusion_layers", fus)):
        mod.load_state_dict({k[len(name) + 1:]: v for k, v in sd.items() if k.startswith(name + ".")})
        mod.double()
    N = 2500
    x = torch.randn(N, o)
    use = list(tasks) if subset is None else subset
    from helpers import assert_grads_close_or_relu_flips
    from oracle.testing import ReluTap
    xr = x.double().requires_grad_(True)
    with ReluTap(keep_graph=True) as tap:
        raw = {t: clf[t](xr) for t in use}                                               # :549
        pl = {t: proj[t](raw[t]) for t in raw}                                           # :552
        names = list(pl)
        enh = ctt(torch.stack([pl[t] for t in names], dim=1))                            # :555-559
        ref = {t: fus[t](enh[:, i]) for i, t in enumerate(names)}                        # :562-565
    gout = {t: torch.randn(N, tasks[t]) for t in use}
    loss64 = sum((ref[t] * gout[t].double()).sum() for t in use)
    m = m.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    out = m.forward_clf(xg, None if subset is None else subset)
    assert list(out) == use
    sum((out[t] * gout[t].to(DEV)).sum() for t in use).backward()
    for t in use:
        assert_close(out[t], ref[t].float(), 1e-4, f"refined logits {t}")
    # gradients: within 1e-4 of float64, directly or after the exactly computed effect of flipped ReLU derivatives is removed
    # (2 x N x T x 64 ReLU inputs; a handful sit within fp32 rounding of the kink for any seed)
    got = dict(m.named_parameters())
    pnames, plist, hip = ["x"], [xr], [xg.grad]
    for name, mod in (("clf_dict", clf), ("clf_proj_layers", proj), ("cross_task_transformer", ctt), ("fusion_layers", fus)):
        for k, p in mod.named_parameters():
            pnames.append(f"{name}.{k}")
            plist.append(p)
            hip.append(got[f"{name}.{k}"].grad)
    assert_grads_close_or_relu_flips(pnames, hip, tap, loss64, plist, 1e-4, f"logit fusion {subset}")


@pytest.mark.parametrize("N", [1, 37, 128, 1003, 16000])
def test_head_block_kernel_matches_float64_and_the_three_launch_path(N):
    """agnn_heads_fwd_f32 (one launch: Linear(128, 64) -> ReLU -> LayerNorm -> Linear(64, C_t) for all tasks) vs the per-task
    modules in float64, and vs the same model on the three-launch path (library GEMM, segmented LayerNorm, grouped projection):
    logits, input gradient and every parameter gradient.  1e-4 relative to max(1, |ref|max) against float64; the two HIP
    paths share the backward kernels and differ by rounding only (2e-5)."""
    import copy
    import analysisgnn_amd.heads as H
    tasks = {"cadence": 4, "localkey": 50, "hrythm": 2, "romanNumeral": 185, "pcset": 94, "root": 38, "section": 2, "quality": 15, "degree1": 22}
    import torch.nn as nn
    torch.manual_seed(3)
    clf = nn.ModuleDict({t: nn.Sequential(nn.Linear(128, 64), nn.ReLU(), nn.LayerNorm(64), nn.Linear(64, c)) for t, c in tasks.items()})
    with torch.no_grad():
        for m in clf.values():
            m[2].weight.uniform_(0.5, 1.5)
            m[2].bias.uniform_(-0.3, 0.3)
    g = torch.Generator().manual_seed(N)
    x = torch.randn(N, 128, generator=g)
    gout = torch.randn(N, sum(tasks.values()), generator=g)
    ref = copy.deepcopy(clf).double()
    xr = x.double().requires_grad_(True)
    want = torch.cat([ref[t](xr) for t in tasks], dim=1)
    (want * gout.double()).sum().backward()
    got = {}
    keep = H.HEADS_FUSED
    for fused in (True, False):
        H.HEADS_FUSED = fused
        try:
            m = copy.deepcopy(clf).to(DEV)
            xg = x.to(DEV).requires_grad_(True)
            logits, offs = H.fused_head_logits(m, xg, list(tasks))
            (logits * gout.to(DEV)).sum().backward()
            got[fused] = (logits.detach(), xg.grad, {n: p.grad for n, p in m.named_parameters()})
        finally:
            H.HEADS_FUSED = keep
        assert_close(logits, want, 1e-4, f"logits (fused={fused})")
        assert_close(xg.grad, xr.grad, 1e-4, f"dx (fused={fused})")
        for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
            assert_close(p.grad, q.grad, 1e-4, f"d{n} (fused={fused})")
    assert_close(got[True][0], got[False][0], 2e-5, "logits: one launch vs three")
    assert_close(got[True][1], got[False][1], 2e-5, "dx: one launch vs three")
    for n in got[True][2]:
        assert_close(got[True][2][n], got[False][2][n], 2e-5, f"d{n}: one launch vs three")


def test_head_block_kernel_argument_checks():
    from analysisgnn_amd import _lib
    lib = _lib.load()
    import ctypes as C
    offs = (C.c_int32 * 3)(0, 4, 4)                      # a task without classes
    x = torch.zeros(8, 128, device=DEV)
    rc = lib.agnn_heads_fwd_f32(x.data_ptr(), 128, 8, 128, 64, 2, x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), 1e-5,
                                x.data_ptr(), None, x.data_ptr(), offs, x.data_ptr(), x.data_ptr(), 128, x.data_ptr(), x.data_ptr(),
                                x.data_ptr(), 4, None)
    assert rc != 0 and b"offs must increase" in lib.agnn_last_error()
    rc = lib.agnn_heads_fwd_f32(x.data_ptr(), 128, 8, 96, 64, 2, x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), 1e-5,
                                x.data_ptr(), None, x.data_ptr(), offs, x.data_ptr(), x.data_ptr(), 128, x.data_ptr(), x.data_ptr(),
                                x.data_ptr(), 4, None)
    assert rc != 0 and b"built for" in lib.agnn_last_error()
