"""The captured training step (what bench.py times) computes what the eager step computes: same loss, bit-identical
gradients, and a replay is reproducible.  C2-shaped model on 8 x 500-note subgraphs, dropout 0 (the dropout masks are
a function of the device-side step counter, which eager steps and replays advance differently)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _same_to_rounding(a, b, what=""):
    """Deferred weight gradients are issued TOGETHER at the flush points (linear.weight_grad_batch): the same products, but the
    number of row slices a product is cut into — the summation order — depends on what else is pending, so schedules that group
    them differently agree to fp32 rounding, not bit for bit.  2e-6 of the largest magnitude."""
    a, b = a.detach().double(), b.detach().double()
    err, scale = float((a - b).abs().max()), float(b.abs().max())
    assert err <= 2e-6 * scale + 1e-12, f"{what}: {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("defer", [False, True])
def test_graph_replay_equals_eager_step_and_is_reproducible(defer):
    """`defer`: the projections' weight gradients leave their place in the backward pass and run at the flush points of the
    hybrid encoder (dp.defer_weight_grads) — the same kernels on the same operands, so the gradients must not change by a bit
    (checked against the undeferred eager step)."""
    from analysisgnn_amd import dp, graph
    from analysisgnn_amd.heads import multitask_cross_entropy
    from analysisgnn_amd.models import TorchAnalysisGNN
    from analysisgnn_amd.synth import make_batch, torch_inputs
    dev = torch.device("cuda", 0)
    tasks = {"cadence": 4, "localkey": 50, "romanNumeral": 185, "hrythm": 2, "pcset": 94}
    g = make_batch(8, 500)
    I = torch_inputs(g, 25, dev, seed=0)
    labels = torch.stack([torch.randint(0, c, (I["batch_size"],), generator=torch.Generator().manual_seed(i)).to(dev)
                          for i, c in enumerate(tasks.values())])
    torch.manual_seed(0)
    model = TorchAnalysisGNN(g.metadata(), 25, 256, 128, tasks, 3, dropout=0.0, use_jk=False, logit_fusion=False).to(dev).train()
    flat = dp.FlatGradBuffer(model.parameters(), views=False)
    was = graph.index_cache_enabled
    graph.index_cache_enabled = False
    dp.enable_wgrad_overlap(True, "sequence")
    try:
        def fwd_bwd():
            flat.zero()
            x = model.encode(I["pitch_spelling"], I["key_signature"], I["x_dict"], I["edge_index_dict"], I["batch_dict"],
                             I["batch_size"], None, None)
            logits, offs, _ = model.forward_clf_fused(x)
            loss = 0.1 * x.pow(2).mean() + multitask_cross_entropy(logits, offs, labels, 0.1, -1).sum()
            loss.backward()
            flat.pack()
            return loss

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                loss_e = fwd_bwd()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        g_eager, l_eager = flat.flat.clone(), float(loss_e)
        assert torch.isfinite(g_eager).all() and float(g_eager.abs().max()) > 0
        if defer:
            dp.defer_weight_grads(True)
            with torch.cuda.stream(side):
                loss_d = fwd_bwd()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize()
            assert float(loss_d) == l_eager
            _same_to_rounding(flat.flat, g_eager, "deferred vs plain schedule")
            g_eager = flat.flat.clone()          # what the captured (deferred) step must reproduce BIT FOR BIT
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            loss_g = fwd_bwd()
        flat.flat.zero_()
        cg.replay()
        torch.cuda.synchronize()
        g1, l1 = flat.flat.clone(), float(loss_g)
        flat.flat.fill_(7.0)                     # a replay must overwrite every gradient slot
        cg.replay()
        torch.cuda.synchronize()
        g2 = flat.flat.clone()
        assert l1 == l_eager
        assert torch.equal(g1, g_eager), float((g1 - g_eager).abs().max())
        assert torch.equal(g1, g2)
    finally:
        dp.enable_wgrad_overlap(False)
        dp.defer_weight_grads(False)
        graph.index_cache_enabled = was


@pytest.mark.parametrize("late", [True, False])
def test_deferred_weight_gradients_on_ragged_sampled_batch(late):
    """dp.defer_weight_grads with both backward schedules of the hybrid encoder (the sequence branch behind a late-created node
    / in autograd order) on a neighbour-sampled batch whose three subgraphs have DIFFERENT lengths (padded sequences, every
    layer trimmed, the last one without edges): the loss identical, every gradient equal to the plain step's to fp32 rounding (the
    deferred products are issued as one batch: another summation order), the deferred step itself reproducible bit for bit."""
    from analysisgnn_amd import dp, encoders, graph
    from analysisgnn_amd.heads import MultiTaskLoss, training_loss
    from analysisgnn_amd.models import TorchAnalysisGNN
    from analysisgnn_amd.synth import make_score_graph, merge_sampled, sample_hops, torch_inputs
    dev = torch.device("cuda", 0)
    tasks = {"cadence": 4, "localkey": 50, "hrythm": 2}
    g = merge_sampled([sample_hops(make_score_graph(seed=sd, n_notes=400), nt, (5, 5), seed=sd, first_target=20)
                       for sd, nt in ((1, 150), (2, 90), (3, 201))])
    I = torch_inputs(g, 25, dev, seed=0)
    labels = torch.stack([torch.randint(0, c, (I["batch_size"],), generator=torch.Generator().manual_seed(i)).to(dev)
                          for i, c in enumerate(tasks.values())])
    torch.manual_seed(0)
    model = TorchAnalysisGNN(g.metadata(), 25, 256, 128, tasks, 3, dropout=0.0, use_jk=False, logit_fusion=False).to(dev).train()
    clf = MultiTaskLoss(list(tasks)).to(dev)
    both = torch.nn.ModuleDict({"m": model, "c": clf})
    flat = dp.FlatGradBuffer(both.parameters(), views=False)
    was, was_late = graph.index_cache_enabled, encoders.LATE_SEQUENCE_BACKWARD
    graph.index_cache_enabled = False
    dp.enable_wgrad_overlap(True, "sequence")

    def fwd_bwd():
        flat.zero()
        x = model.encode(I["pitch_spelling"], I["key_signature"], I["x_dict"], I["edge_index_dict"], I["batch_dict"], I["batch_size"],
                         I["neighbor_mask_node"], I["neighbor_mask_edge"])
        logits, offs, _ = model.forward_clf_fused(x)
        loss, _ = training_loss(logits, offs, labels, x, 0.1, 0.1, -1, task_params=clf.weights())
        loss.backward()
        flat.pack()
        return float(loss), flat.flat.clone()
    try:
        l0, g0 = fwd_bwd()
        assert torch.isfinite(g0).all() and float(g0.abs().max()) > 0
        encoders.LATE_SEQUENCE_BACKWARD = late
        dp.defer_weight_grads(True)
        l1, g1 = fwd_bwd()
        assert l1 == l0
        _same_to_rounding(g1, g0, "deferred vs plain schedule")
        l2, g2 = fwd_bwd()
        assert l2 == l1 and torch.equal(g2, g1)               # the deferred schedule itself is reproducible bit for bit
    finally:
        dp.defer_weight_grads(False)
        dp.enable_wgrad_overlap(False)
        encoders.LATE_SEQUENCE_BACKWARD = was_late
        graph.index_cache_enabled = was


def test_adjacent_parameter_layout_gives_the_same_step():
    """dp.plan_parameters + FlatAdamW put the task-head layers and the GRU direction pairs back to back, so the fused
    schedule reads them through views (params.cat_rows / stack_rows) instead of cat / pack launches: same loss and the same
    gradient for every parameter as the model whose parameters live in separate allocations."""
    from analysisgnn_amd import dp, graph
    from analysisgnn_amd.heads import multitask_cross_entropy
    from analysisgnn_amd.models import TorchAnalysisGNN
    from analysisgnn_amd.params import adjacent
    from analysisgnn_amd.synth import make_batch, torch_inputs
    dev = torch.device("cuda", 0)
    tasks = {"cadence": 4, "localkey": 50, "romanNumeral": 185, "hrythm": 2, "pcset": 94}
    g = make_batch(4, 500)
    I = torch_inputs(g, 25, dev, seed=0)
    labels = torch.stack([torch.randint(0, c, (I["batch_size"],), generator=torch.Generator().manual_seed(i)).to(dev)
                          for i, c in enumerate(tasks.values())])

    def build():
        torch.manual_seed(0)
        return TorchAnalysisGNN(g.metadata(), 25, 256, 128, tasks, 3, dropout=0.0, use_jk=False, logit_fusion=False).to(dev).train()

    def step(model):
        x = model.encode(I["pitch_spelling"], I["key_signature"], I["x_dict"], I["edge_index_dict"], I["batch_dict"],
                         I["batch_size"], None, None)
        logits, offs, _ = model.forward_clf_fused(x)
        loss = 0.1 * x.pow(2).mean() + multitask_cross_entropy(logits, offs, labels, 0.1, -1).sum()
        loss.backward()
        torch.cuda.synchronize()
        return float(loss)

    was = graph.index_cache_enabled
    graph.index_cache_enabled = False
    try:
        a = build()
        la = step(a)
        b = build()
        params, tight = dp.plan_parameters(b)
        flat = dp.FlatGradBuffer(params, views=False, tight=tight)
        dp.FlatAdamW(params, flat, lr=1e-3)
        heads = list(b.clf_dict.values())
        assert adjacent([m[0].weight for m in heads]) and adjacent([m[3].bias for m in heads])
        assert adjacent([b.encoder.rnn.weight_hh_l1, b.encoder.rnn.weight_hh_l1_reverse])
        assert not adjacent([m[3].bias for m in a.clf_dict.values()])
        flat.zero()
        lb = step(b)
        assert la == lb
        ga = dict(a.named_parameters())
        for n, p in b.named_parameters():
            assert p.grad is not None and torch.equal(p.grad, ga[n].grad), n
    finally:
        graph.index_cache_enabled = was


def test_single_stream_captured_step_is_stable_over_many_replays():
    """The step captured on ONE stream (sequence branch and weight gradients on the main stream) takes ROCm's fast replay
    path, where a memset node once left stale counters to the CSR build (an out-of-range write after a few replays): 40
    replays must keep producing the eager gradients."""
    from analysisgnn_amd import dp, graph
    from analysisgnn_amd.encoders import _HybridMixin
    from analysisgnn_amd.heads import training_loss
    from analysisgnn_amd.models import TorchAnalysisGNN
    from analysisgnn_amd.synth import make_batch, torch_inputs
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
    was, was_overlap = graph.index_cache_enabled, _HybridMixin.overlap_sequence_branch
    graph.index_cache_enabled = False
    _HybridMixin.overlap_sequence_branch = False
    dp.enable_wgrad_overlap(False)
    try:
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
        for _ in range(40):
            cg.replay()
        torch.cuda.synchronize()
        assert torch.equal(flat.flat, g_eager), float((flat.flat - g_eager).abs().max())
        # no CSR build of any replay met a position outside its row (the device status word the builds share): a
        # recurrence of the stale-counter state raises here instead of passing on an index with missing edges
        from analysisgnn_amd import _lib
        _lib.check_device_status(dev)
    finally:
        graph.index_cache_enabled = was
        _HybridMixin.overlap_sequence_branch = was_overlap


@pytest.mark.parametrize("mode", ["accumulate", "flat_views"])
def test_existing_grads_with_deferred_and_overlapped_weight_grads(mode):
    """with dp.defer_weight_grads / enable_wgrad_overlap a gradient may only be produced late (or on the weight-
    gradient stream) while its parameter has no `.grad` — otherwise AccumulateGrad adds at once, on the main stream.  The
    guard looks through the cats / stacks / packs of parameters (linear.all_steal).  Two micro-batches accumulated into
    `.grad`, and FlatGradBuffer(views=True) (gradients pre-set as views): equal to the plain schedule to fp32 rounding."""
    from analysisgnn_amd import dp, graph
    from analysisgnn_amd.heads import MultiTaskLoss, training_loss
    from analysisgnn_amd.linear import join_wgrad
    from analysisgnn_amd.models import TorchAnalysisGNN
    from analysisgnn_amd.synth import make_batch, torch_inputs
    dev = torch.device("cuda", 0)
    tasks = {"cadence": 4, "localkey": 50, "hrythm": 2}
    batches = []
    for seed in (0, 1):
        g = make_batch(5, 500, first_seed=10 * seed)
        I = torch_inputs(g, 25, dev, seed=seed)
        labels = torch.stack([torch.randint(0, c, (I["batch_size"],), generator=torch.Generator().manual_seed(7 * seed + i)).to(dev)
                              for i, c in enumerate(tasks.values())])
        batches.append((I, labels))
    torch.manual_seed(0)
    model = TorchAnalysisGNN(g.metadata(), 25, 256, 128, tasks, 3, dropout=0.0, use_jk=False, logit_fusion=False).to(dev).train()
    clf = MultiTaskLoss(list(tasks)).to(dev)
    both = torch.nn.ModuleDict({"m": model, "c": clf})
    params = [p for p in both.parameters() if p.requires_grad]
    was = graph.index_cache_enabled
    graph.index_cache_enabled = False
    flat = dp.FlatGradBuffer(params, views=True) if mode == "flat_views" else None

    def run(off_chain: bool):
        dp.enable_wgrad_overlap(off_chain, "all")
        dp.defer_weight_grads(off_chain)
        if flat is not None:
            flat.zero()
        else:
            for p in params:
                p.grad = None
        for I, labels in batches:
            x = model.encode(I["pitch_spelling"], I["key_signature"], I["x_dict"], I["edge_index_dict"], I["batch_dict"], I["batch_size"],
                             None, None)
            logits, offs, _ = model.forward_clf_fused(x)
            loss, _ = training_loss(logits, offs, labels, x, 0.1, 0.1, -1, task_params=clf.weights())
            loss.backward()
        join_wgrad()
        torch.cuda.synchronize()
        return [p.grad.detach().clone() for p in params]
    try:
        g0 = run(False)
        g1 = run(True)
        assert all(torch.isfinite(t).all() for t in g0) and max(float(t.abs().max()) for t in g0) > 0
        for a, b, (name, _) in zip(g0, g1, both.named_parameters()):
            _same_to_rounding(b, a, name)
    finally:
        dp.defer_weight_grads(False)
        dp.enable_wgrad_overlap(False)
        graph.index_cache_enabled = was


def test_nested_wgrad_stream_inside_a_capture():
    """Round 2's capture crash (c6b44df): a `wgrad_stream` entered while ALREADY on the weight-gradient stream made that
    stream wait for itself, and hipStreamEndCapture crashed.  The guard (linear.wgrad_stream.__enter__: no wait when the current
    stream is the weight-gradient stream) is pinned here: nested entry inside a capture, replayed twice."""
    from analysisgnn_amd import dp, linear
    dev = torch.device("cuda", 0)
    x = torch.arange(1024, dtype=torch.float32, device=dev)
    out = torch.zeros_like(x)
    dp.enable_wgrad_overlap(True, "all")
    try:
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                       # warm-up outside the capture (allocations, stream creation)
            with linear.wgrad_stream(dev, x):
                with linear.wgrad_stream(dev, x):
                    out.copy_(x)
            linear.join_wgrad()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            with linear.wgrad_stream(dev, x) as outer:
                assert outer.on
                y = x * 2.0
                with linear.wgrad_stream(dev, y) as inner:  # already on the stream
                    assert inner.on
                    out.copy_(y + 1.0)
            linear.join_wgrad()
        for _ in range(2):
            out.zero_()
            cg.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, x * 2.0 + 1.0)
    finally:
        dp.enable_wgrad_overlap(False)


def test_split_backward_two_buckets_equal_the_unsplit_step():
    """N > 1 schedule of bench.py on one rank: the autograd graph cut behind the input layers (models.split_backward), the
    gradient buffer in two buckets (dp.plan_parameters(late=...), pack("early") / finish_backward / pack("late")).  Same kernels
    in the same order (the deferred weight-gradient products grouped differently: equal to fp32 rounding); the late bucket holds
    exactly the input layers."""
    from analysisgnn_amd import dp, graph
    from analysisgnn_amd.heads import MultiTaskLoss, training_loss
    from analysisgnn_amd.models import TorchAnalysisGNN
    from analysisgnn_amd.synth import make_sampled_batch, torch_inputs
    dev = torch.device("cuda", 0)
    tasks = {"cadence": 4, "localkey": 50, "hrythm": 2}
    g = make_sampled_batch(4, 500, (5, 5), first_seed=3)
    I = torch_inputs(g, 25, dev, seed=0)
    labels = torch.stack([torch.randint(0, c, (I["batch_size"],), generator=torch.Generator().manual_seed(i)).to(dev)
                          for i, c in enumerate(tasks.values())])
    torch.manual_seed(0)
    model = TorchAnalysisGNN(g.metadata(), 25, 256, 128, tasks, 3, dropout=0.0, use_jk=False, logit_fusion=False).to(dev).train()
    clf = MultiTaskLoss(list(tasks)).to(dev)
    both = torch.nn.ModuleDict({"m": model, "c": clf})
    was = graph.index_cache_enabled
    graph.index_cache_enabled = False
    dp.enable_wgrad_overlap(True, "sequence")
    dp.defer_weight_grads(True)

    def run(split):
        model.split_backward = split
        late = model.late_parameters() if split else []
        params, tight = dp.plan_parameters(both, late=late)
        flat = dp.FlatGradBuffer(params, views=False, tight=tight, late=late)
        flat.zero()
        x = model.encode(I["pitch_spelling"], I["key_signature"], I["x_dict"], I["edge_index_dict"], I["batch_dict"], I["batch_size"],
                         I["neighbor_mask_node"], I["neighbor_mask_edge"])
        logits, offs, _ = model.forward_clf_fused(x)
        loss, _ = training_loss(logits, offs, labels, x, 0.1, 0.1, -1, task_params=clf.weights())
        loss.backward()
        if split:
            assert all(p.grad is None for p in late) and sum(p.grad is not None for p in params) > 50
            work = flat.all_reduce_early_async()
            assert work is None                                   # one rank: nothing to ship
            model.finish_backward()
            flat.all_reduce_late_and_finish(work)
            assert flat.cut == flat.flat.numel() - sum((p.numel() + 3) // 4 * 4 for p in late)
        else:
            flat.all_reduce_mean()
        torch.cuda.synchronize()
        return {n: p.grad.detach().clone() for n, p in both.named_parameters()}, float(loss)
    try:
        g0, l0 = run(False)
        g1, l1 = run(True)
        assert l0 == l1
        for n in g0:
            _same_to_rounding(g1[n], g0[n], n)
    finally:
        model.split_backward = False
        dp.defer_weight_grads(False)
        dp.enable_wgrad_overlap(False)
        graph.index_cache_enabled = was


@pytest.mark.parametrize("enc", ["hybridgnn", "hgt"])
def test_every_backward_schedule_switch_gives_the_same_gradients(enc):
    """bench.py captures the step under up to 12 schedule variants and keeps the fastest replay: {sequence branch behind a late
    node / in autograd order} x {its deferred products in the main flush / in its own} x {all / 3/4 / half of the main flush's
    products at the flush point, the rest on the sequence branch's stream}; the graph stack's second layer may wait for the GRU's
    inner input projection.  Every combination must produce the plain (nothing deferred) step's loss exactly and its gradients
    to fp32 rounding (batched products sum in another order), and must itself be reproducible bit for bit.  HGT: the relation
    weights' gradients are among the deferred products."""
    from analysisgnn_amd import dp, encoders, graph, gru, linear
    from analysisgnn_amd.heads import MultiTaskLoss, training_loss
    from analysisgnn_amd.models import TorchAnalysisGNN
    from analysisgnn_amd.synth import make_batch, make_score_graph, merge_sampled, sample_hops, torch_inputs
    dev = torch.device("cuda", 0)
    tasks = {"cadence": 4, "localkey": 50, "hrythm": 2}
    if enc == "hgt":
        g = make_batch(3, 150, first_seed=11, add_beats=True, add_measures=True, reverse_metrical_edges=True)
    else:
        g = merge_sampled([sample_hops(make_score_graph(seed=sd, n_notes=400), 150, (5, 5), seed=sd, first_target=20) for sd in (1, 2, 3)])
    I = torch_inputs(g, 25, dev, seed=0)
    labels = torch.stack([torch.randint(0, c, (I["batch_size"],), generator=torch.Generator().manual_seed(i)).to(dev)
                          for i, c in enumerate(tasks.values())])
    torch.manual_seed(0)
    model = TorchAnalysisGNN(g.metadata(), 25, 256, 128, tasks, 3, dropout=0.0, use_jk=False, logit_fusion=False, encoder_type=enc).to(dev).train()
    clf = MultiTaskLoss(list(tasks)).to(dev)
    flat = dp.FlatGradBuffer(torch.nn.ModuleDict({"m": model, "c": clf}).parameters(), views=False)
    saved = (graph.index_cache_enabled, encoders.LATE_SEQUENCE_BACKWARD, linear.ITEMS_HOME, linear.FLUSH_KEEP, gru.YIELD_TO_PROJECTIONS)
    graph.index_cache_enabled = False

    def fwd_bwd():
        flat.zero()
        x = model.encode(I["pitch_spelling"], I["key_signature"], I["x_dict"], I["edge_index_dict"], I["batch_dict"], I["batch_size"],
                         I["neighbor_mask_node"], I["neighbor_mask_edge"])
        logits, offs, _ = model.forward_clf_fused(x)
        loss, _ = training_loss(logits, offs, labels, x, 0.1, 0.1, -1, task_params=clf.weights())
        loss.backward()
        flat.pack()
        torch.cuda.synchronize()
        return float(loss), flat.flat.clone()
    try:
        gru.YIELD_TO_PROJECTIONS = False
        l0, g0 = fwd_bwd()
        assert torch.isfinite(g0).all() and float(g0.abs().max()) > 0
        dp.defer_weight_grads(True)
        for late in (True, False):
            for home in (True, False):
                for keep in (1.0, 0.75, 0.5, 0.0):
                    for yld in ((True, False) if (late, home, keep) == (True, True, 0.75) else (True,)):
                        encoders.LATE_SEQUENCE_BACKWARD, linear.ITEMS_HOME, linear.FLUSH_KEEP, gru.YIELD_TO_PROJECTIONS = late, home, keep, yld
                        what = f"late={late} home={home} keep={keep} yield={yld}"
                        l1, g1 = fwd_bwd()
                        assert l1 == l0, what
                        _same_to_rounding(g1, g0, what)
                        l2, g2 = fwd_bwd()
                        assert l2 == l1 and torch.equal(g2, g1), what
    finally:
        dp.defer_weight_grads(False)
        graph.index_cache_enabled, encoders.LATE_SEQUENCE_BACKWARD, linear.ITEMS_HOME, linear.FLUSH_KEEP, gru.YIELD_TO_PROJECTIONS = saved


@pytest.mark.parametrize("enc", ["hybridgnn", "hgt", "metricalgnn"])
@pytest.mark.parametrize("defer", [False, True])
def test_step_never_reads_memory_it_has_not_written(enc, defer):
    """Every `torch.empty` of the step NaN-filled (torch.utils.deterministic.fill_uninitialized_memory): loss and gradients
    must come out finite and bit for bit what they are on ordinary recycled memory.  A kernel that leaves part of its output
    unwritten, or neutralises a stand-in operand by `0 * x`, passes every parity test until the allocator hands it a block
    that holds a NaN (round 3: the recurrence kernels' dropout-scale stand-in)."""
    from analysisgnn_amd import dp, graph
    from analysisgnn_amd.heads import MultiTaskLoss, training_loss, unit_gradient
    from analysisgnn_amd.models import TorchAnalysisGNN
    from analysisgnn_amd.synth import make_batch, torch_inputs
    dev = torch.device("cuda", 0)
    tasks = {"cadence": 4, "localkey": 50, "hrythm": 2}
    hetero = enc != "hybridgnn"
    g = make_batch(6, 400, first_seed=3, add_beats=hetero, add_measures=hetero)
    I = torch_inputs(g, 25, dev, seed=0)
    labels = torch.stack([torch.randint(0, c, (I["batch_size"],), generator=torch.Generator().manual_seed(i)).to(dev)
                          for i, c in enumerate(tasks.values())])
    torch.manual_seed(0)
    model = TorchAnalysisGNN(g.metadata(), 25, 256, 128, tasks, 3, dropout=0.0, use_jk=False, logit_fusion=False, encoder_type=enc).to(dev).train()
    clf = MultiTaskLoss(list(tasks)).to(dev)
    mods = torch.nn.ModuleDict({"m": model, "c": clf})
    flat = dp.FlatGradBuffer(mods.parameters(), views=False)
    saved = (graph.index_cache_enabled, torch.are_deterministic_algorithms_enabled(), torch.is_deterministic_algorithms_warn_only_enabled(),
             torch.utils.deterministic.fill_uninitialized_memory)
    graph.index_cache_enabled = False

    def fwd_bwd():
        flat.zero()
        x = model.encode(I["pitch_spelling"], I["key_signature"], I["x_dict"], I["edge_index_dict"], I["batch_dict"], I["batch_size"],
                         I["neighbor_mask_node"], I["neighbor_mask_edge"])
        logits, offs, _ = model.forward_clf_fused(x)
        loss, _ = training_loss(logits, offs, labels, x, 0.1, 0.1, -1, task_params=clf.weights())
        loss.backward(gradient=unit_gradient(dev))
        flat.pack()
        torch.cuda.synchronize()
        return loss.detach().clone(), flat.flat.clone()
    try:
        dp.defer_weight_grads(defer)
        l0, g0 = fwd_bwd()
        torch.use_deterministic_algorithms(True, warn_only=True)
        torch.utils.deterministic.fill_uninitialized_memory = True
        l1, g1 = fwd_bwd()
        assert torch.isfinite(l1) and torch.isfinite(g1).all()
        assert torch.equal(l0, l1) and torch.equal(g0, g1)
    finally:
        dp.defer_weight_grads(False)
        torch.use_deterministic_algorithms(saved[1], warn_only=saved[2])
        graph.index_cache_enabled, torch.utils.deterministic.fill_uninitialized_memory = saved[0], saved[3]
