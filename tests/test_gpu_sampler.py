"""Device-side batch assembly (csrc/sampler.hip, analysisgnn_amd/batching.py) against oracle/sampler_ref.py: node ids and
edge slots bit for bit; the padded static-shape batch gives the model the same target rows as the compact batch; the
whole sampled step replays from ONE captured hipGraph with a different batch per replay."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import assert_close  # noqa: E402

DEV = torch.device("cuda:0")


def _store(n_scores=4, n_notes=800, tasks=None):
    from analysisgnn_amd.batching import ScoreStore
    from analysisgnn_amd.synth import make_score_graph
    graphs = [make_score_graph(seed=60 + i, n_notes=n_notes) for i in range(n_scores)]
    return ScoreStore(graphs, 25, DEV, tasks=tasks or {"cadence": 4, "localkey": 50}, seed=3), graphs


@pytest.mark.parametrize("fan,cap,T", [((5, 5), (64, 64), 500), ((1, 2), (128, 96), 200), ((3,), (32,), 64)])
def test_sampler_matches_oracle_bit_for_bit(fan, cap, T):
    from analysisgnn_amd import _lib
    from analysisgnn_amd.batching import DeviceSampler
    from oracle import sampler_ref as S
    store, _ = _store()
    B = 6
    smp = DeviceSampler(store, B, T, fan, cap, seed=11)
    rng = np.random.default_rng(1)
    rowptr = [c.rowptr.cpu().numpy() for c in store.csr]
    base = rowptr[0][0]
    col = [c.col.cpu().numpy() for c in store.csr]
    for step in (1, 2):
        wins = store.random_windows(B, T, rng)
        smp.set_windows(wins)
        batch = smp.sample()
        torch.cuda.synchronize()
        # the store's CSR segments index one shared col array: rebase rowptr for the oracle
        gid, edges, dropped = S.sample_hops([rp - 0 for rp in rowptr], col, wins, T, fan, cap, seed=11, step=step)
        assert np.array_equal(smp.node_gid.cpu().numpy(), gid)
        for r, et in enumerate(store.edge_types):
            assert np.array_equal(batch["edge_index_dict"][et].cpu().numpy(), edges[r]), et
        # gathered attributes: features, spelling / key, labels of the targets
        g = torch.from_numpy(gid.astype(np.int64)).to(DEV)
        real = g >= 0
        assert torch.equal(smp.x[real], store.x[g[real]]) and float(smp.x[~real].abs().sum()) == 0.0
        assert torch.equal(batch["pitch_spelling"][real], store.attrs[0][g[real]])
        assert torch.equal(batch["label_matrix"], store.attrs[2:, g[:B * T]])
    _lib.check_device_status(DEV)                                   # no capacity overflow at these capacities


def test_capacity_cut_is_a_statistic_not_an_error():
    """Sources cut by cap[h] are counted in the sampler's own `drops` word; the device status word (index corruption,
    agnn_check_status) stays clean, so one crowded batch does not fail every later check of the process."""
    from analysisgnn_amd import _lib
    from analysisgnn_amd.batching import DeviceSampler
    store, _ = _store()
    word = _lib.status_word(DEV)
    before = int(word.item())
    smp = DeviceSampler(store, 2, 100, (5, 5), (4, 4), seed=1)
    smp.set_windows(np.array([300, 1200], dtype=np.int32))
    smp.sample()
    torch.cuda.synchronize()
    assert smp.dropped() > 0
    assert int(word.item()) == before
    _lib.check_device_status(DEV)


def _crowded_store(n_targets, n_rel, fan):
    """One score whose first `n_targets` notes each have `fan` in-neighbours per relation, ALL distinct and all outside
    the window [0, n_targets): n_targets * n_rel * fan distinct candidates in a single subgraph."""
    from analysisgnn_amd.batching import ScoreStore
    from analysisgnn_amd.synth import ScoreGraph
    n = n_targets * (1 + n_rel * fan) + 8
    ei = {}
    for r in range(n_rel):
        dst = np.repeat(np.arange(n_targets), fan)
        src = n_targets + (dst * n_rel + r) * fan + np.tile(np.arange(fan), n_targets)
        ei[("note", f"rel{r}", "note")] = np.stack([src, dst]).astype(np.int64)
    g = ScoreGraph(num_nodes={"note": n}, edge_index=ei, batch={"note": np.zeros(n, dtype=np.int64)},
                   onset_div=np.arange(n, dtype=np.int64), duration_div=np.ones(n, dtype=np.int64))
    return ScoreStore([g], 25, DEV, tasks={"a": 3}, seed=0)


def test_crowded_subgraph_is_deterministic_and_terminates():
    """1 500 distinct out-of-window candidates in one subgraph (more than the kernel's candidate list, fewer than
    its table): the cap smallest ids are kept whatever the order of the atomics — bit-identical to the oracle."""
    from analysisgnn_amd import _lib
    from analysisgnn_amd.batching import DeviceSampler
    from oracle import sampler_ref as S
    T, R, fan, cap = 300, 1, (5, 2), (64, 32)
    store = _crowded_store(T, R, fan[0])
    smp = DeviceSampler(store, 1, T, fan, cap, seed=5)
    wins = np.array([0], dtype=np.int32)
    smp.set_windows(wins)
    batch = smp.sample()
    torch.cuda.synchronize()
    rowptr = [c.rowptr.cpu().numpy() for c in store.csr]
    col = [c.col.cpu().numpy() for c in store.csr]
    gid, edges, dropped = S.sample_hops(rowptr, col, wins, T, fan, cap, seed=5, step=1)
    assert dropped == T * R * fan[0] - cap[0]
    assert np.array_equal(smp.node_gid.cpu().numpy(), gid)
    for r, et in enumerate(store.edge_types):
        assert np.array_equal(batch["edge_index_dict"][et].cpu().numpy(), edges[r]), et
    assert smp.dropped() == dropped
    _lib.check_device_status(DEV)


def test_more_candidates_than_the_table_holds_is_an_error_not_a_hang():
    """5 000 distinct candidates > the 4 096-entry LDS table: the launch terminates (every probe loop is bounded), the
    sources that found no place are left out, and the device status word reports it."""
    from analysisgnn_amd import _lib
    from analysisgnn_amd.batching import DeviceSampler
    T, R, fan, cap = 500, 2, (5,), (64,)
    store = _crowded_store(T, R, fan[0])
    word = _lib.status_word(DEV)
    before = int(word.item())
    smp = DeviceSampler(store, 1, T, fan, cap, seed=5)
    smp.set_windows(np.array([0], dtype=np.int32))
    batch = smp.sample()
    torch.cuda.synchronize()
    assert int(word.item()) > before
    with pytest.raises(_lib.AgnnError):
        _lib.check_device_status(DEV)
    word.zero_()                                                    # leave the shared word clean for the other tests
    gid = smp.node_gid.cpu().numpy()
    assert (gid[T:] >= T).all()                                     # the hop block is full of real out-of-window notes
    for et in store.edge_types:
        e = batch["edge_index_dict"][et].cpu().numpy()
        live = e[0] >= 0
        assert (gid[e[0][live]] >= 0).all() and (e[1][live] < T).all()


def test_padded_batch_equals_compact_batch_on_the_model():
    """Same sampled subgraphs, once as the padded static-shape batch the sampler writes and once compacted the way a
    NeighborLoader would hand them over (padding slots and (-1, -1) edges removed, per-hop counts = the real counts):
    the target rows of the encoder output agree to rounding.  Checked against the CPU restatement too."""
    from analysisgnn_amd.batching import DeviceSampler
    from analysisgnn_amd.encoders import HybridGNN
    from oracle import encoders_ref as E
    store, _ = _store(tasks={"a": 3})
    B, T, H = 3, 200, 32
    smp = DeviceSampler(store, B, T, (5, 5), (48, 48), seed=2)
    smp.set_windows(store.random_windows(B, T, np.random.default_rng(4)))
    batch = smp.sample()
    torch.cuda.synchronize()
    md = (["note"], store.edge_types)
    torch.manual_seed(0)
    m = HybridGNN(metadata=md, input_channels=H, hidden_channels=H, num_layers=3, dropout=0.0).eval()
    P = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(DEV)
    gen = torch.Generator().manual_seed(5)
    feat = torch.randn(store.num_notes, H, generator=gen)
    gid = smp.node_gid.cpu().long()
    x = torch.where((gid >= 0).unsqueeze(1), feat[gid.clamp(min=0)], torch.zeros(1, H))
    with torch.no_grad():
        out = m(x_dict={"note": x.to(DEV)}, edge_index_dict=batch["edge_index_dict"], batch_dict=batch["batch_dict"],
                batch_size=smp.batch_size, neighbor_mask_node=batch["neighbor_mask_node"], neighbor_mask_edge=batch["neighbor_mask_edge"])
    # compact form for the CPU path
    keep = gid >= 0
    new_id = torch.cumsum(keep.long(), 0) - 1
    blocks = smp.num_sampled_nodes["note"]
    nb = np.concatenate([[0], np.cumsum(blocks)])
    nodes_per_hop = [int(keep[nb[h]:nb[h + 1]].sum()) for h in range(len(blocks))]
    ei_c, edges_per_hop = {}, {}
    for et, e in batch["edge_index_dict"].items():
        e = e.cpu()
        counts, parts, lo = [], [], 0
        for eh in smp.num_sampled_edges[et]:
            seg = e[:, lo:lo + eh]
            seg = seg[:, seg[0] >= 0]
            parts.append(new_id[seg])
            counts.append(int(seg.shape[1]))
            lo += eh
        ei_c[et], edges_per_hop[et] = torch.cat(parts, dim=1), counts
    xc = x[keep]
    bc = batch["batch_dict"]["note"].cpu()[keep]
    with torch.no_grad():
        ref = E.hybrid_gnn(P, "", md, 3, {"note": xc}, ei_c, {"note": bc}, smp.batch_size, {"note": nodes_per_hop}, edges_per_hop)
    assert_close(out, ref, 1e-4, "padded batch vs compact batch")


def test_sampled_step_replays_from_one_graph_with_a_fresh_batch_each_time():
    """sample -> gather -> CSR build -> forward -> objective -> backward captured ONCE; every replay runs on other windows and
    another random step and must give the loss / gradients of the same batch run eagerly."""
    from analysisgnn_amd import dp, graph
    from analysisgnn_amd.batching import DeviceSampler
    from analysisgnn_amd.heads import training_loss
    from analysisgnn_amd.models import TorchAnalysisGNN
    tasks = {"cadence": 4, "localkey": 50, "romanNumeral": 185}
    store, _ = _store(n_scores=5, n_notes=900, tasks=tasks)
    B, T = 4, 500
    smp = DeviceSampler(store, B, T, (5, 5), (64, 64), seed=9)
    md = (["note"], store.edge_types)
    torch.manual_seed(0)
    model = TorchAnalysisGNN(md, 25, 256, 128, tasks, 3, dropout=0.0, use_jk=False, logit_fusion=False).to(DEV).train()
    params, tight = dp.plan_parameters(model)
    flat = dp.FlatGradBuffer(params, views=False, tight=tight)
    was = graph.index_cache_enabled
    graph.index_cache_enabled = False
    dp.enable_wgrad_overlap(False)
    rng = np.random.default_rng(8)
    try:
        def step(resample=True):
            flat.zero()
            b = smp.sample() if resample else smp.batch
            x = model.encode(b["pitch_spelling"], b["key_signature"], b["x_dict"], b["edge_index_dict"], b["batch_dict"], b["batch_size"],
                             b["neighbor_mask_node"], b["neighbor_mask_edge"])
            logits, offs, _ = model.forward_clf_fused(x)
            loss, _ = training_loss(logits, offs, b["label_matrix"], x, 0.1, 0.1, -1)
            loss.backward()
            flat.pack()
            return loss

        smp.set_windows(store.random_windows(B, T, rng))
        side = torch.cuda.Stream(device=DEV)
        side.wait_stream(torch.cuda.current_stream(DEV))
        with torch.cuda.stream(side):
            for _ in range(2):
                step()
        torch.cuda.current_stream(DEV).wait_stream(side)
        torch.cuda.synchronize()
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            loss_g = step()
        losses = []
        for it in range(3):
            smp.set_windows(store.random_windows(B, T, rng))
            cg.replay()
            torch.cuda.synchronize()
            g_graph, l_graph = flat.flat.clone(), float(loss_g)
            gid_graph = smp.node_gid.clone()
            l_eager = float(step(resample=False))                   # the same batch (buffers as the replay left them), eagerly
            torch.cuda.synchronize()
            assert torch.equal(smp.node_gid, gid_graph)
            assert l_graph == l_eager
            assert torch.equal(flat.flat, g_graph)
            losses.append(l_graph)
        assert len(set(losses)) == 3                                # three different batches
    finally:
        graph.index_cache_enabled = was


def _hetero_store(n_scores=4, n_notes=800, tasks=None):
    from analysisgnn_amd.batching import ScoreStore
    from analysisgnn_amd.synth import make_score_graph
    graphs = [make_score_graph(seed=80 + i, n_notes=n_notes, add_beats=True, add_measures=True) for i in range(n_scores)]
    for g in graphs:                                   # the C3 relation set: four note-note types + note->beat + note->measure
        g.edge_index = {et: e for et, e in g.edge_index.items() if et[0] == "note"}
    return ScoreStore(graphs, 25, DEV, tasks=tasks or {"cadence": 4, "localkey": 50}, seed=3), graphs


@pytest.mark.parametrize("fan,cap,T,gcap", [((5, 5), (64, 64), 500, {"beat": 160, "measure": 48}), ((3,), (32,), 64, {"beat": 24, "measure": 4})])
def test_metrical_members_match_oracle_bit_for_bit(fan, cap, T, gcap):
    """Beats / measures of a sampled batch (agnn_sample_members) against oracle/sampler_ref.sample_members: group slots, membership
    edge slots, gathered features — integers compared exactly; the second case is tight enough to cut groups (counted in
    `drops`, not an error)."""
    from analysisgnn_amd import _lib
    from analysisgnn_amd.batching import DeviceSampler
    from oracle import sampler_ref as S
    store, graphs = _hetero_store()
    assert store.group_types == ["beat", "measure"]
    B = 5
    smp = DeviceSampler(store, B, T, fan, cap, seed=11, group_capacity=gcap)
    assert smp.metadata()[1][-2:] == [("note", "connects", "beat"), ("note", "connects", "measure")]
    rng = np.random.default_rng(2)
    wins = store.random_windows(B, T, rng)
    smp.set_windows(wins)
    batch = smp.sample()
    torch.cuda.synchronize()
    gid = smp.node_gid.cpu().numpy()
    dropped = 0
    for t in store.group_types:
        gof = store.group_of[t].cpu().numpy()
        ggid, edges, dr = S.sample_members(gid, gof, wins, T, cap, gcap[t])
        dropped += dr
        assert np.array_equal(smp.group_gid[t].cpu().numpy(), ggid), t
        e = batch["edge_index_dict"][("note", "connects", t)].cpu().numpy()
        assert np.array_equal(e, edges), t
        live = e[0] >= 0
        assert live.sum() >= B * T * 0.9 if not dr else live.sum() > 0          # (nearly) every target note has its beat / measure
        # an edge joins a note and a group of the SAME subgraph, and the group is the note's own
        sub_note = batch["batch_dict"]["note"].cpu().numpy()[e[0][live]]
        sub_grp = batch["batch_dict"][t].cpu().numpy()[e[1][live]]
        assert np.array_equal(sub_note, sub_grp)
        assert np.array_equal(gof[gid[e[0][live]]], ggid[e[1][live]])
        gg = torch.from_numpy(ggid.astype(np.int64)).to(DEV)
        real = gg >= 0
        assert torch.equal(smp.group_x[t][real], store.group_x[t][gg[real]]) and float(smp.group_x[t][~real].abs().sum()) == 0.0
        assert batch["neighbor_mask_node"][t] == [B * gcap[t]] + [0] * len(cap)
        assert sum(batch["neighbor_mask_edge"][("note", "connects", t)]) == smp.num_nodes
    assert (dropped > 0) == (T == 64)
    assert smp.dropped() >= dropped
    _lib.check_device_status(DEV)


def test_hgt_on_the_padded_hetero_batch_equals_the_compact_batch():
    """C3's encoder (HybridHGT, note + beat + measure, six relation types) on the static-shape batch the device sampler writes
    (padding notes / groups / (-1, -1) edge slots, per-hop CAPACITIES as trim counts) against the CPU restatement on the same
    subgraphs compacted the way a loader would hand them over (real per-hop counts): target rows agree to 1e-4."""
    from analysisgnn_amd.batching import DeviceSampler
    from analysisgnn_amd.hgt import HybridHGT
    from oracle import encoders_ref as E
    store, _ = _hetero_store(tasks={"a": 3})
    B, T, H = 3, 200, 32
    smp = DeviceSampler(store, B, T, (5, 5), (48, 48), seed=2, group_capacity={"beat": 72, "measure": 24})
    smp.set_windows(store.random_windows(B, T, np.random.default_rng(4)))
    batch = smp.sample()
    torch.cuda.synchronize()
    assert smp.dropped() == 0
    md = smp.metadata()
    torch.manual_seed(1)
    m = HybridHGT(metadata=md, input_channels=H, hidden_channels=H, num_layers=3, heads=4, dropout=0.0).eval()
    P = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(DEV)
    gen = torch.Generator().manual_seed(5)
    ids = {"note": smp.node_gid.cpu().long(), **{t: smp.group_gid[t].cpu().long() for t in store.group_types}}
    total = {"note": store.num_notes, **{t: int(store.group_x[t].shape[0]) for t in store.group_types}}
    x = {}
    for t in md[0]:
        feat = torch.randn(total[t], H, generator=gen)
        x[t] = torch.where((ids[t] >= 0).unsqueeze(1), feat[ids[t].clamp(min=0)], torch.zeros(1, H))
    with torch.no_grad():
        out = m(x_dict={t: v.to(DEV) for t, v in x.items()}, edge_index_dict=batch["edge_index_dict"], batch_dict=batch["batch_dict"],
                batch_size=smp.batch_size, neighbor_mask_node=batch["neighbor_mask_node"], neighbor_mask_edge=batch["neighbor_mask_edge"])
    keep = {t: ids[t] >= 0 for t in md[0]}
    new_id = {t: torch.cumsum(keep[t].long(), 0) - 1 for t in md[0]}
    nodes_per_hop = {}
    for t in md[0]:
        nb = np.concatenate([[0], np.cumsum(smp.num_sampled_nodes[t])])
        nodes_per_hop[t] = [int(keep[t][nb[h]:nb[h + 1]].sum()) for h in range(len(nb) - 1)]
    ei_c, edges_per_hop = {}, {}
    for et, e in batch["edge_index_dict"].items():
        e = e.cpu()
        counts, parts, lo = [], [], 0
        for eh in smp.num_sampled_edges[et]:
            seg = e[:, lo:lo + eh]
            seg = seg[:, seg[0] >= 0]
            parts.append(torch.stack([new_id[et[0]][seg[0]], new_id[et[2]][seg[1]]]))
            counts.append(int(seg.shape[1]))
            lo += eh
        ei_c[et], edges_per_hop[et] = torch.cat(parts, dim=1), counts
    xc = {t: x[t][keep[t]] for t in md[0]}
    bc = {t: batch["batch_dict"][t].cpu()[keep[t]] for t in md[0]}
    with torch.no_grad():
        ref = E.hybrid_hgt(P, "", md, 3, 4, xc, ei_c, bc, smp.batch_size, nodes_per_hop, edges_per_hop)
    assert_close(out, ref, 1e-4, "padded hetero batch vs compact batch")


@pytest.mark.parametrize("hetero", [False, True])
def test_pooled_batch_matches_oracle_and_the_padded_batch(hetero):
    """agnn_sample_compact: the padded hop blocks squeezed into batch-wide pools — node ids, subgraph ids and remapped edge slots
    bit for bit against oracle/sampler_ref.compact applied to the padded batch of the same windows / seed / step; the second
    hop's pool is set too small on purpose (nodes past its end are dropped with their edges and counted).  With roomy pools the
    encoder's target rows on the pooled batch equal those on the padded batch."""
    from analysisgnn_amd import _lib
    from analysisgnn_amd.batching import DeviceSampler
    from analysisgnn_amd.encoders import HybridGNN
    from oracle import sampler_ref as S
    store, _ = _hetero_store() if hetero else _store()
    B, T, fan, cap = 6, 200, (5, 5), (48, 48)
    wins = store.random_windows(B, T, np.random.default_rng(6))
    pad = DeviceSampler(store, B, T, fan, cap, seed=4)
    pad.set_windows(wins)
    bp = pad.sample()
    torch.cuda.synchronize()
    gid_p = pad.node_gid.cpu().numpy()
    real = [int((gid_p[B * T + h * B * 48:B * T + (h + 1) * B * 48] >= 0).sum()) for h in range(2)]
    for pool in ((real[0] + 7, max(real[1] - 5, 1)), (real[0] + 16, real[1] + 16)):
        smp = DeviceSampler(store, B, T, fan, cap, seed=4, pool=pool)
        smp.set_windows(wins)
        b = smp.sample()
        torch.cuda.synchronize()
        n_rel = len(store.edge_types)
        e_pad = [bp["edge_index_dict"][et].cpu().numpy() for et in store.edge_types]
        gid, edges, batch, dropped = S.compact(gid_p, e_pad, B, T, cap, pool)
        assert np.array_equal(smp.node_gid.cpu().numpy(), gid)
        assert np.array_equal(b["batch_dict"]["note"].cpu().numpy(), batch)
        for r, et in enumerate(store.edge_types):
            assert np.array_equal(b["edge_index_dict"][et].cpu().numpy(), edges[r]), et
        assert (dropped > 0) == (pool[1] < real[1]) and smp.dropped() >= dropped
        assert b["neighbor_mask_node"]["note"] == [B * T, pool[0], pool[1]] and smp.num_nodes == B * T + sum(pool)
        for t in store.group_types:                                   # membership edges on the pool layout
            e = b["edge_index_dict"][("note", "connects", t)].cpu().numpy()
            live = e[0] >= 0
            assert np.array_equal(e[0][live], np.nonzero(live)[0]) and e.shape[1] == smp.num_nodes
            assert np.array_equal(batch[e[0][live]], b["batch_dict"][t].cpu().numpy()[e[1][live]])
            gof = store.group_of[t].cpu().numpy()
            assert np.array_equal(gof[gid[e[0][live]]], smp.group_gid[t].cpu().numpy()[e[1][live]])
            assert b["neighbor_mask_edge"][("note", "connects", t)] == [B * T + pool[0], pool[1]]
    _lib.check_device_status(DEV)
    if hetero:
        return
    # same subgraphs, two layouts: the encoder's target rows agree
    H = 32
    md = (["note"], store.edge_types)
    torch.manual_seed(3)
    m = HybridGNN(metadata=md, input_channels=H, hidden_channels=H, num_layers=3, dropout=0.0).to(DEV).eval()
    feat = torch.randn(store.num_notes, H, generator=torch.Generator().manual_seed(5)).to(DEV)
    outs = []
    for s_, bb in ((pad, bp), (smp, b)):
        g = s_.node_gid.long()
        x = torch.where((g >= 0).unsqueeze(1), feat[g.clamp(min=0)], torch.zeros(1, H, device=DEV))
        with torch.no_grad():
            outs.append(m(x_dict={"note": x}, edge_index_dict={et: bb["edge_index_dict"][et] for et in store.edge_types},
                          batch_dict={"note": bb["batch_dict"]["note"]}, batch_size=s_.batch_size,
                          neighbor_mask_node={"note": bb["neighbor_mask_node"]["note"]},
                          neighbor_mask_edge={et: bb["neighbor_mask_edge"][et] for et in store.edge_types}))
    assert_close(outs[1], outs[0], 1e-5, "pooled vs padded batch")
