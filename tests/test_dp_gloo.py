"""The N>1 path on CPU: two processes, `gloo` backend, 127.0.0.1 rendezvous.  Checks the data-parallel
harness (analysisgnn_amd/dp.py): disjoint subgraph shards, flat gradient buffer, SUM all-reduce / world,
max-over-ranks timing reduction — against a single-process run over the union of the shards.
The model here is the CPU oracle encoder (tests may use oracle/): the HIP path needs a GPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _OracleModel(torch.nn.Module):
    """Thin nn.Module around the functional oracle HybridGNN (parameters live in a ParameterDict-like list)."""

    def __init__(self, state, metadata, layers):
        super().__init__()
        self.names = list(state.keys())
        self.params = torch.nn.ParameterList([torch.nn.Parameter(v.clone()) for v in state.values()])
        self.metadata, self.layers = metadata, layers

    def forward(self, I):
        from oracle import encoders_ref as E
        P = dict(zip(self.names, self.params))
        return E.hybrid_gnn(P, "", self.metadata, self.layers, I["x_dict"], I["edge_index_dict"], I["batch_dict"], I["batch_size"])


def _make(seed_graphs, H=8):
    from analysisgnn_amd.encoders import HybridGNN
    from analysisgnn_amd.synth import collate, make_score_graph, torch_inputs
    g = collate([make_score_graph(seed=s, n_notes=24) for s in seed_graphs])
    torch.manual_seed(0)
    ref = HybridGNN(metadata=g.metadata(), input_channels=H, hidden_channels=H, num_layers=2, dropout=0.0)
    state = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    return _OracleModel(state, g.metadata(), 2), torch_inputs(g, in_channels=H, seed=sum(seed_graphs))


def _worker(rank, world, port, ret):
    from analysisgnn_amd import dp
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    r, l, w = dp.init_distributed("gloo")
    assert (r, w) == (rank, world)
    units = dp.shard_units(4, rank, world)                      # 4 subgraphs over 2 ranks: {0,2} and {1,3}
    model, I = _make(units)
    flat = dp.FlatGradBuffer(model.parameters())
    for _ in range(2):                                          # second pass: zero() really clears
        flat.zero()
        out = model(I)
        (out.pow(2).sum() / 48).backward()                      # sum over this rank's 48 target notes / per-rank count
        flat.all_reduce_mean()
    # the same step with the gradient message cut in two buckets (dp.plan_parameters(late=...): the late parameters at the end of
    # the buffer, the early bucket shipped asynchronously while "the rest of backward" — here: nothing — runs): bit-identical
    plist = list(model.parameters())
    late = plist[:2]                                            # any subset: plan_parameters moves it to the end
    params, tight = dp.plan_parameters(model, late=late)
    assert [id(p) for p in params[-2:]] == [id(p) for p in late]
    fb = dp.FlatGradBuffer(params, views=False, tight=tight, late=late)
    fb.zero()
    (model(I).pow(2).sum() / 48).backward()
    work = fb.all_reduce_early_async()
    assert work is not None and 0 < fb.cut < fb.flat.numel()
    fb.all_reduce_late_and_finish(work)
    by_param = {id(p): p.grad.clone() for p in params}
    fu = dp.FlatGradBuffer(params, views=False, tight=tight)
    fu.zero()
    (model(I).pow(2).sum() / 48).backward()
    fu.all_reduce_mean()
    same = all(torch.equal(by_param[id(p)], p.grad) for p in params)
    dp.barrier_and_sync()
    t = dp.max_over_ranks(float(rank + 1))
    ret[rank] = (units, flat.flat.clone(), t, same)
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_gradient_mean_matches_single_process():
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    (u0, g0, t0, same0), (u1, g1, t1, same1) = ret[0], ret[1]
    assert same0 and same1                                       # two-bucket all-reduce == one-message all-reduce, bit for bit
    assert sorted(u0 + u1) == [0, 1, 2, 3] and not set(u0) & set(u1)
    assert torch.equal(g0, g1)                                   # replicas hold identical averaged gradients
    assert t0 == t1 == 2.0                                       # MAX over ranks
    # single process: mean over the two shards' losses == the DP average
    from analysisgnn_amd import dp
    acc = None
    for units in (u0, u1):
        model, I = _make(units)
        flat = dp.FlatGradBuffer(model.parameters())
        (model(I).pow(2).sum() / 48).backward()
        acc = flat.flat.clone() if acc is None else acc + flat.flat
    assert torch.allclose(g0, acc / 2, rtol=1e-5, atol=1e-7)


def test_flat_buffer_views_and_clip():
    from analysisgnn_amd import dp
    m = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Linear(4, 2))
    flat = dp.FlatGradBuffer(m.parameters())
    m(torch.ones(5, 3)).sum().backward()
    assert all(p.grad.data_ptr() >= flat.flat.data_ptr() for p in m.parameters())
    n = float(torch.linalg.vector_norm(flat.flat))
    total = flat.clip_norm_(0.5)
    assert abs(float(total) - n) < 1e-6 and float(torch.linalg.vector_norm(flat.flat)) <= 0.5 + 1e-5
    flat.zero()
    assert all(float(p.grad.abs().sum()) == 0 for p in m.parameters())
    assert dp.shard_units(7, 1, 3) == [1, 4]


def test_flat_adamw_matches_torch_adamw():
    """FlatAdamW (one flat buffer, a handful of launches) follows torch.optim.AdamW's update rule."""
    import copy
    from analysisgnn_amd import dp
    torch.manual_seed(0)
    a = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    b = copy.deepcopy(a)
    ref = torch.optim.AdamW(a.parameters(), lr=5e-3, weight_decay=5e-3)
    flat = dp.FlatGradBuffer(b.parameters(), views=False)
    opt = dp.FlatAdamW(b.parameters(), flat, lr=5e-3, weight_decay=5e-3)
    x = torch.randn(11, 5)
    for _ in range(4):
        ref.zero_grad()
        a(x).pow(2).sum().backward()
        ref.step()
        flat.zero()
        b(x).pow(2).sum().backward()
        flat.all_reduce_mean()
        opt.step()
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.allclose(p, q, rtol=1e-4, atol=1e-6)
