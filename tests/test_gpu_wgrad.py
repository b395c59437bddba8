"""fp32-MFMA weight-gradient kernel (agnn_wgrad_f32) vs plain PyTorch fp32 (float64 accumulation on CPU as the
yardstick).  Tolerance 1e-4 relative to max(1,|ref|max); the MFMA is an exact fp32 fmaf chain, observed ~1e-6."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import assert_close  # noqa: E402

DEV = "cuda:0"


@pytest.mark.parametrize("n,out_f,in_f", [(2048, 2, 2), (4097, 64, 128), (16000, 256, 256), (16000, 256, 1024),
                                          (16000, 1344, 128), (5000, 130, 66), (16000, 690, 1344), (16001, 768, 256)])
def test_wgrad_matches_reference(n, out_f, in_f, monkeypatch):
    from analysisgnn_amd import linear
    from analysisgnn_amd.linear import weight_grad
    monkeypatch.setattr(linear, "MAX_OUT_IN", 1 << 40)   # exercise the kernel on every shape, not only the dispatched ones
    g = torch.Generator().manual_seed(n + out_f)
    dy = torch.randn(n, out_f, generator=g)
    x = torch.randn(n, in_f, generator=g)
    ref_w = (dy.double().t() @ x.double()).float()
    ref_b = dy.double().sum(0).float()
    dw, db = weight_grad(dy.to(DEV), x.to(DEV), True)
    assert_close(dw, ref_w, 1e-4, "dW")
    assert_close(db, ref_b, 1e-4, "db")
    dw2, none = weight_grad(dy.to(DEV), x.to(DEV), False)
    assert none is None and torch.equal(dw2, dw)                    # deterministic, bias optional


def test_results_land_in_given_slots():
    """dw_out / db_out: the two directions of a GRU layer write straight into the stacked [2, 3H, H] / [2, 3H] gradients."""
    from analysisgnn_amd.linear import weight_grad
    g = torch.Generator().manual_seed(3)
    dy = torch.randn(4096, 2 * 384, generator=g).to(DEV)
    x = torch.randn(4096, 2 * 128, generator=g).to(DEV)
    dw = torch.full((2, 384, 128), float("nan"), device=DEV)
    db = torch.full((2, 384), float("nan"), device=DEV)
    for d in range(2):
        r = weight_grad(dy[:, d * 384:(d + 1) * 384], x[:, d * 128:(d + 1) * 128], True, dw_out=dw[d], db_out=db[d])
        assert r[0].data_ptr() == dw[d].data_ptr() and r[1].data_ptr() == db[d].data_ptr()
        assert_close(dw[d], (dy[:, d * 384:(d + 1) * 384].double().t() @ x[:, d * 128:(d + 1) * 128].double()).float(), 1e-4)
        assert_close(db[d], dy[:, d * 384:(d + 1) * 384].double().sum(0).float(), 1e-4)
    small = weight_grad(dy[:100, :384], x[:100, :128], True, dw_out=dw[0], db_out=db[0])      # below MIN_ROWS: library, same slots
    assert small[0].data_ptr() == dw[0].data_ptr()
    assert_close(dw[0], (dy[:100, :384].double().t() @ x[:100, :128].double()).float(), 1e-4)


def test_strided_views_and_fallback():
    from analysisgnn_amd.linear import weight_grad
    g = torch.Generator().manual_seed(0)
    big = torch.randn(4096, 512, generator=g).to(DEV)
    dy, x = big[:, 128:256], big[:, 256:320]                          # column slices: ld = 512
    dw, db = weight_grad(dy, x, True)
    assert_close(dw, (dy.double().t() @ x.double()).float(), 1e-4)
    odd = torch.randn(4096, 153, generator=g).to(DEV)                 # odd width: library fallback, same answer
    dw3, _ = weight_grad(dy, odd, False)
    assert_close(dw3, (dy.double().t() @ odd.double()).float(), 1e-4)


def test_linear_module_gradients():
    from analysisgnn_amd.linear import Linear
    torch.manual_seed(0)
    m = Linear(64, 32).to(DEV)
    ref = torch.nn.Linear(64, 32).to(DEV)
    ref.load_state_dict(m.state_dict())
    x = torch.randn(3000, 64, device=DEV, requires_grad=True)
    xr = x.detach().clone().requires_grad_(True)
    m(x).pow(2).sum().backward()
    ref(xr).pow(2).sum().backward()
    assert_close(m.weight.grad, ref.weight.grad, 1e-4)
    assert_close(m.bias.grad, ref.bias.grad, 1e-4)
    assert_close(x.grad, xr.grad, 1e-4)


def test_wgrad_stream_overlap_gives_identical_gradients():
    """One training step of the C2-shaped model (small) with weight gradients on their own stream vs on the launching
    stream: every parameter gradient bit-identical (same kernels, same order of summation)."""
    from analysisgnn_amd import dp, linear
    from analysisgnn_amd.heads import multitask_cross_entropy
    from analysisgnn_amd.models import TorchAnalysisGNN
    from analysisgnn_amd.synth import make_batch, torch_inputs
    tasks = {"a": 4, "b": 50, "c": 185, "d": 2}
    g = make_batch(8, 300)
    I = torch_inputs(g, 25, "cuda:0", seed=0)
    labels = torch.stack([torch.randint(0, c, (I["batch_size"],), generator=torch.Generator().manual_seed(i)).to("cuda:0")
                          for i, c in enumerate(tasks.values())])
    grads = []
    for overlap in (False, True):
        torch.manual_seed(0)
        model = TorchAnalysisGNN(g.metadata(), 25, 256, 128, tasks, 2, dropout=0.0, use_jk=False, logit_fusion=False).to("cuda:0").train()
        flat = dp.FlatGradBuffer(model.parameters(), views=False)
        dp.enable_wgrad_overlap(overlap)
        try:
            flat.zero()
            x = model.encode(I["pitch_spelling"], I["key_signature"], I["x_dict"], I["edge_index_dict"], I["batch_dict"],
                             I["batch_size"], None, None)
            logits, offs, _ = model.forward_clf_fused(x)
            loss = 0.1 * x.pow(2).mean() + multitask_cross_entropy(logits, offs, labels, 0.1, -1).sum()
            loss.backward()
            flat.pack()
            torch.cuda.synchronize()
            grads.append(flat.flat.clone())
        finally:
            dp.enable_wgrad_overlap(False)
    assert torch.isfinite(grads[0]).all()
    assert grads[0].abs().max() > 0
    assert torch.equal(grads[0], grads[1])


def test_wgrad_overlap_with_preallocated_grads_stays_on_the_main_stream():
    """Parameters whose .grad already exists (FlatGradBuffer(views=True), gradient accumulation) make AccumulateGrad ADD on
    the main stream, which does not wait for the weight-gradient stream: such layers must not fork (linear._steals).
    The 153-wide note input (weight gradient computed on 154 columns) must come back as a contiguous tensor."""
    from analysisgnn_amd import dp, linear
    torch.manual_seed(0)
    dev = "cuda:0"
    lin = linear.Linear(153, 64).to(dev)
    buf = torch.zeros(5000, 156, device=dev)
    buf[:, :153] = torch.randn(5000, 153, device=dev)
    x = buf[:, :153]
    g = torch.randn(5000, 64, device=dev)
    ref_w = g.t() @ x
    ref_b = g.sum(0)
    for pre in (False, True):
        lin.weight.grad = torch.zeros_like(lin.weight) if pre else None
        lin.bias.grad = torch.zeros_like(lin.bias) if pre else None
        dp.enable_wgrad_overlap(True)
        try:
            assert linear._steals(lin.weight) == (not pre)
            lin(x).backward(g)
            linear.join_wgrad()
            torch.cuda.synchronize()
        finally:
            dp.enable_wgrad_overlap(False)
        assert lin.weight.grad.is_contiguous()
        assert_close(lin.weight.grad, ref_w, 1e-4, f"dW (pre-allocated grad: {pre})")
        assert_close(lin.bias.grad, ref_b, 1e-4, "db")


def test_weight_grad_odd_width_with_padded_rows():
    """in = 153 (25 features + two 64-wide embeddings) on rows padded to 156 floats: served by the kernel on 154 columns."""
    from analysisgnn_amd import linear
    torch.manual_seed(0)
    n = 4099
    buf = torch.zeros(n, 156, device="cuda:0")
    buf[:, :153] = torch.randn(n, 153, device="cuda:0")
    x = buf[:, :153]
    dy = torch.randn(n, 256, device="cuda:0")
    dw, db = linear.weight_grad(dy, x, True)
    ref = dy.double().t() @ x.double()
    assert dw.shape == (256, 153)
    assert_close(dw, ref.float(), 1e-5, "dw odd")
    assert_close(db, dy.double().sum(0).float(), 1e-5, "db")
    xc = x.contiguous()                                  # no spare column: library path
    dw2, _ = linear.weight_grad(dy, xc, False)
    assert_close(dw2, ref.float(), 1e-5, "dw odd (library)")
