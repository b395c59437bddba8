"""Which switch makes FlatGradBuffer(views=True) + off-chain weight gradients differ from the plain schedule (one-off diagnosis)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from analysisgnn_amd import dp, graph, encoders
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
    labels = torch.stack([torch.randint(0, c, (I["batch_size"],), generator=torch.Generator().manual_seed(7 * seed + i)).to(dev) for i, c in enumerate(tasks.values())])
    batches.append((I, labels))
torch.manual_seed(0)
model = TorchAnalysisGNN(g.metadata(), 25, 256, 128, tasks, 3, dropout=0.0, use_jk=False, logit_fusion=False).to(dev).train()
clf = MultiTaskLoss(list(tasks)).to(dev)
both = torch.nn.ModuleDict({"m": model, "c": clf})
params = [p for p in both.parameters() if p.requires_grad]
graph.index_cache_enabled = False
flat = dp.FlatGradBuffer(params, views=True)

def run(overlap, defer, late, nb=2):
    dp.enable_wgrad_overlap(overlap, "all")
    dp.defer_weight_grads(defer)
    encoders.LATE_SEQUENCE_BACKWARD = late
    flat.zero()
    for I, labels in batches[:nb]:
        x = model.encode(I["pitch_spelling"], I["key_signature"], I["x_dict"], I["edge_index_dict"], I["batch_dict"], I["batch_size"], None, None)
        logits, offs, _ = model.forward_clf_fused(x)
        loss, _ = training_loss(logits, offs, labels, x, 0.1, 0.1, -1, task_params=clf.weights())
        loss.backward()
    join_wgrad()
    torch.cuda.synchronize()
    return [p.grad.detach().clone() for p in params]

names = [n for n, _ in both.named_parameters()]
for nb in (1, 2):
    g0 = run(False, False, True, nb)
    g0b = run(False, False, True, nb)
    print(f"nb={nb} plain twice: max diff", max(float((a - b).abs().max()) for a, b in zip(g0, g0b)))
    for overlap, defer, late in [(True, False, True), (False, True, True), (False, True, False), (True, True, True)]:
        g1 = run(overlap, defer, late, nb)
        bad = [(n, float((a - b).abs().max())) for n, a, b in zip(names, g0, g1) if not torch.equal(a, b)]
        print(f"nb={nb} overlap={overlap} defer={defer} late={late}: {len(bad)} differ", bad[:6])
