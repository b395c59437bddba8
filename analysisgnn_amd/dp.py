"""Single-node data parallelism: one process per GPU, gradients all-reduced over RCCL/xGMI.

Replaces what Lightning's implicit DDP does for the reference
(analysisgnn/train/train_analysisgnn.py:138-146, :246-255): sampled subgraphs are independent
units (block-diagonal batches, no cross edges), so ranks take disjoint subgraphs, run the hot
path locally and exchange only gradients — ONE collective per step on a flat fp32 buffer.
MI355X notes: ~5 M parameters = ~20 MB; a node's 8 GPUs are fully connected by xGMI links, so
one large all-reduce (all links busy) beats per-parameter messages; parameters' `.grad` are views
into the flat buffer, so backward writes straight into the message (no pack/unpack copies).
Works on the `gloo` backend with CPU tensors too (that is how the N>1 path is tested without GPUs).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> tuple:
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # AGNN_DIST_BACKEND=gloo lets several ranks share ONE GPU (rehearsing the N>1 flow on a 1-GPU box)
            backend = os.environ.get("AGNN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


def shard_units(n_units: int, rank: int, world: int) -> List[int]:
    """Subgraph ids of this rank: {i : i mod world == rank} (DistributedSampler semantics)."""
    return list(range(rank, n_units, world))


def _aligned_offsets(sizes: Sequence[int], align: int = 4, tight: Optional[Sequence[bool]] = None) -> List[int]:
    """Start offset of every slot (+ the end): slots start on `align`-float boundaries, except that a slot marked `tight`
    starts right where the previous one ends (members of a group that is consumed as one concatenated operand)."""
    offs, end = [], 0
    for i, k in enumerate(sizes):
        start = end if (tight is not None and tight[i] and i > 0) else (end + align - 1) // align * align
        offs.append(start)
        end = start + k
    offs.append((end + align - 1) // align * align)
    return offs


def plan_parameters(model: torch.nn.Module, late: Iterable[torch.nn.Parameter] = ()):
    """(params, tight): the trainable parameters in the order the flat buffers should hold them, and the ids of those that
    must sit right behind their predecessor.  `late`: parameters whose gradients the backward pass produces LAST (the input
    layers) — placed at the END of the buffer so that everything before them is one contiguous message that can be all-reduced
    while they are still being computed (FlatGradBuffer(late=...), all_reduce_early_async).  Modules may expose `adjacent_parameter_groups()` -> lists of parameters
    that the hot path consumes concatenated (task-head layers, GRU direction pairs): laid out back to back, the cat /
    stack is a view of the flat parameter buffer instead of a launch per step (params.cat_rows / stack_rows).  All other
    parameters follow in `model.parameters()` order."""
    groups = []
    for m in model.modules():
        fn = getattr(m, "adjacent_parameter_groups", None)
        if fn is not None:
            groups.extend([[p for p in g] for g in fn()])
    late_ids = {id(p) for p in late}
    seen, params, tight = set(), [], set()
    for g in groups:
        if any(id(p) in late_ids for p in g):
            raise ValueError("plan_parameters: an adjacency group cannot hold late parameters")
        g = [p for p in g if p.requires_grad and id(p) not in seen]
        for i, p in enumerate(g):
            seen.add(id(p))
            params.append(p)
            if i > 0:
                tight.add(id(p))
    for want_late in (False, True):
        for p in model.parameters():
            if p.requires_grad and id(p) not in seen and (id(p) in late_ids) == want_late:
                seen.add(id(p))
                params.append(p)
    return params, tight


class FlatGradBuffer:
    """All gradients of `params` in one contiguous fp32 buffer (the all-reduce message).

    Two modes:
      * `views=True`  — `.grad` of each parameter is a view into the buffer, backward accumulates straight into
        the message (no pack step; costs one tiny accumulate launch per parameter, fine when GPU-bound);
      * `views=False` — autograd hands over fresh gradient tensors (no per-parameter accumulate launches, which
        dominate when a step is launch-bound: ~130 parameters here) and `pack()` gathers them with ONE cat;
        `.grad` then become views of the buffer so the optimizer sees the reduced / clipped values.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], views: bool = True, tight: Optional[set] = None,
                 late: Iterable[torch.nn.Parameter] = ()):
        """`late`: the parameters of the second bucket (must be the LAST ones of `params`, dp.plan_parameters(model, late=...)
        puts them there): `pack("early")` / `pack("late")` gather the two buckets separately, `all_reduce_early_async` ships
        the first while the backward pass still works on the second."""
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        self.sizes = [p.numel() for p in self.params]
        # every slot starts on a 16-byte boundary (the HIP kernels read parameters / gradients as float4), except the
        # members of an adjacency group (plan_parameters), which are consumed through one view of the whole group
        self.offsets = _aligned_offsets(self.sizes, tight=[tight is not None and id(p) in tight for p in self.params])
        n = self.offsets[-1]
        self.views = views
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("FlatGradBuffer: fp32 parameters on one device expected")
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self._pad = torch.zeros(4, dtype=torch.float32, device=dev)
        # a parameter that took no gradient this step (a layer that keeps no edge: encoders.HeteroConv) packs from here
        self._none = torch.zeros(max(self.sizes) if self.sizes else 1, dtype=torch.float32, device=dev)
        late_ids = {id(p) for p in late}
        self.n_early = len(self.params)
        if late_ids:
            flags = [id(p) in late_ids for p in self.params]
            self.n_early = flags.index(True) if True in flags else len(flags)
            if not all(flags[self.n_early:]) or sum(flags) != len(late_ids):
                raise ValueError("FlatGradBuffer: the late parameters must be the last ones of the buffer (dp.plan_parameters(model, late=...))")
        self.cut = self.offsets[self.n_early]                # flat[:cut] = early bucket, flat[cut:] = late bucket
        if views:
            self._assign_views()

    def _assign_views(self) -> None:
        for p, k, o in zip(self.params, self.sizes, self.offsets):
            p.grad = self.flat[o:o + k].view_as(p)

    def zero(self) -> None:
        if self.views:
            self.flat.zero_()
        else:
            for p in self.params:
                p.grad = None
            self._packed = set()

    def pack(self, part: Optional[str] = None) -> None:
        """views=False: gather the fresh gradients into the flat buffer (one launch; `part` = "early" / "late": one bucket, one
        launch each).  Also the join point of the weight-gradient stream (linear.enable_wgrad_overlap)."""
        from .linear import join_wgrad
        join_wgrad()
        if self.views:
            return
        done = getattr(self, "_packed", set())
        if not isinstance(done, set):
            done = {"early", "late"} if done else set()
        for name, lo, hi in (("early", 0, self.n_early), ("late", self.n_early, len(self.params))):
            if (part is not None and part != name) or name in done or lo == hi:
                done.add(name) if lo == hi else None
                continue
            parts = []
            for i in range(lo, hi):
                p, k, o, o_next = self.params[i], self.sizes[i], self.offsets[i], self.offsets[i + 1]
                parts.append(p.grad.reshape(-1) if p.grad is not None else self._none[:k])
                if o_next - o > k:
                    parts.append(self._pad[: o_next - o - k])
            torch.cat(parts, out=self.flat[self.offsets[lo]:self.offsets[hi]])
            for i in range(lo, hi):
                p, k, o = self.params[i], self.sizes[i], self.offsets[i]
                p.grad = self.flat[o:o + k].view_as(p)
            done.add(name)
        self._packed = done

    def all_reduce_mean(self, world: Optional[int] = None) -> None:
        """SUM over ranks then divide: the same mean DDP applies."""
        self.pack()
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(dist.get_world_size())

    def all_reduce_early_async(self):
        """Bucket 1 of 2 (what Lightning DDP's reverse-order buckets do, train/train_analysisgnn.py:246-255): gather and ship
        everything but the late parameters NOW — the collective runs on the communicator's stream beside whatever the caller
        queues next (the input layers' backward).  Returns the work handle for `all_reduce_late_and_finish` (None: one rank)."""
        self.pack("early")
        if dist.is_initialized() and dist.get_world_size() > 1 and self.cut > 0:
            return dist.all_reduce(self.flat[:self.cut], op=dist.ReduceOp.SUM, async_op=True)
        return None

    def all_reduce_late_and_finish(self, work) -> None:
        """Bucket 2 of 2 (small: the input layers), then wait for bucket 1 and divide: the buffer holds the mean, bit-identical
        to `all_reduce_mean` (SUM over ranks is element-wise: where a message is cut does not change a sum)."""
        self.pack("late")
        if dist.is_initialized() and dist.get_world_size() > 1:
            if self.cut < self.flat.numel():
                dist.all_reduce(self.flat[self.cut:], op=dist.ReduceOp.SUM)
            if work is not None:
                work.wait()
            self.flat.div_(dist.get_world_size())

    def clip_norm_(self, max_norm: float) -> torch.Tensor:
        """clip_grad_norm_ on the flat view (one norm, no per-parameter launches, no host sync)."""
        total = torch.linalg.vector_norm(self.flat)
        self.flat.mul_(torch.clamp(max_norm / (total + 1e-6), max=1.0))
        return total


class FlatAdamW:
    """AdamW over ONE flat parameter buffer (decoupled weight decay, bias correction — torch.optim.AdamW's update
    rule, reference optimizer: models/analysis.py:1380-1381).  Parameters are re-pointed at views of the buffer,
    gradients come from a `FlatGradBuffer`, so a step is a handful of whole-model elementwise launches instead of
    per-parameter lists: ~5 M parameters in ~130 tensors make the foreach path launch-bound."""

    def __init__(self, params: Iterable[torch.nn.Parameter], grads: FlatGradBuffer, lr=1e-3, betas=(0.9, 0.999),
                 eps=1e-8, weight_decay=1e-2):
        self.params = [p for p in params if p.requires_grad]
        assert [id(p) for p in self.params] == [id(p) for p in grads.params], "same parameter order as the gradient buffer"
        self.grads = grads
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        dev = self.params[0].device
        self.flat = torch.zeros(grads.offsets[-1], dtype=torch.float32, device=dev)       # same 16-byte-aligned layout
        for p, o in zip(self.params, grads.offsets):
            self.flat[o:o + p.numel()] = p.detach().reshape(-1)
            p.data = self.flat[o:o + p.numel()].view_as(p)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)

    @torch.no_grad()
    def step(self, max_norm: float = 0.0) -> None:
        """One AdamW update; `max_norm > 0` first clips the global gradient norm (clip_grad_norm_ semantics).
        Graph-capturable: the step counter and the bias corrections live on the device.  On a GPU the whole thing is
        the two launches of `agnn_adamw_f32`; on CPU tensors (gloo tests) the same arithmetic in torch ops."""
        if not hasattr(self, "_t"):
            self._t = torch.zeros((), dtype=torch.float32, device=self.flat.device)
        g = self.grads.flat
        if self.flat.is_cuda:
            from . import _lib
            lib = _lib.load()
            if not hasattr(self, "_ws"):
                self._ws = torch.empty(int(lib.agnn_adamw_workspace_bytes()), dtype=torch.uint8, device=self.flat.device)
                self.last_norm = torch.zeros((), dtype=torch.float32, device=self.flat.device)
            _lib.check(lib.agnn_adamw_f32(self.flat.data_ptr(), g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.flat.numel(),
                                          float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.wd),
                                          float(max_norm), self._t.data_ptr(), self.last_norm.data_ptr(), 0, self._ws.data_ptr(),
                                          self._ws.numel(), _lib.stream_ptr(self.flat.device)), "agnn_adamw_f32")
            return
        if max_norm > 0:
            self.grads.clip_norm_(max_norm)
        b1, b2 = self.betas
        self._t += 1.0
        bc1 = 1.0 - (b1 ** self._t)
        bc2 = 1.0 - (b2 ** self._t)
        self.flat.mul_(1.0 - self.lr * self.wd)
        self.m.mul_(b1).add_(g, alpha=1.0 - b1)
        self.v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
        denom = (self.v.sqrt() / bc2.sqrt()).add_(self.eps)
        self.flat.addcdiv_(self.m / bc1, denom, value=-self.lr)


def enable_wgrad_overlap(flag: bool = True, scope="all") -> None:
    """Issue weight-gradient GEMMs on their own HIP stream (joined in FlatGradBuffer.pack).  Requires gradients to be
    None when backward starts — FlatGradBuffer(views=False).zero() — see linear.py."""
    from . import linear
    linear.enable_wgrad_overlap(flag, scope)


def defer_weight_grads(flag: bool = True) -> None:
    """Weight / bias gradients of the projections on the backward pass's own stream are postponed to where that stream
    would otherwise idle (linear.defer_weight_grads); `FlatGradBuffer.pack` runs whatever is still pending."""
    from .linear import defer_weight_grads as _set
    _set(flag)


def fill_missing_grads(module_or_params) -> int:
    """Give every trainable parameter that took no gradient this step a ZERO gradient; returns how many.  The encoders prune
    structurally dead branches (hgt._HGTCore: a node type nothing reads; encoders.HeteroConv: a layer that keeps no edge hands
    zeros itself), where the reference's graph still reaches those parameters through empty index ops and gives zeros.
    `torch.optim.AdamW` SKIPS a parameter whose grad is None (no weight decay, no moment decay), so a stock optimizer
    reproduces the reference's update only after this call; FlatGradBuffer.pack + FlatAdamW substitute zeros themselves."""
    params = module_or_params.parameters() if isinstance(module_or_params, torch.nn.Module) else module_or_params
    n = 0
    for p in params:
        if p.requires_grad and p.grad is None:
            p.grad = torch.zeros_like(p)
            n += 1
    return n


def barrier_and_sync() -> None:
    if dist.is_initialized():
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def max_over_ranks(x: float) -> float:
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return x
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([x], dtype=torch.float64, device=dev)  # tiny scalar exchange, outside the timed region
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
