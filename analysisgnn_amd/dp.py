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


class FlatGradBuffer:
    """All gradients of `params` in one contiguous fp32 buffer; `.grad` of each parameter is a view."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        o = 0
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("FlatGradBuffer: fp32 parameters on one device expected")
            p.grad = self.flat[o:o + p.numel()].view_as(p)
            o += p.numel()

    def zero(self) -> None:
        self.flat.zero_()

    def all_reduce_mean(self, world: Optional[int] = None) -> None:
        """SUM over ranks then divide: the same mean DDP applies."""
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(dist.get_world_size())

    def clip_norm_(self, max_norm: float) -> torch.Tensor:
        """clip_grad_norm_ on the flat view (one norm, no per-parameter launches, no host sync)."""
        total = torch.linalg.vector_norm(self.flat)
        self.flat.mul_(torch.clamp(max_norm / (total + 1e-6), max=1.0))
        return total


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
