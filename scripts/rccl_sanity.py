#!/usr/bin/env python3
"""One-rank RCCL sanity check (backend "nccl" on ROCm): process group, all-reduce of a flat gradient buffer, barrier.
The multi-GPU scaling run itself is the driver's; this only proves the code path imports and executes on the box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from analysisgnn_amd import dp
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
m = torch.nn.Linear(1024, 1024).cuda()
flat = dp.FlatGradBuffer(m.parameters())
m(torch.randn(8, 1024, device="cuda")).sum().backward()
before = flat.flat.clone()
dist.all_reduce(flat.flat, op=dist.ReduceOp.SUM)
dp.barrier_and_sync()
assert torch.equal(before, flat.flat)
print("rccl one-rank all_reduce ok:", flat.flat.numel(), "floats; max_over_ranks ->", dp.max_over_ranks(1.5))
dist.destroy_process_group()
