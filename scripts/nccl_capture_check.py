#!/usr/bin/env python3
"""hipGraph capture of the step with an initialised RCCL process group in the process (one rank): the NCCL watchdog
thread issues HIP calls of its own, which invalidates a capture made in the default (global) capture-error mode."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(1024, device="cuda:0")
dist.all_reduce(t)
torch.cuda.synchronize()
import bench
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "20", "--warmup", "3"]
bench.main()
dist.destroy_process_group()
