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
# 1. what the default (global) capture-error mode does with the watchdog thread alive: a bare capture of a few launches
x = torch.ones(1 << 20, device="cuda:0")
for mode in ("global", "thread_local"):
    ok, why = True, ""
    try:
        for rep in range(5):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=mode):
                y = x * 2.0
                for _ in range(200):
                    y = y + 1.0
            g.replay()
        torch.cuda.synchronize()
    except Exception as e:                       # noqa: BLE001
        ok, why = False, f"{type(e).__name__}: {str(e)[:160]}"
    print(f"[nccl_capture_check] capture_error_mode={mode}: {'5 captures + replays ok' if ok else 'FAILED ' + why}", flush=True)
# 2. the bench step itself (bench.py picks thread_local whenever a process group exists)
import bench
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-other", "--steps", "20", "--warmup", "3"] + os.environ.get("NCC_ARGS", "").split()
bench.main()
dist.destroy_process_group()
