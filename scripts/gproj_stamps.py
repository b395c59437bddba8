#!/usr/bin/env python3
"""In-kernel time stamps of the persistent grouped-projection kernels (a library built with -DGP_STAMPS, passed as AGNN_LIB;
WHAT=fwd | dx): per wave s_memtime at entry, after the prologue, when the first image is ready, and per stage after the
tiles / after the first barrier / after the image is rewritten.  Prints the median over 32 workgroups of each interval, per wave."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from analysisgnn_amd import _lib
from analysisgnn_amd.heads import grouped_projection
dev = "cuda:0"
classes = list(bench.TASK_DICT.values())
offs = [0]
for c in classes:
    offs.append(offs[-1] + c)
N, K = int(os.environ.get("N", "16000")), 64
a = torch.randn(N, len(classes) * K, device=dev)
w = torch.randn(offs[-1], K, device=dev) * 0.1
b = torch.randn(offs[-1], device=dev)
WHAT = os.environ.get("WHAT", "fwd")                      # fwd | dx
if WHAT == "fwd":
    with torch.no_grad():
        for _ in range(3):
            grouped_projection(a, w, b, offs, K)
else:
    a.requires_grad_(True)
    g = torch.randn(N, offs[-1], device=dev)
    for _ in range(3):
        grouped_projection(a, w, b, offs, K).backward(g)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["AGNN_LIB"])
buf = (ctypes.c_ulonglong * (32 * 12 * 16))()
assert lib.agnn_debug_gproj_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(32, 12, 16).astype(np.int64)      # [workgroup][wave][stamp]
blocks = -(-(-(-N // 32)) // 256)                         # row blocks of the first workgroups (persistent: one workgroup per CU)
if WHAT == "fwd":
    stages = blocks * 2                                   # two chunks of groups per row block (21 heads)
    n, nm = 3 + 3 * (stages - 1) + 1, 3 + 3 * (stages - 1)
    first = ["prologue", "first image"]
else:                                                     # (the forward of the same call left later stamps behind: cut)
    n, nm = 2 + 3 * (blocks - 1) + 1, 2 + 3 * (blocks - 1)
    first = ["prologue + first image"]
print(f"{WHAT}: s_memtime ticks (core clocks), median over 32 workgroups; multipliers (waves 0..7): {n} stamps, movers (8..11): {nm}")
names = first + [f"stage {k // 3}: {('tiles / units', 'barrier', 'image rewrite')[k % 3]}" for k in range(n - 1 - len(first))]
d = np.diff(st[:, :8, :n], axis=2)
for k in range(n - 1):
    print(f"  {names[k]:28s} " + " ".join(f"{np.median(d[:, w, k]):7.0f}" for w in range(8)))
names = first + [f"stage {k // 3}: {('loads issued', 'barrier', 'image rewrite')[k % 3]}" for k in range(nm - 1 - len(first))]
d = np.diff(st[:, 8:, :nm], axis=2)
for k in range(nm - 1):
    print(f"  {names[k]:28s} " + " ".join(f"{np.median(d[:, w, k]):7.0f}" for w in range(4)))
print(f"  whole (entry -> last stamp), median over workgroups: {np.median(st[:, :8, n - 1].max(axis=1) - st[:, :, 0].min(axis=1)):.0f}")
