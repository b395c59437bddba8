#!/usr/bin/env python3
"""In-kernel time stamps of the persistent grouped-projection forward (a library built with -DGP_STAMPS, passed as AGNN_LIB):
per workgroup s_memtime at entry, after the prologue, when the first chunk image is ready, and per stage after the tiles /
after the first barrier / after the chunk image is rewritten.  Prints the median over workgroups of each interval."""
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
n = int((st[0, 0] > 0).sum())
nm = int((st[0, 8] > 0).sum())
print(f"s_memtime ticks (core clocks), median over 32 workgroups; multipliers (waves 0..7): {n} stamps, movers (8..11): {nm}")
first = ["prologue", "first image"] if WHAT == "fwd" else ["prologue + first image"]
names = first + [f"stage {k // 3}: {('tiles', 'barrier', 'image rewrite')[k % 3]}" for k in range(n - 1 - len(first))]
d = np.diff(st[:, :8, :n], axis=2)
for k in range(n - 1):
    print(f"  {names[k]:24s} " + " ".join(f"{np.median(d[:, w, k]):7.0f}" for w in range(8)))
names = first + [f"stage {k // 3}: {('loads issued', 'barrier', 'image rewrite')[k % 3]}" for k in range(nm - 1 - len(first))]
d = np.diff(st[:, 8:, :nm], axis=2)
for k in range(nm - 1):
    print(f"  {names[k]:24s} " + " ".join(f"{np.median(d[:, w, k]):7.0f}" for w in range(4)))
print(f"  whole (entry -> last stamp), median over workgroups: {np.median(st[:, :8, n - 1].max(axis=1) - st[:, :, 0].min(axis=1)):.0f}")
