#!/usr/bin/env python3
"""Is a hipMemsetAsync node ordered before its consumer when a ONE-stream captured graph is replayed (ROCm 7.2)?

Round 1 saw two faults under one trigger (the training step captured on one stream, ~30 replays): an aperture violation in
torch's embedding backward (rocprim partition, fed by zero-filled buffers) and an out-of-range write of the CSR build's
k_scatter, whose counters were zeroed by a hipMemsetAsync node.  This probe isolates the suspected mechanism without any
out-of-range access: per replay  [memset(buf, 0)] -> [err += count(buf != 0)] -> [buf.fill_(7)]  (the fill plays the later
node that recycles the block in the capture pool).  err must stay 0.  Variants: memset by hipMemsetAsync (a memset node)
vs by a fill kernel; one stream vs a forked capture.  Run once; prints one JSON line per variant."""
import ctypes
import json
import sys

import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int


def run(variant: str, n: int, replays: int, fork: bool):
    dev = torch.device("cuda", 0)
    buf = torch.full((n,), 7, dtype=torch.int32, device=dev)
    err = torch.zeros((), dtype=torch.int64, device=dev)
    pad = torch.randn(1 << 20, device=dev)
    side = torch.cuda.Stream(device=dev)

    def body():
        s = torch.cuda.current_stream(dev)
        if variant == "memset_node":
            rc = hip.hipMemsetAsync(buf.data_ptr(), 0, n * 4, s.cuda_stream)
            assert rc == 0, rc
        else:
            buf.fill_(0)
        if fork:                                   # an unrelated branch, as the sequence branch of the step
            side.wait_stream(s)
            with torch.cuda.stream(side):
                pad.mul_(1.0001)
        err.add_((buf != 0).sum())
        buf.fill_(7)
        if fork:
            s.wait_stream(side)

    warm = torch.cuda.Stream(device=dev)
    warm.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(warm):
        body()
    torch.cuda.current_stream(dev).wait_stream(warm)
    torch.cuda.synchronize()
    err.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    for _ in range(replays):
        g.replay()
    torch.cuda.synchronize()
    return {"variant": variant, "forked": fork, "n_words": n, "replays": replays, "nonzero_words_seen": int(err.item())}


if __name__ == "__main__":
    for n in (130_002, 1 << 22):
        for variant in ("memset_node", "fill_kernel"):
            for fork in (False, True):
                print(json.dumps(run(variant, n, 300, fork)))
                sys.stdout.flush()
