#!/usr/bin/env python3
"""Host time spent inside each hipGraph replay() call of the captured training step (per graph object).
AGNN_SERIAL=1: sequence branch on the main stream (single-stream graph)."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
if os.environ.get("AGNN_SERIAL") == "1":
    import analysisgnn_amd.encoders as enc
    enc._HybridMixin.overlap_sequence_branch = False
orig_replay = torch.cuda.CUDAGraph.replay
acc = collections.defaultdict(list)
stamps = []
def timed_replay(self):
    t0 = time.perf_counter()
    orig_replay(self)
    t1 = time.perf_counter()
    acc[id(self)].append(t1 - t0)
    stamps.append((id(self), t0, t1))
torch.cuda.CUDAGraph.replay = timed_replay
import bench
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "50", "--warmup", "5"] + sys.argv[1:]
bench.main()
for k, v in acc.items():
    v2 = v[len(v) // 2:]
    print(f"graph {k}: {len(v)} replays, host time {sum(v2) / len(v2) * 1e3:.3f} ms per call (second half)", file=sys.stderr)
big = max(acc, key=lambda k: sum(acc[k]))
calls = [s for s in stamps if s[0] == big]
per = [(calls[i + 1][1] - calls[i][1]) * 1e3 for i in range(len(calls) // 2, len(calls) - 1)]
print(f"big graph: start-to-start {sum(per) / len(per):.3f} ms", file=sys.stderr)
