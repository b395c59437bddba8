#!/usr/bin/env python3
"""Replays the stream assignment the ROCm hipGraph executor gives a captured graph (clr hip_graph_internal: depth-first from
the roots; a node's first edge keeps its stream, every further edge takes the next one modulo DEBUG_HIP_FORCE_GRAPH_QUEUES)
on a DOT dump (AGNN_GRAPH_DOT=... bench.py), and prints node -> stream with the fork points: two independent chains that
land on one stream are serialised at replay whatever the dependencies say.   usage: graph_streams.py <dot> [n_streams]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
names = {}
raw = {int(m.group(1)): m.group(3) for m in re.finditer(r'"graph_0_node_(\d+)"\[[^\]]*label="(\d+)\n([^\n"]*)', txt)}
dem = subprocess.run(["c++filt", "-p"], input="\n".join(raw[i] for i in sorted(raw)), capture_output=True, text=True).stdout.split("\n")
for i, d in zip(sorted(raw), dem):
    names[i] = d.replace("(anonymous namespace)::", "").replace("at::native::", "")[:50]
edges, preds = {}, {}
for a, b in re.findall(r'"graph_0_node_(\d+)" -> "graph_0_node_(\d+)"', txt):
    edges.setdefault(int(a), []).append(int(b))
    preds.setdefault(int(b), []).append(int(a))
stream = {}
sys.setrecursionlimit(10000)


def visit(n, s):
    if n in stream:
        return
    stream[n] = s
    for c in edges.get(n, []):
        visit(c, s)
        s = (s + 1) % N


s = 0
for r in sorted(n for n in names if n not in preds):
    visit(r, s)
    s = (s + 1) % N
for i in sorted(names):
    p = preds.get(i, [])
    note = "" if p == [i - 1] else f"   <- {p}"
    print(f"{i:4d} s{stream.get(i, -1)} {names[i]}{note}")
