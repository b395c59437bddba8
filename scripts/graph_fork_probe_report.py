#!/usr/bin/env python3
"""Reads the kernel trace of graph_fork_probe.py: for the last replay, start of the first m / s kernel relative to A's end."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
gemm = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("Cijk")]
last = gemm[-1]
A = rows[last]
rest = rows[last + 1:]
def kind(r):
    n = r["Kernel_Name"]
    return "m" if "sin" in n else ("s" if "cos" in n else ("join" if "add" in n.lower() else "?"))
t0 = A["e"]
print(f"A: {(A['e'] - A['s']) / 1e3:.1f} us on q{A['Queue_Id']}")
seen = {}
for r in rest:
    k = kind(r)
    seen.setdefault(k, []).append(r)
for k, rs in seen.items():
    print(f"{k}: {len(rs)} kernels on queues {sorted(set(x['Queue_Id'] for x in rs))}, first starts {(rs[0]['s'] - t0) / 1e3:+.1f} us after A ends, "
          f"last ends {(rs[-1]['e'] - t0) / 1e3:+.1f} us, each {(rs[0]['e'] - rs[0]['s']) / 1e3:.1f} us")
