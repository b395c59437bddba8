#!/usr/bin/env python3
"""Median k_wgrad / k_wgrad_reduce / library durations per shape from a kernel-trace CSV of scripts/bench_wgrad.py
(the script launches, per shape and round: k_wgrad, k_wgrad_reduce, library GEMM, library column sum)."""
import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
groups, cur = [], None
for name, d in seq:
    if "k_wgradE" in name or name.startswith("k_wgrad(") or ("k_wgrad" in name and "reduce" not in name):
        cur = {"k": d, "r": None, "lib": []}
        groups.append(cur)
    elif "k_wgrad_reduce" in name and cur is not None:
        cur["r"] = d
    elif cur is not None and ("Cijk" in name or "reduce_kernel" in name):
        cur["lib"].append(d)
for i in range(0, len(groups), 6):
    g = groups[i:i + 6]
    print(f"shape#{i // 6}: k_wgrad {st.median(x['k'] for x in g):7.1f}  reduce {st.median(x['r'] for x in g if x['r']):6.1f}  "
          f"library {st.median(sum(x['lib']) for x in g):7.1f}")
