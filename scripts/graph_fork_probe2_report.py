#!/usr/bin/env python3
"""Timeline of the last replay of graph_fork_probe2.py from a rocprofv3 kernel trace."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
gru = [i for i, r in enumerate(rows) if "k_gru_fwd" in r["Kernel_Name"]]
last_gru = gru[-1] if gru else None
# the A of this replay: the last big (8192-thread-row) GEMM before it
A = max(i for i in range(last_gru) if rows[i]["Kernel_Name"].startswith("Cijk") and int(rows[i]["Grid_Size_X"]) * int(rows[i]["Grid_Size_Y"]) > 30000 or False) if False else None
if last_gru is not None:
    cands = [i for i in range(last_gru) if rows[i]["Kernel_Name"].startswith("Cijk")]
    A = cands[-2]   # two GEMMs precede the GRU kernel in a replay: A (the probe's) and the GRU layer's input projection
else:
    A = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("Cijk")][-1]
t0 = rows[A]["e"]
print(f"A ends at 0; (A took {(rows[A]['e'] - rows[A]['s']) / 1e3:.0f} us on q{rows[A]['Queue_Id']})")
first = {}
for r in rows[A + 1:A + 80]:
    n = r["Kernel_Name"]
    tag = "G(gru)" if "k_gru_fwd" in n else ("s" if "cos" in n else ("m" if "sin" in n else ("gi-gemm" if n.startswith("Cijk") else ("join" if "add" in n.lower() else n[:20]))))
    if tag not in first or tag in ("G(gru)", "join", "gi-gemm"):
        first[tag] = 1
        print(f"  first {tag:8s} q{r['Queue_Id']}  start {(r['s'] - t0) / 1e3:8.1f}  dur {(r['e'] - r['s']) / 1e3:7.1f}")
    if tag == "join":
        break
