#!/usr/bin/env python3
"""Last replay of graph_window_probe.py: start of the long kernel and of every m kernel relative to the end of A."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
A = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("Cijk")][-1]
t0 = rows[A]["e"]
ms, L = [], None
for r in rows[A + 1:]:
    n = r["Kernel_Name"]
    if "sin" in n:
        ms.append((r["s"] - t0) / 1e3)
    elif "sleep" in n.lower() or "spin" in n.lower():
        L = ((r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3)
print(f"long kernel: starts {L[0]:+.0f} us after A, runs {L[1]:.0f} us" if L else "long kernel not found: " + str(sorted({r['Kernel_Name'][:40] for r in rows[A+1:]})))
print("m starts (us after A):", " ".join(f"{t:.0f}" for t in ms))
