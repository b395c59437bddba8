#!/usr/bin/env python3
"""Kernel sequence of the last full step in a rocprofv3 kernel-trace CSV (start offset, duration, #kernels running, name)."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "k_mtce"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void ", "", n)
    n = re.sub(r"at::native::", "", n)
    m = re.match(r"(Cijk_\w+?_MT\d+x\d+x\d+)", n)
    if m: return m.group(1)
    return n[:70]
ends = []
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ends = [x for x in ends if x > s]
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} {len(ends) + 1}  {short(r['Kernel_Name'])}")
    ends.append(e)
