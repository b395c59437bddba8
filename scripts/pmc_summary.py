#!/usr/bin/env python3
"""Median counter value per (kernel, counter) from a rocprofv3 --pmc counter_collection CSV."""
import csv, glob, statistics, sys
pat, path = sys.argv[1], sys.argv[2]
acc = {}
for f in sorted(glob.glob(path, recursive=True)):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            k = (r["Kernel_Name"][:70].replace(",", ";"), r["Counter_Name"], r["Grid_Size"])
            acc.setdefault(k, []).append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k[0]} | grid {k[2]} | {k[1]:28s} median {statistics.median(v):16.1f}  n={len(v)}")
