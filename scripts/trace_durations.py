#!/usr/bin/env python3
"""Print per-dispatch durations (us) of kernels whose name contains PATTERN from a rocprofv3 kernel-trace CSV."""
import csv, glob, sys
pat, path = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(path))[0]
rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    print(f'{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:.2f},{r["Kernel_Name"][:60].replace(",", ";")}')
