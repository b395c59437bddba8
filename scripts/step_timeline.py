#!/usr/bin/env python3
"""One training step's kernel timeline out of a `rocprofv3 --kernel-trace` CSV: start (us from the step's first kernel),
duration, hardware queue, kernel name — to see which dependent chain the step's wall time follows.  A step ends with
k_adamw; the one before last of the trace is printed (a hipGraph replay).  usage: step_timeline.py <kernel_trace.csv> [min_us]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("void ", "").replace("agnn::", "").replace("(anonymous namespace)::", "")
    r["n"] = re.sub(r"\(.*", "", n)[:56]
rows.sort(key=lambda r: r["s"])
ends = [i for i, r in enumerate(rows) if "k_adamw" in r["n"]]
a, b = ends[-3], ends[-2]
step = rows[a + 1:b + 1]
t0 = step[0]["s"]
print(f"step: {len(step)} kernels, {(step[-1]['e'] - t0) / 1e3:.1f} us from first start to last end, "
      f"{sum(r['e'] - r['s'] for r in step) / 1e3:.1f} us of kernel time")
for r in step:
    d = (r["e"] - r["s"]) / 1e3
    if d >= min_us:
        print(f"{(r['s'] - t0) / 1e3:8.1f} {d:7.1f} q{r['Queue_Id']} {r['n']}")
