#!/usr/bin/env python3
"""Timeline of the LAST captured-step replay in a rocprofv3 kernel-trace CSV: per stream busy time, and the gaps of the
busiest stream with what ran on the other streams meanwhile.  timeline.py TRACE.csv [marker_kernel_substring]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "k_mtce"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
# one step = kernels between two consecutive marker kernels (steady state: take the last full interval)
a, b = idx[-2], idx[-1]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
span = (int(step[-1]["End_Timestamp"]) - t0) / 1e3
print(f"step span {span:.1f} us, {len(step)} kernels")
by = collections.defaultdict(list)
for r in step:
    by[r.get("Stream_Id", r.get("Queue_Id"))].append(r)
for s, rs in by.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e3
    print(f"stream {s}: {len(rs)} kernels, busy {busy:.1f} us, first at {(int(rs[0]['Start_Timestamp'])-t0)/1e3:.1f}, last end {(int(rs[-1]['End_Timestamp'])-t0)/1e3:.1f}")
main = max(by.values(), key=len)
prev_end = int(main[0]["Start_Timestamp"])
gaps = []
for r in main:
    s = int(r["Start_Timestamp"])
    if s - prev_end > 3000:
        gaps.append(((prev_end - t0) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:60]))
    prev_end = max(prev_end, int(r["End_Timestamp"]))
print("gaps > 3 us on the main stream (start, length, next kernel):")
for g in gaps:
    print(f"  at {g[0]:8.1f} us  gap {g[1]:7.1f} us  before {g[2]}")
print(f"sum of gaps {sum(g[1] for g in gaps):.1f} us")
