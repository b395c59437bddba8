#!/usr/bin/env python3
"""Per-step kernel breakdown from a rocprofv3 kernel_stats.csv of bench.py: stats_per_step.py FILE STEPS_PROFILED [TOP]."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
for r in rows[:top]:
    t = int(r["TotalDurationNs"]) / 1e3 / steps
    print(f"{t:8.1f} us/step {int(r['Calls']) / steps:6.1f} calls {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:110]}")
print(f"total {sum(int(r['TotalDurationNs']) for r in rows) / 1e3 / steps:.1f} us/step")
