"""Timeline of ONE replayed step out of a rocprofv3 --kernel-trace CSV: start / end / duration per kernel, one column per HIP
queue.  usage: python scripts/step_timeline.py <..._kernel_trace.csv> [first kernel name substring = k_embed_cat_fwd]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "k_embed_cat_fwd"
marks = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
a, b = marks[-3], marks[-2]                      # a full step well inside the replayed stretch
# the step's first kernels (RNG advance, fills) sit just before the mark: start 3 kernels earlier
a, b = a - 3, b - 3
t0 = int(rows[a]["Start_Timestamp"])
qs = {}
print(f"# {b - a} kernels, {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
for r in rows[a:b]:
    q = qs.setdefault(r["Queue_Id"], len(qs))
    st, en = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if n.startswith("Cijk"):
        n = "LIB " + n.split("_MT")[1][:12]
    print(f"{st:8.1f} {en:8.1f} {en - st:7.1f} q{q} {'    ' * q}{n[:64]}")
