import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_mtce(" in r["Kernel_Name"]]
step = rows[idx[-2]:idx[-1]]
t0 = int(step[0]["Start_Timestamp"])
cnt = collections.Counter(r["Queue_Id"] for r in step)
print("kernels per Queue_Id in the last step:", dict(cnt))
prev = None
for r in step:
    q = r["Queue_Id"]
    name = r["Kernel_Name"]
    key = any(k in name for k in ("k_gru", "k_make_keys"))
    if q != prev or key:
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} us  queue {q}  {re.sub(r'.anonymous namespace.::', '', name)[:50]}")
    prev = q
