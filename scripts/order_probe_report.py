import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sin = [i for i, r in enumerate(rows) if "sin" in r["Kernel_Name"]]
last3 = sin[-3:]
t_side = int(rows[last3[0]]["Start_Timestamp"])
mul = [r for r in rows if "MulFunctor" in r["Kernel_Name"] or "mul" in r["Kernel_Name"].lower()]
# kernels of the last replay: those within 5 ms before the last sin end
t_end = int(rows[last3[-1]]["End_Timestamp"])
win = [r for r in mul if t_end - 5_000_000 < int(r["Start_Timestamp"]) <= t_end]
# the last replay = the final contiguous run: take the last (N+5) mul kernels
n = int(sys.argv[2])
win = win[-(n + 5):]
t_fork = int(win[4]["End_Timestamp"])
done_before = sum(1 for r in win[5:] if int(r["End_Timestamp"]) <= t_side)
print(f"N={n}: side branch starts {(t_side - t_fork)/1e3:.0f} us after the fork; {done_before} of {n} main-branch kernels had finished; main branch ends {(int(win[-1]['End_Timestamp']) - t_fork)/1e3:.0f} us after the fork")
