import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
side_pat = ("sin", "tanh")
side_idx = [i for i, r in enumerate(rows) if any(p in r["Kernel_Name"] for p in side_pat)]
last = side_idx[-12:]                       # the final work() call
t_side0 = int(rows[last[0]]["Start_Timestamp"])
big = [r for r in rows if "Cijk" in r["Kernel_Name"] and (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) > 150000]
big = big[-24:]
g0 = int(big[0]["Start_Timestamp"])
done = sum(1 for r in big if int(r["End_Timestamp"]) <= t_side0)
print(f"big GEMMs {len(big)}, avg {(sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in big)/len(big))/1e3:.1f} us; side branch started "
      f"{(t_side0 - g0)/1e3:.1f} us after the chain began, {done} big GEMMs complete (fork is after #4); side span "
      f"{(int(rows[last[-1]]['End_Timestamp']) - t_side0)/1e3:.1f} us")
