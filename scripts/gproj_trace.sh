#!/bin/bash
# kernel-trace average / min duration of the grouped-projection kernels in scripts/bench_gproj.py (env passes through)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/gp_tr
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/gp_tr -o g --output-format csv -- python3 $R/scripts/bench_gproj.py > /dev/null 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/gp_tr/g_kernel_stats.csv")):
    if "gproj" in r["Name"]:
        print("  ", r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:22], r["Calls"], "avg %.1f us  min %.1f us" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
rm -rf $R/gpurun_out/gp_tr
