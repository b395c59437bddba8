#!/bin/bash
# Round-2 C3 (HGT) profiles: kernel-trace stats of the c3 bench, FETCH_SIZE / WRITE_SIZE passes for k_hgt_fwd.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2_prof_c3f -o c --output-format csv -- python3 $R/bench.py --workload c3 --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r2_prof_c3f.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r2_pmc_c3_fetch -o p --output-format csv -- python3 $R/bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r2_pmc_c3_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/r2_pmc_c3_write -o p --output-format csv -- python3 $R/bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r2_pmc_c3_write.log 2>&1
cd $R
for d in r2_pmc_c3_fetch r2_pmc_c3_write; do python3 scripts/pmc_summary.py k_hgt "gpurun_out/$d/*counter_collection.csv"; done
tail -1 gpurun_out/r2_prof_c3f.log
