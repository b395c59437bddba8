#!/bin/bash
# Round-2 profiles: kernel-trace stats of the default bench run (sampled batch) and of the trimmed aggregation launches,
# then FETCH_SIZE / WRITE_SIZE in separate --pmc passes (gfx950: TCC has 4 slots; FETCH_SIZE x2 correction applied later).
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2_prof_bench -o b --output-format csv -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other > $R/gpurun_out/r2_prof_bench.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2_prof_spmm -o s --output-format csv -- python3 $R/scripts/spmm_trim_case.py 256 > $R/gpurun_out/r2_prof_spmm.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r2_pmc_fetch -o p --output-format csv -- python3 $R/scripts/spmm_trim_case.py 256 > $R/gpurun_out/r2_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/r2_pmc_write -o p --output-format csv -- python3 $R/scripts/spmm_trim_case.py 256 > $R/gpurun_out/r2_pmc_write.log 2>&1
cd $R
for d in r2_pmc_fetch r2_pmc_write; do python3 scripts/pmc_summary.py k_spmm "gpurun_out/$d/*/*counter_collection.csv"; done > gpurun_out/r2_spmm_pmc_summary.txt 2>&1 || true
find gpurun_out/r2_prof_bench gpurun_out/r2_prof_spmm -name "*kernel_stats.csv" | head
