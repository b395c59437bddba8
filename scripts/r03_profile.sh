#!/bin/bash
# Round-3 profiles: kernel-trace stats of the default bench run (c2s) and of the device-sampled HGT step (c3d), the trimmed
# aggregation launches alone, then FETCH_SIZE / WRITE_SIZE in SEPARATE --pmc passes (gfx950: FETCH_SIZE x2 correction applied later).
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3_prof_bench -o b --output-format csv -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other > $R/gpurun_out/r3_prof_bench.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3_prof_c3d -o b --output-format csv -- python3 $R/bench.py --workload c3d --steps 30 --warmup 5 --no-cpu-baseline --no-other > $R/gpurun_out/r3_prof_c3d.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3_prof_spmm -o s --output-format csv -- python3 $R/scripts/spmm_trim_case.py 256 > $R/gpurun_out/r3_prof_spmm.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r3_pmc_fetch -o p --output-format csv -- python3 $R/scripts/spmm_trim_case.py 256 > $R/gpurun_out/r3_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/r3_pmc_write -o p --output-format csv -- python3 $R/scripts/spmm_trim_case.py 256 > $R/gpurun_out/r3_pmc_write.log 2>&1
cd $R
for d in r3_pmc_fetch r3_pmc_write; do python3 scripts/pmc_summary.py k_spmm "gpurun_out/$d/**/*counter_collection.csv"; done > gpurun_out/r3_spmm_pmc_summary.txt 2>&1 || true
python3 scripts/step_timeline.py gpurun_out/r3_prof_bench/b_kernel_trace.csv > gpurun_out/r3_step_timeline_c2s.txt 2>&1 || true
# what travels back is capped at 64 MiB: keep the summaries, drop the raw traces and counter dumps
find gpurun_out/r3_prof_bench gpurun_out/r3_prof_c3d gpurun_out/r3_prof_spmm gpurun_out/r3_pmc_fetch gpurun_out/r3_pmc_write \( -name "*kernel_trace.csv" -o -name "*counter_collection.csv" \) -delete
find gpurun_out/r3_prof_bench gpurun_out/r3_prof_c3d gpurun_out/r3_prof_spmm -name "*kernel_stats.csv" | head
tail -2 gpurun_out/r3_prof_bench.log | cut -c1-300
cat gpurun_out/r3_spmm_pmc_summary.txt | tail -12
