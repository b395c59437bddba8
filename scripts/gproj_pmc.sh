#!/bin/bash
# PMC passes over scripts/bench_gproj.py (the grouped head projection alone).  usage: gproj_pmc.sh <tag> (env AGNN_GPROJ_FWD passes through)
set -e
R=$GRAFT_REPO_ROOT
TAG=$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 GRBM_GUI_ACTIVE SQ_WAVES -d $R/gpurun_out/$TAG/p1 -o p --output-format csv -- python3 $R/scripts/bench_gproj.py > $R/gpurun_out/$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum -d $R/gpurun_out/$TAG/p2 -o p --output-format csv -- python3 $R/scripts/bench_gproj.py >> $R/gpurun_out/$TAG.log 2>&1 || true
cd $R
for d in p1 p2; do python3 scripts/pmc_summary.py k_gproj_fwd "gpurun_out/$TAG/$d/*counter_collection.csv"; done > gpurun_out/$TAG/summary.txt 2>&1 || true
rm -rf gpurun_out/$TAG/p1 gpurun_out/$TAG/p2
cat gpurun_out/$TAG/summary.txt
