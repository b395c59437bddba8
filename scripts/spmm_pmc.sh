#!/bin/bash
# PMC passes over scripts/bench_spmm.py (hetero-SpMM variants, C2 layer shape); summaries by scripts/pmc_summary.py.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $line -d $R/gpurun_out/spmm_pmc$i -o p --output-format csv -- python3 $R/scripts/bench_spmm.py > $R/gpurun_out/spmm_pmc$i.log 2>&1 || echo "pass $i failed"
done <<'PASSES'
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum
TCC_BUSY_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_sum
FETCH_SIZE
WRITE_SIZE
PASSES
cd $R
for j in $(seq 1 $i); do python3 scripts/pmc_summary.py k_spmm_fast "gpurun_out/spmm_pmc$j/*counter_collection.csv"; done > gpurun_out/spmm_pmc_summary.txt 2>&1 || true
