#!/bin/bash
# Round 5, first GPU call: the extended issue-rate table, the memory-path rates, and the issue counters of the shipped rasteriser.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout -k 10 300 ./tools/microbench/issue_rates > gpurun_out/r05/issue_rates.txt 2>&1; echo "issue_rates rc $?"
timeout -k 10 200 ./tools/microbench/gather_rates > gpurun_out/r05/gather_rates.txt 2>&1; echo "gather_rates rc $?"
export KBENCH_SIZES=1280x800x250
timeout -k 10 500 tools/pmc_kbench.sh issue \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
  "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" \
  "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT" \
  "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU SQ_INSTS_SALU" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS" \
  "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_SALU" \
  "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_IFETCH" \
  > gpurun_out/r05/pmc_issue_summary.txt 2>&1
grep raster_tiles gpurun_out/r05/pmc_issue_summary.txt
