#!/bin/bash
# Round 5, session 19: counters of the two self-attention kernels (the shipped 16x16x32 one; the 32x32x16 one it replaced, debug
# build with MI355X_SD_ATTN_NO_M16=1) on the step's attention launches: matrix-pipe busy cycles, VALU / LDS instruction counts
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
C1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
C2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU"
C3="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES"
C4="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_ANY"
{
  echo "== shipped library: attention16_kernel (16x16x32 MFMA)"
  bash scripts/pmc_c.sh attn_probe 6 attention "$C1" "$C2" "$C3" "$C4"
  echo "== debug-switch library, MI355X_SD_ATTN_NO_M16=1: attention_kernel (32x32x16 MFMA)"
  PMC_LIB=mi355x_sd_dbg MI355X_SD_ATTN_NO_M16=1 bash scripts/pmc_c.sh attn_probe 6 attention "$C1" "$C2" "$C3" "$C4"
} > $O/r05_s19_attn_counters.txt 2>&1
cat $O/r05_s19_attn_counters.txt | cut -c1-200 | head -90
