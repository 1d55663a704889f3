#!/bin/bash
# Round 6, session 41: rocprofv3 counters of the four-wave tile (gemm_w4_kernel), FF1 GEGLU 8192x10240x1280 and fused QKV 8192x3840x1280
# separately, plain-C probe (4 weight copies in rotation), one --pmc pass per list, no tracing other than the kernel trace
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
: > $O/r06_s41_w4_counters.txt
for m in 0x1 0x4; do
  echo "== gemm_probe mask $m (0x1: FF1 GEGLU 8192x10240x1280, 5 tiles per block; 0x4: fused QKV 8192x3840x1280, 480 tiles on 256 blocks)" >> $O/r06_s41_w4_counters.txt
  bash scripts/pmc_c.sh gemm_probe "12 $m 4" gemm_w4 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
    "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE" >> $O/r06_s41_w4_counters.txt 2>&1
done
cut -c1-20,78-170 $O/r06_s41_w4_counters.txt
