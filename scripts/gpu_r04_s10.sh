#!/bin/bash
# Round 4, session 10: L2 counters of the two largest GEMM launches of the final code (hit rate, fabric read / write requests).
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
T=r04_s10
P="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
( echo "== gemm 8192x3840x1280 (fused QKV; 256x160 interleaved loop, column groups of 8)"; GEMM_SHAPE=8192x3840x1280 timeout 200 bash scripts/pmc.sh scripts/gemm_one.py gemm_pipe_kernel "$P"
  echo "== gemm 8192x10240x1280 GEGLU (FF1; 256x320 streaming loop, column groups of 4)"; GEMM_SHAPE=8192x10240x1280 GEGLU=1 timeout 200 bash scripts/pmc.sh scripts/gemm_one.py gemm_pipe_kernel "$P" ) > $O/${T}_l2_counters.txt 2>&1
cat $O/${T}_l2_counters.txt
