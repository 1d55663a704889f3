#!/bin/bash
# round-3 GPU session 4: debugging the epilogue rework (wrong elements at 4100x700x640; NaN of the first early-residual form)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
: > $O/r03_s4_debug.txt
d() { env "$@" timeout 120 python scripts/debug_epi.py 2>&1 | grep -v amdgpu.ids >> $O/r03_s4_debug.txt; }
d X=0
d MI355X_SD_GEMM_NO_BIAS_ACC=1
d MI355X_SD_GEMM_NO_EPI_BATCH=1
d MI355X_SD_GEMM_NO_PRE=1
d MI355X_SD_GEMM_NO_BIAS_ACC=1 MI355X_SD_GEMM_NO_EPI_BATCH=1 MI355X_SD_GEMM_NO_PRE=1
d MI355X_SD_GEMM_LOADERS=0
cat $O/r03_s4_debug.txt
timeout 420 python -m pytest tests/test_gpu_gemm_variants.py -m gpu -q 2>&1 | tail -15 > $O/r03_s4_tests.txt
cat $O/r03_s4_tests.txt
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "linear or conv3x3 or pipelined or geglu or layernorm_folded or sdpa" 2>&1 | tail -8 >> $O/r03_s4_tests.txt
tail -8 $O/r03_s4_tests.txt
