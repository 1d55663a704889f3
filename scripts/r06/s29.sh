#!/bin/bash
# Round 6, session 29: after removing the e4m3 four-wave kernel -- variant tests, SD3 tests
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
( timeout 1200 python -m pytest tests/test_gpu_gemm_variants.py tests/test_gpu_sd3.py tests/test_gpu_kernels.py -x -q -m gpu ) > $O/r06_s29_tests.txt 2>&1; tail -5 $O/r06_s29_tests.txt
