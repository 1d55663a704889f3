#!/bin/bash
# Round 6, session 37: variant tests with the ragged WS case
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
( timeout 1200 python -m pytest tests/test_gpu_gemm_variants.py -x -q -m gpu ) > $O/r06_s37_tests.txt 2>&1; tail -4 $O/r06_s37_tests.txt
