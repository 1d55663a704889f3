#!/bin/bash
# Round 6, session 30: ragged d = 64 attention cases (plain and base-2), forward() restoring the staged host's in_scale
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
( timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py -x -q -m gpu -k "sdpa or scale or forward or denoise or staged" ) > $O/r06_s30_tests.txt 2>&1; tail -5 $O/r06_s30_tests.txt
