#!/bin/bash
# Round 6, session 20: the C++ planner on odd latent sizes (forward_upsample_size) against the Python-planned model; plain-C client
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_cexec.py -x -q -m gpu --durations=5 ) > $O/r06_s20_cexec_tests.txt 2>&1
tail -12 $O/r06_s20_cexec_tests.txt
