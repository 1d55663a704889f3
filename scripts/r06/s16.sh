#!/bin/bash
# Round 6, session 16: the whole GPU suite, the driver's command, with durations
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=40 ) > $O/r06_s16_pytest_gpu.txt 2>&1
tail -60 $O/r06_s16_pytest_gpu.txt
