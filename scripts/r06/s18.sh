#!/bin/bash
# Round 6, session 18: the whole GPU suite (the driver's command) with durations, after the suite-time cuts
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=25 ) > $O/r06_s18_pytest_gpu.txt 2>&1
tail -45 $O/r06_s18_pytest_gpu.txt
