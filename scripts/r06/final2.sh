#!/bin/bash
# Round 6: final measurement set + the whole GPU suite (the driver's command) on the final tree
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash scripts/r06/final.sh
( time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=15 ) > gpurun_out/r06_f_pytest_gpu.txt 2>&1
tail -24 gpurun_out/r06_f_pytest_gpu.txt
