#!/bin/bash
# Round 6, session 17: the suite's three longest items after the cuts (oracle loop once; fp16 child without the SDXL draw), with the
# parity children's per-case wall times
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 1200 python -m pytest tests/test_gpu_parity_loops.py tests/test_gpu_fp16.py tests/test_gpu_unet.py::test_euler30_latents_vs_float64_oracle_loop -x -q -s -m gpu --durations=10 ) > $O/r06_s17_slow3.txt 2>&1
grep -n "PARITY_TIMING\|passed\|failed\|real\|s call\|s setup" $O/r06_s17_slow3.txt | cut -c1-1500
nproc; free -g | head -2
