#!/bin/bash
# Round 4, session 7: the whole -m gpu suite WITHOUT -x (session 6 stopped at its first failure: the CPU-exported VAE program on the
# device), then that test on the generic GEMM loop (is it the pipelined loops?) and with a debug dump of per-launch differences.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
T=r04_s7
( time timeout 2700 python -X faulthandler -m pytest tests -m gpu -q 2>&1 | grep -v "amdgpu.ids" > /tmp/suite.log ) 2> /tmp/suite.time
( grep -E "^(FAILED|ERROR)|passed|failed" /tmp/suite.log | cut -c1-300; echo ...; tail -80 /tmp/suite.log | cut -c1-300; cat /tmp/suite.time ) > $O/${T}_pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed" $O/${T}_pytest_gpu.log | head -20
( MI355X_SD_NO_PIPE=1 timeout 300 python -m pytest tests/test_gpu_export.py -m gpu -q -k "without_a_gpu" 2>&1 | tail -5 | cut -c1-300 ) > $O/${T}_export_nopipe.txt; cat $O/${T}_export_nopipe.txt
timeout 300 python scripts/debug_cpu_exported_vae.py > $O/${T}_debug_vae.txt 2>&1; tail -40 $O/${T}_debug_vae.txt | cut -c1-200
