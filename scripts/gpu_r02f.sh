#!/bin/bash
# round-2 GPU session F: the whole -m gpu suite on the final code
set -x
O=gpurun_out/r02f; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -s --durations=15 > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log
tail -30 $O/pytest_gpu.log; tail -3 $O/smoke.log
