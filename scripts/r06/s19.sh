#!/bin/bash
# Round 6, session 19: the C++ planner's class embeddings / timestep_cond against the Python-planned model; smoke with the fp16 child
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_cexec.py tests/test_gpu_export.py -x -q -m gpu --durations=5 ) > $O/r06_s19_cexec_tests.txt 2>&1
tail -15 $O/r06_s19_cexec_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_s19_smoke.txt 2>&1; tail -4 $O/r06_s19_smoke.txt
