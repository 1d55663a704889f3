#!/bin/bash
# Round 4, session 19 (the round's last GPU seconds): the deferred-store variant (scripts/experiments/r05_deferred_stores.patch,
# prebuilt into build_exp/defer) through the prebuilt linear-shape probe: do the hashes equal the shipped library's?
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/build_exp/defer timeout 9 build_exp/gemm_probe 5 > $O/r04_s19_defer_probe.txt 2>&1
cat $O/r04_s19_defer_probe.txt
