#!/bin/bash
# Round 6, session 38: what cold weights cost a launch -- the step's linear shapes with ONE resident weight matrix (what every isolated
# probe so far measured) against a rotation of copies larger than the Infinity Cache (what the step's layers see)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Iscripts/c -Lpaddlemix_amd -lmi355x_sd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/gemm_probe.c $L -o /tmp/gemm_probe || exit 1
export LD_LIBRARY_PATH=paddlemix_amd
R=$O/r06_s38_cold_weights.txt; : > $R
for round in 1 2; do
  for wrot in 1 16 128; do
    echo "== round $round: $wrot weight copies in rotation" >> $R
    timeout 300 /tmp/gemm_probe 1024 0xFF $wrot 2>&1 | grep -v "^#" >> $R
  done
done
cat $R
