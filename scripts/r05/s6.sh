#!/bin/bash
# Round 5, session 6: the shader clock while the attention kernel runs (s_memtime vs the 100-MHz wall clock, stamped by the kernel)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Lpaddlemix_amd -lmi355x_sd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/attn_probe.c $L -o /tmp/attn_probe || exit 1
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/paddlemix_amd
{
  for il in 0 1 0 1; do echo "== MI355X_SD_ATTN_ABL=64 MI355X_SD_ATTN_IL=$il"; MI355X_SD_ATTN_IL=$il MI355X_SD_ATTN_ABL=64 timeout 100 /tmp/attn_probe 20 | grep "self\|clock"; done
} > $O/r05_s6_attn_clock.txt 2>&1
cut -c1-200 $O/r05_s6_attn_clock.txt
