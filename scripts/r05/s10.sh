#!/bin/bash
# Round 5, session 10: the attention loops again, SUSTAINED (3000 launches per shape: a 20-launch probe right after idle sits in the
# clock ramp -- 420 us where 6000 launches average 364 us, profiles/r05_s9_attn_long.txt)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Lpaddlemix_amd -lmi355x_sd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/attn_probe.c $L -o /tmp/attn_probe || exit 1
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/paddlemix_amd
{
  for r in 1 2; do for il in 0 1 5 8 3; do echo "== MI355X_SD_ATTN_IL=$il (round $r)"; MI355X_SD_ATTN_IL=$il timeout 100 /tmp/attn_probe 3000 | grep "self"; done; done
} > $O/r05_s10_attn_sustained.txt 2>&1
cut -c1-120 $O/r05_s10_attn_sustained.txt
