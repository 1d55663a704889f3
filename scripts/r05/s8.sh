#!/bin/bash
# Round 5, session 8: two query tiles per wave, one wave per SIMD (attention_q2_kernel) vs the other loops
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Lpaddlemix_amd -lmi355x_sd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
for p in attn_check attn_probe; do gcc -std=c11 -O2 scripts/c/$p.c $L -o /tmp/$p || exit 1; done
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/paddlemix_amd
{
  for il in 8; do echo "== accuracy, MI355X_SD_ATTN_IL=$il"; MI355X_SD_ATTN_IL=$il timeout 120 /tmp/attn_check; done
} > $O/r05_s8_attn_check.txt 2>&1
{
  for r in 1 2; do for il in 0 8 0 8; do echo "== MI355X_SD_ATTN_IL=$il (round $r)"; MI355X_SD_ATTN_IL=$il timeout 100 /tmp/attn_probe 20 | grep "self"; done; done
} > $O/r05_s8_attn_probe.txt 2>&1
grep -h "FAIL\|all cases\|FAILED\|==" $O/r05_s8_attn_check.txt | head -60
cut -c1-120 $O/r05_s8_attn_probe.txt
