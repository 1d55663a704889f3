#!/bin/bash
# Round 5, session 11: the SDXL step (plain C, hipGraph replay, 30 steps) with the 32x32x16 / the 16x16x32 / the pipelined attention loop
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Lpaddlemix_amd -lmi355x_sd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/step_bench.c $L -o /tmp/step_bench || exit 1
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/paddlemix_amd
{
  for r in 1 2 3; do for il in 0 8 1; do echo "== step: MI355X_SD_ATTN_IL=$il (round $r)"; MI355X_SD_ATTN_IL=$il timeout 100 /tmp/step_bench scripts/c/sdxl_unet_config.json 8 128 128 77 30 3; done; done
} > $O/r05_s11_step_attn.txt 2>&1
grep -h "==\|ms_per_step" $O/r05_s11_step_attn.txt | cut -c1-200
