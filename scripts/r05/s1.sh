#!/bin/bash
# Round 5, session 1: pipelined attention (accuracy vs float64, A/B vs the non-pipelined kernel), staggered start of persistent GEMMs.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Lpaddlemix_amd -lmi355x_sd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
for p in attn_check attn_probe gemm_probe step_bench; do gcc -std=c11 -O2 scripts/c/$p.c $L -o /tmp/$p || exit 1; done
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/paddlemix_amd
{
  for il in 0 1 2 3 4; do echo "== accuracy, MI355X_SD_ATTN_IL=$il"; MI355X_SD_ATTN_IL=$il timeout 120 /tmp/attn_check; done
} > $O/r05_s1_attn_check.txt 2>&1
{
  for r in 1 2; do for il in 0 1 2 3 4; do echo "== MI355X_SD_ATTN_IL=$il (round $r)"; MI355X_SD_ATTN_IL=$il timeout 100 /tmp/attn_probe 20; done; done
} > $O/r05_s1_attn_probe.txt 2>&1
{
  for st in 0 "125,4,3" "250,2,3" "250,4,3" "60,8,3" "125,4,2" 0; do echo "== MI355X_SD_GEMM_STAGGER=$st"; MI355X_SD_GEMM_STAGGER=$st timeout 100 /tmp/gemm_probe 20; done
} > $O/r05_s1_gemm_stagger.txt 2>&1
{
  for r in 1 2; do
    for v in "0 0" "1 0" "1 125,4,3" "2 125,4,3"; do set -- $v
      echo "== step: MI355X_SD_ATTN_IL=$1 MI355X_SD_GEMM_STAGGER=$2 (round $r)"
      MI355X_SD_ATTN_IL=$1 MI355X_SD_GEMM_STAGGER=$2 timeout 100 /tmp/step_bench scripts/c/sdxl_unet_config.json 8 128 128 77 30 3
    done
  done
} > $O/r05_s1_step.txt 2>&1
grep -h "FAIL\|all cases\|FAILED\|==" $O/r05_s1_attn_check.txt | head -40
grep -h "==\|self\|launches of one" $O/r05_s1_attn_probe.txt | cut -c1-150
grep -h "==\|shapes of one step\|x 1280x1280\|3840x1280\|10240x1280\|1920x 640\|5120x 640" $O/r05_s1_gemm_stagger.txt | cut -c1-150
grep -h "==\|ms_per_step\|steps_per" $O/r05_s1_step.txt | cut -c1-250
