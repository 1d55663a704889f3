#!/bin/bash
# Round 4, session 15: the IEEE-half build of the library (the element type of the bench line's parity_mode) through the same plain-C
# step bench, next to the bfloat16 build, one box.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Lpaddlemix_amd -lmi355x_sd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/step_bench.c $L -o /tmp/step_bench || exit 1
{
  for v in base f16 base f16; do
    echo "== $v"
    LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/build_exp/$v timeout 100 /tmp/step_bench scripts/c/sdxl_unet_config.json 8 128 128 77 30 3
  done
} > $O/r04_s15_c_step_bf16_fp16.txt 2>&1
cut -c1-200 $O/r04_s15_c_step_bf16_fp16.txt
