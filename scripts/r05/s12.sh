#!/bin/bash
# Round 5, session 12: in-step A/B of the two-co-resident-blocks tile (id 129: 128x160, four waves, two blocks per CU) for every
# launch the picker gives the eight-wave 256x160 tile, sustained (30 steps per run, interleaved)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Lpaddlemix_amd -lmi355x_sd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/step_bench.c $L -o /tmp/step_bench || exit 1
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/paddlemix_amd
{
  for r in 1 2; do for m in "" "160:129" "160:128"; do echo "== step: MI355X_SD_GEMM_TILE_MAP=$m (round $r)"; MI355X_SD_GEMM_TILE_MAP=$m timeout 100 /tmp/step_bench scripts/c/sdxl_unet_config.json 8 128 128 77 30 3; done; done
} > $O/r05_s12_step_tile129.txt 2>&1
grep -h "==\|ms_per_step" $O/r05_s12_step_tile129.txt | sed 's/"launches.*//' | cut -c1-200
