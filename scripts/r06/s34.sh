#!/bin/bash
# Round 6, session 34: tile rasterisation of the four-wave tile inside the SDXL bs-8 step (debug build: MI355X_SD_W4_GM), plain-C step bench
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Iscripts/c -Lpaddlemix_amd -lmi355x_sd_dbg -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/step_bench.c $L -o /tmp/step_bench || exit 1
export LD_LIBRARY_PATH=paddlemix_amd
R=$O/r06_s34_w4_gm.txt; : > $R
for round in 1 2; do
  for gm in -4 -2 -8 -16 4 8 2 -5; do
    echo -n "round $round  MI355X_SD_W4_GM=$gm  " >> $R
    MI355X_SD_W4_GM=$gm timeout 200 /tmp/step_bench scripts/c/sdxl_unet_config.json 8 128 128 77 40 5 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.3f ms/step' % d['ms_per_step'])" >> $R
  done
done
cat $R
