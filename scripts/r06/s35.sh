#!/bin/bash
# Round 6, session 35: which launches the 64 x 64 ring tile takes (debug build: MI355X_SD_SMALL_POLICY=nt_max:tiles_min:nt_max2), SD-1.5 bs 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Iscripts/c -Lpaddlemix_amd -lmi355x_sd_dbg -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/step_bench.c $L -o /tmp/step_bench || exit 1
export LD_LIBRARY_PATH=paddlemix_amd
R=$O/r06_s35_small_policy.txt; : > $R
for round in 1 2; do
  for pol in 48:160:96 32:160:96 24:160:96 64:160:96 96:160:96 48:80:96 48:320:96 48:160:48 48:64:128 20:160:40 40:160:80; do
    echo -n "round $round  policy $pol  " >> $R
    MI355X_SD_SMALL_POLICY=$pol timeout 100 /tmp/step_bench scripts/c/sd15_unet_config.json 1 64 64 77 200 20 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.4f ms/step' % d['ms_per_step'])" >> $R
  done
done
cat $R
