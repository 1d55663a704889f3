#!/bin/bash
# Round 6, session 32: split-K policy of the batch-1 step (debug build: MI355X_SD_SPLITK_POLICY=max_tiles:target_blocks), plain-C step bench
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Iscripts/c -Lpaddlemix_amd -lmi355x_sd_dbg -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/step_bench.c $L -o /tmp/step_bench || exit 1
export LD_LIBRARY_PATH=paddlemix_amd
R=$O/r06_s32d_splitk_policy.txt; : > $R
for round in 1 2; do
  for pol in 160:416:4 160:416:2 160:416:3 160:416:6 160:416:8 160:416:12 256:416:6 160:320:6; do
    echo -n "round $round  policy $pol  " >> $R
    MI355X_SD_SPLITK_POLICY=$pol timeout 100 /tmp/step_bench scripts/c/sd15_unet_config.json 1 64 64 77 200 20 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.4f ms/step  %d launches' % (d['ms_per_step'], d['launches']))" >> $R
  done
done
cat $R
