#!/bin/bash
# Round 6, session 39: non-temporal C stores in the four-wave tile's epilogue (debug build: MI355X_SD_W4_NT_STORE), SDXL step + SD3 step
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Iscripts/c -Lpaddlemix_amd -lmi355x_sd_dbg -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/step_bench.c $L -o /tmp/step_bench || exit 1
gcc -std=c11 -O2 scripts/c/gemm_probe.c $L -o /tmp/gemm_probe || exit 1
export LD_LIBRARY_PATH=paddlemix_amd
R=$O/r06_s39_nt_store.txt; : > $R
for round in 1 2 3; do
  echo -n "round $round  default stores       " >> $R
  timeout 200 /tmp/step_bench scripts/c/sdxl_unet_config.json 8 128 128 77 60 5 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step' % d['ms_per_step'])" >> $R
  echo -n "round $round  non-temporal stores  " >> $R
  MI355X_SD_W4_NT_STORE=1 timeout 200 /tmp/step_bench scripts/c/sdxl_unet_config.json 8 128 128 77 60 5 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step' % d['ms_per_step'])" >> $R
done
for arm in default nt; do
  if [ $arm = nt ]; then export MI355X_SD_W4_NT_STORE=1; else unset MI355X_SD_W4_NT_STORE; fi
  echo "== isolated, 128 weight copies, $arm" >> $R
  timeout 300 /tmp/gemm_probe 1024 0x55 128 2>&1 | grep -v "^#" >> $R
done
unset MI355X_SD_W4_NT_STORE
cat $R
