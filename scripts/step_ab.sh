#!/bin/bash
# In-step A/B of a debug-build switch with the plain-C step bench (scripts/c/step_bench.c: no Python in the process, seconds per point):
# the protocol behind every "ms per step, interleaved rounds" figure of round 6 in profiles/ (r06_s7, s12, s24, s32, s34, s39 ...).
#   usage (on the GPU box): bash scripts/step_ab.sh <sdxl|sd15> "<VAR=value [VAR=value ...]>" [rounds=3] [out file]
#   e.g.  bash scripts/step_ab.sh sdxl "MI355X_SD_NO_W4=1"          # the four-wave tile off
#         bash scripts/step_ab.sh sd15 "MI355X_SD_NO_SMALL=1" 3 gpurun_out/small_tile_ab.txt
# Both arms run the debug-switch library (libmi355x_sd_dbg.so: the only build that reads MI355X_SD_* switches), alternating, so box and
# clock drift hit both. Boxes of the pool differ by up to 10 %: only differences inside one invocation mean anything.
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
CFG=$1; SW=$2; ROUNDS=${3:-3}; OUT=${4:-/dev/stdout}
case $CFG in
  sdxl) ARGS="scripts/c/sdxl_unet_config.json 8 128 128 77 60 5" ;;
  sd15) ARGS="scripts/c/sd15_unet_config.json 1 64 64 77 200 20" ;;
  *) echo "config: sdxl | sd15"; exit 2 ;;
esac
L="-I/opt/rocm/include -Iinclude -Iscripts/c -Lpaddlemix_amd -lmi355x_sd_dbg -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/step_bench.c $L -o /tmp/step_bench || exit 1
export LD_LIBRARY_PATH=paddlemix_amd
ms() { python3 -c "import json,sys; print('%.3f ms/step' % json.loads(sys.stdin.read())['ms_per_step'])"; }
for r in $(seq 1 $ROUNDS); do
  echo -n "round $r  default            " >> $OUT
  timeout 300 /tmp/step_bench $ARGS 2>&1 | tail -1 | ms >> $OUT
  echo -n "round $r  $SW  " >> $OUT
  env $SW timeout 300 /tmp/step_bench $ARGS 2>&1 | tail -1 | ms >> $OUT
done
