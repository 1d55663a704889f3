#!/bin/bash
# Round 6, session 24: the four-wave tile at 256 x 160 (N = 640 / 1280 launches): variant tests, step A/B (plain-C step bench, debug
# build: MI355X_SD_NO_W4_160), per-shape tables
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_gemm_variants.py -x -q -m gpu ) > $O/r06_s24_tests.txt 2>&1; tail -5 $O/r06_s24_tests.txt
L="-I/opt/rocm/include -Iinclude -Iscripts/c -Lpaddlemix_amd -lmi355x_sd_dbg -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/step_bench.c $L -o /tmp/step_bench || exit 1
export LD_LIBRARY_PATH=paddlemix_amd
R=$O/r06_s24_step_ab.txt; : > $R
for round in 1 2 3; do
  echo "round $round  picker with the four-wave 256 x 160 tile:" >> $R
  timeout 200 /tmp/step_bench scripts/c/sdxl_unet_config.json 8 128 128 77 60 5 2>&1 | tail -1 | cut -c1-220 >> $R
  echo "round $round  MI355X_SD_NO_W4_160=1:" >> $R
  MI355X_SD_NO_W4_160=1 timeout 200 /tmp/step_bench scripts/c/sdxl_unet_config.json 8 128 128 77 60 5 2>&1 | tail -1 | cut -c1-220 >> $R
done
cat $R
for sw in w4n160 eightwave; do
  if [ $sw = eightwave ]; then export MI355X_SD_NO_W4_160=1; else unset MI355X_SD_NO_W4_160; fi
  MI355X_SD_LIB=dbg BENCH_SHAPES=1 timeout 300 python bench.py --no-cpu-baseline --no-parity-mode --steps 10 2> $O/r06_s24_per_shape_$sw.txt > /dev/null
  echo "== $sw"; grep "TFLOP/s" $O/r06_s24_per_shape_$sw.txt | head -14
done
