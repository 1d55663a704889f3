#!/bin/bash
# Round 6, session 2: the four-wave 256x256 GEMM tile (csrc/gemm_w4.hip) -- bit-identity, then sustained A/B against the picker's
# tiles on the step's wide shapes (plain-C probe on the debug-switch library, no torch in the process).
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Iscripts/c -Lpaddlemix_amd -lmi355x_sd_dbg -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/gemm_probe.c $L -o /tmp/gemm_probe || exit 1
export LD_LIBRARY_PATH=paddlemix_amd
R=$O/r06_s2_w4_probe.txt; : > $R
MASK=0x55   # shapes 0 (FF1 GEGLU), 2 (fused QKV), 4 (FF1 at 640 GEGLU), 6 (QKV at 640)
echo "== hashes: bias added in the epilogue on both sides (MI355X_SD_GEMM_NO_BIAS_ACC=1), 3 launches" >> $R
MI355X_SD_GEMM_NO_BIAS_ACC=1 timeout 120 /tmp/gemm_probe 3 $MASK 2>&1 | grep -v "^#" >> $R
echo "-- four-wave tile" >> $R
MI355X_SD_GEMM_NO_BIAS_ACC=1 MI355X_SD_GEMM_TILE_MAP="320:258,160:258,257:258" timeout 120 /tmp/gemm_probe 3 $MASK 2>&1 | grep -v "^#" >> $R
for round in 1 2; do
  echo "== sustained, 2000 launches per shape, round $round: picker" >> $R
  timeout 300 /tmp/gemm_probe 2000 $MASK 2>&1 | grep -v "^#" >> $R
  for s in 0 1 2 3; do
    echo "-- four-wave tile, schedule $s (0: barrier step 18, a piece every 2 steps; 1: 20/1; 2: 16/2; 3: 24/2)" >> $R
    MI355X_SD_W4_SCHED=$s MI355X_SD_GEMM_TILE_MAP="320:258,160:258,257:258" timeout 300 /tmp/gemm_probe 2000 $MASK 2>&1 | grep -v "^#" >> $R
  done
done
cat $R
timeout 900 python -m pytest tests/test_gpu_gemm_variants.py -m gpu -q -x 2>&1 | tail -6 > $O/r06_s2_variants_tests.txt
cat $O/r06_s2_variants_tests.txt
timeout 600 python -m pytest "tests/test_gpu_unet.py::test_two_models_on_two_streams_match_their_serial_runs_bit_for_bit" tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -6 > $O/r06_s2_abi12_tests.txt
cat $O/r06_s2_abi12_tests.txt
