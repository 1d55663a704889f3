#!/bin/bash
# Round 6, session 3: timing ablations of the four-wave tile (debug build, wrong results by construction) + the two-streams test, diagnostic form
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Iscripts/c -Lpaddlemix_amd -lmi355x_sd_dbg -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/gemm_probe.c $L -o /tmp/gemm_probe || exit 1
export LD_LIBRARY_PATH=paddlemix_amd
R=$O/r06_s3_w4_ablation.txt; : > $R
MASK=0x5   # shapes 0 (FF1 GEGLU), 2 (fused QKV)
echo "== picker" >> $R
timeout 300 /tmp/gemm_probe 1000 $MASK 2>&1 | grep -v "^#" >> $R
for s in 0 10 11 12 16 17 13 14 15; do
  echo "-- four-wave tile, variant $s (0 shipped; 10 no DMA; 11 no reads; 12 no barrier/waits; 16 no DMA+reads; 17 no DMA+barrier; 13 MFMA only; 14 no epilogue; 15 MFMA only, no epilogue)" >> $R
  MI355X_SD_W4_SCHED=$s MI355X_SD_GEMM_TILE_MAP="320:258,160:258,257:258" timeout 300 /tmp/gemm_probe 1000 $MASK 2>&1 | grep -v "^#" | grep -v "shapes of one step" >> $R
done
cat $R
timeout 600 python -m pytest "tests/test_gpu_unet.py::test_two_models_on_two_streams_match_their_serial_runs_bit_for_bit" tests/test_gpu_seams.py -m gpu -q 2>&1 | tail -12 > $O/r06_s3_tests.txt
cat $O/r06_s3_tests.txt
