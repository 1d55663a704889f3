#!/bin/bash
# Round 6, session 5: four-wave tile, one memory instruction per MFMA gap
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Iscripts/c -Lpaddlemix_amd -lmi355x_sd_dbg -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/gemm_probe.c $L -o /tmp/gemm_probe || exit 1
export LD_LIBRARY_PATH=paddlemix_amd
R=$O/${1:-r06_s5}_w4_probe.txt; : > $R
MASK=0x55
for round in 1 2; do
echo "== round $round: picker" >> $R
timeout 300 /tmp/gemm_probe 1500 $MASK 2>&1 | grep -v "^#" >> $R
for s in 0 1; do
  echo "-- four-wave tile, schedule $s" >> $R
  MI355X_SD_W4_SCHED=$s MI355X_SD_GEMM_TILE_MAP="320:258,160:258,257:258" timeout 300 /tmp/gemm_probe 1500 $MASK 2>&1 | grep -v "^#" >> $R
done
done
for s in 10 14 15; do
  echo "-- four-wave tile, ablation $s (10 no DMA; 11 no reads; 12 no barrier/waits; 14 no epilogue; 15 MFMA only, no epilogue)" >> $R
  MI355X_SD_W4_SCHED=$s MI355X_SD_GEMM_TILE_MAP="320:258,160:258,257:258" timeout 300 /tmp/gemm_probe 1000 0x5 2>&1 | grep -v "^#" | grep -v "shapes of one step" >> $R
done
cat $R
timeout 900 python -m pytest tests/test_gpu_gemm_variants.py -m gpu -q -x -k "four-waves or picker" 2>&1 | tail -4
