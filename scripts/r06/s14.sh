#!/bin/bash
# Round 6, session 14: SD3 per-shape tables with / without the four-wave tile (debug build)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
for sw in w4 now4; do
  if [ $sw = now4 ]; then export MI355X_SD_NO_W4=1; else unset MI355X_SD_NO_W4; fi
  MI355X_SD_LIB=dbg BENCH_SHAPES=1 timeout 300 python bench.py --workload sd3-1024-bs8 --no-cpu-baseline --no-parity-mode --steps 10 2> $O/r06_s14_sd3_per_shape_$sw.txt > /tmp/b.json
  echo "== $sw"; grep "TFLOP/s" $O/r06_s14_sd3_per_shape_$sw.txt | head -14
done
unset MI355X_SD_NO_W4
MI355X_SD_LIB=dbg BENCH_SHAPES=1 timeout 300 python bench.py --workload sd3-1024-bs8-w8a8 --no-cpu-baseline --no-parity-mode --steps 10 2> $O/r06_s14_sd3_w8a8_per_shape.txt > /tmp/b.json
echo "== w8a8"; grep "TFLOP/s" $O/r06_s14_sd3_w8a8_per_shape.txt | head -14
