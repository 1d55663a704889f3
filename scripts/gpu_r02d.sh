#!/bin/bash
# round-2 GPU session D: GEMM loader waves (256x160 three-stage tile): correctness, isolated shapes, A/B inside the SDXL step
set -x
O=gpurun_out/r02d; mkdir -p $O
export TMPDIR=/tmp
MI355X_SD_GEMM_LOADERS=4 timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "linear or conv3x3 or pipelined or geglu" > $O/kernels_lw4.log 2>&1; echo "rc=$?" >> $O/kernels_lw4.log
for lw in 0 4; do
  for shp in 8192x1280x1280 8192x3840x1280 8192x1280x5120 32768x640x640 32768x1920x640; do
    MI355X_SD_GEMM_LOADERS=$lw GEMM_SHAPE=$shp timeout 60 python scripts/gemm_one.py >> $O/gemm_one_lw$lw.log 2>&1
  done
  MI355X_SD_GEMM_LOADERS=$lw CONV=8x32x32x1280x1280 timeout 60 python scripts/gemm_one.py >> $O/gemm_one_lw$lw.log 2>&1
done
BENCH_SHAPES=1 MI355X_SD_GEMM_LOADERS=0 timeout 150 python bench.py --no-cpu-baseline --steps 20 > $O/bench_lw0.json 2> $O/bench_lw0.err
BENCH_SHAPES=1 MI355X_SD_GEMM_LOADERS=4 timeout 150 python bench.py --no-cpu-baseline --steps 20 > $O/bench_lw4.json 2> $O/bench_lw4.err
MI355X_SD_GEMM_LOADERS=0 timeout 150 python bench.py --no-cpu-baseline --no-roofline --steps 20 > $O/bench_lw0b.json 2> /dev/null
MI355X_SD_GEMM_LOADERS=4 timeout 150 python bench.py --no-cpu-baseline --no-roofline --steps 20 > $O/bench_lw4b.json 2> /dev/null
tail -2 $O/kernels_lw4.log; grep -h "gemm\|conv" $O/gemm_one_lw0.log $O/gemm_one_lw4.log
