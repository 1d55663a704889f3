#!/bin/bash
# round-2 GPU session E: GEMM loader waves + interleaved fragment reads (MI355X_SD_GEMM_LOADERS=5)
set -x
O=gpurun_out/r02e; mkdir -p $O
export TMPDIR=/tmp
MI355X_SD_GEMM_LOADERS=5 timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "linear or conv3x3 or pipelined or geglu" > $O/kernels_lw5.log 2>&1; echo "rc=$?" >> $O/kernels_lw5.log
for lw in 0 4 5; do
  for shp in 8192x1280x1280 8192x3840x1280 8192x1280x5120 32768x640x640 32768x1920x640; do
    MI355X_SD_GEMM_LOADERS=$lw GEMM_SHAPE=$shp timeout 60 python scripts/gemm_one.py >> $O/gemm_one_lw$lw.log 2>&1
  done
  MI355X_SD_GEMM_LOADERS=$lw CONV=8x32x32x1280x1280 timeout 60 python scripts/gemm_one.py >> $O/gemm_one_lw$lw.log 2>&1
done
for rep in a b; do for lw in 0 4 5; do
  MI355X_SD_GEMM_LOADERS=$lw timeout 150 python bench.py --no-cpu-baseline --no-roofline --steps 20 > $O/bench_lw${lw}$rep.json 2> /dev/null
done; done
BENCH_SHAPES=1 MI355X_SD_GEMM_LOADERS=5 timeout 150 python bench.py --no-cpu-baseline --steps 20 > $O/bench_lw5.json 2> $O/bench_lw5.err
tail -2 $O/kernels_lw5.log; grep -h "gemm\|conv" $O/gemm_one_lw0.log $O/gemm_one_lw4.log $O/gemm_one_lw5.log
