#!/bin/bash
# round-2 GPU session C: attention variants (lazy maximum, base-2 scores, eight-wave kernel), their effect on the SDXL step
set -x
O=gpurun_out/r02c; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python scripts/attn8_probe.py --iters 10 > $O/attn.log 2>&1; echo "rc=$?" >> $O/attn.log
timeout 240 python -m pytest tests/test_gpu_kernels.py -q -x -k "sdpa" > $O/sdpa.log 2>&1; echo "rc=$?" >> $O/sdpa.log
MI355X_SD_FOLD_SCALE=1 timeout 300 python -m pytest tests/test_gpu_unet.py -q -x -s -k "small_unet or fp32_residual" > $O/unet_fold.log 2>&1; echo "rc=$?" >> $O/unet_fold.log
MI355X_SD_ATTN_LAZY=0 timeout 150 python bench.py --no-cpu-baseline --steps 20 > $O/bench_exact.json 2> $O/bench_exact.err
timeout 150 python bench.py --no-cpu-baseline --steps 20 > $O/bench_lazy.json 2> $O/bench_lazy.err
MI355X_SD_FOLD_SCALE=1 timeout 150 python bench.py --no-cpu-baseline --steps 20 > $O/bench_fold.json 2> $O/bench_fold.err
MI355X_SD_FOLD_SCALE=1 MI355X_SD_ATTN8=9 timeout 150 python bench.py --no-cpu-baseline --steps 20 > $O/bench_fold_a8.json 2> $O/bench_fold_a8.err
grep -v ATTN8_JSON $O/attn.log | tail -40
