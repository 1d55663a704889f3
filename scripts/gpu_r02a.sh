#!/bin/bash
# round-2 GPU session A: new parity tests, parity report, bench variants, measured CPU baseline
set -x
O=gpurun_out/r02a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py -q -x -s -k "ragged_kv_tail or fp32_residual or headline or euler30 or from_pretrained" > $O/tests_new.log 2>&1; echo "tests rc=$?" >> $O/tests_new.log
timeout 1200 python scripts/parity_report.py --out $O/parity.json > $O/parity.log 2>&1; echo "parity rc=$?" >> $O/parity.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --residual fp32 --no-cpu-baseline > $O/bench_resid32.json 2> $O/bench_resid32.err
timeout 300 python bench.py --dtype fp16 --residual fp32 --no-cpu-baseline > $O/bench_fp16_resid32.json 2> $O/bench_fp16_resid32.err
timeout 300 python bench.py --text-encoders --no-cpu-baseline --no-roofline --steps 10 > $O/bench_te.json 2> $O/bench_te.err
timeout 600 python scripts/cpu_baseline.py --out $O/cpu_baseline_sdxl-1024-bs8.json > $O/cpu_baseline.log 2>&1
tail -3 $O/tests_new.log; tail -2 $O/parity.log; cat $O/bench_default.json | head -c 1500
