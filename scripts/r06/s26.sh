#!/bin/bash
# Round 6, session 26: the four-wave e4m3 tile (gemm_w4f8.hip): bit-identity vs the phased kernel, W8A8 tests, SD3 W8A8 A/B + per-shape
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_gemm_variants.py tests/test_gpu_sd3.py -x -q -m gpu -k "e4m3 or w8a8 or fp8 or operand_variants" ) > $O/r06_s26_tests.txt 2>&1; tail -8 $O/r06_s26_tests.txt
R=$O/r06_s26_sd3_w8a8_ab.txt; : > $R
for r in 1 2; do
  for sw in w4f8 phased; do
    if [ $sw = phased ]; then export MI355X_SD_NO_W4=1; else unset MI355X_SD_NO_W4; fi
    MI355X_SD_LIB=dbg python bench.py --workload sd3-1024-bs8-w8a8 --steps 60 --warmup 5 --no-cpu-baseline --no-parity-mode --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $r  %-8s %6.2f steps/s  %6.2f ms/step' % ('$sw', d['value'], d['ms_per_step']))" >> $R
  done
done
unset MI355X_SD_NO_W4
cat $R
BENCH_SHAPES=1 timeout 300 python bench.py --workload sd3-1024-bs8-w8a8 --no-cpu-baseline --no-parity-mode --steps 10 2> $O/r06_s26_per_shape_sd3-1024-bs8-w8a8.txt > /tmp/b.json
grep "TFLOP/s" $O/r06_s26_per_shape_sd3-1024-bs8-w8a8.txt | head -8
