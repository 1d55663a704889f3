#!/bin/bash
# Round 6, session 23: the four-wave tile on widened e4m3 matrices (WS instantiations): variant tests, SD3 fp8w per-shape, modes A/B
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_gemm_variants.py tests/test_gpu_sd3.py tests/test_gpu_switches.py -x -q -m gpu ) > $O/r06_s23_tests.txt 2>&1; tail -5 $O/r06_s23_tests.txt
BENCH_SHAPES=1 timeout 300 python bench.py --workload sd3-1024-bs8-fp8w --no-cpu-baseline --no-parity-mode --steps 10 2> $O/r06_s23_per_shape_sd3-1024-bs8-fp8w.txt > /tmp/b.json
grep "TFLOP/s" $O/r06_s23_per_shape_sd3-1024-bs8-fp8w.txt | head -10
: > $O/r06_s23_sd3_modes.txt
for r in 1 2; do
  for w in sd3-1024-bs8 sd3-1024-bs8-fp8w; do
    python bench.py --workload $w --steps 60 --warmup 5 --no-cpu-baseline --no-parity-mode --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $r  %-20s %6.2f steps/s  %6.2f ms/step' % ('$w', d['value'], d['ms_per_step']))" >> $O/r06_s23_sd3_modes.txt
  done
done
cat $O/r06_s23_sd3_modes.txt
