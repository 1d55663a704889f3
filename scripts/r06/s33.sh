#!/bin/bash
# Round 6, session 33: split-K constants 160:416 as the default -- SD3 modes against 128:512 (debug build), tests, SD-1.5 bench line
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
R=$O/r06_s33_splitk_sd3_ab.txt; : > $R
for r in 1 2; do
  for w in sd3-1024-bs8 sd3-1024-bs8-fp8w sd3-1024-bs8-w8a8 sdxl-1024-bs8; do
    for pol in 160:416 128:512; do
      MI355X_SD_SPLITK_POLICY=$pol MI355X_SD_LIB=dbg python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline --no-parity-mode --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $r  %-20s policy $pol  %7.3f ms/step' % ('$w', d['ms_per_step']))" >> $R
    done
  done
done
cat $R
( timeout 1200 python -m pytest tests/test_gpu_gemm_variants.py tests/test_gpu_kernels.py tests/test_gpu_switches.py -x -q -m gpu ) > $O/r06_s33_tests.txt 2>&1; tail -4 $O/r06_s33_tests.txt
python bench.py --workload sd15-512-bs1 --steps 50 --warmup 5 > $O/r06_s33_bench_sd15-512-bs1.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/r06_s33_bench_sd15-512-bs1.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
