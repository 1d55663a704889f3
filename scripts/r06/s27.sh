#!/bin/bash
# Round 6, session 27: gemm_w4f8 variants, per-shape table of SD3 W8A8 (production library)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
T=${1:-a}
BENCH_SHAPES=1 timeout 300 python bench.py --workload sd3-1024-bs8-w8a8 --no-cpu-baseline --no-parity-mode --steps 10 2> $O/r06_s27_${T}_per_shape_w8a8.txt > /tmp/b.json
grep "TFLOP/s" $O/r06_s27_${T}_per_shape_w8a8.txt | head -6
python -c "
import json; d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
