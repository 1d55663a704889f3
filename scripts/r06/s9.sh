#!/bin/bash
# Round 6, session 9: SD-1.5 bs-1 per-shape table, torch SDPA yardstick, comm / seam tests
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
BENCH_SHAPES=1 timeout 300 python bench.py --workload sd15-512-bs1 --no-cpu-baseline --no-parity-mode --steps 50 2> $O/r06_s9_sd15_per_shape.txt > $O/r06_s9_sd15_bench.json
grep "TFLOP/s\| ms " $O/r06_s9_sd15_per_shape.txt | head -70
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06_s9_sd15_bench.json")); print(d["value"], d["ms_per_step"], d["kernel_breakdown_ms"], d["roofline"]["launches_per_step"])
PY
timeout 300 python scripts/sdpa_yardstick.py --out $O/r06_s9_sdpa_yardstick.txt 2>&1 | grep -v amdgpu.ids | tail -5
timeout 600 python -m pytest tests/test_gpu_seams.py tests/test_gpu_cexec.py -m gpu -q -k "joint or comm" 2>&1 | tail -5
