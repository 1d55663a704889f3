#!/bin/bash
# round-3 GPU session 10: the -m gpu suite on the current code (without the two slow subprocess files), GroupNorm single-launch
# statistics with write-through partials: the step on SDXL / SD-1.5
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity_loops.py --deselect tests/test_gpu_gemm_variants.py 2>&1 | tail -25 > $O/r03_s10_tests.txt
cat $O/r03_s10_tests.txt
: > $O/r03_s10_steps.txt
run() {
  local label=$1; shift
  timeout 150 python bench.py --no-cpu-baseline --no-parity-mode --steps 20 "$@" > /tmp/b.json 2>/tmp/b.err
  python - "$label" >> $O/r03_s10_steps.txt <<'PY'
import json,sys
try:
    d=json.load(open("/tmp/b.json")); k=d["kernel_breakdown_ms"]; print(sys.argv[1], "| steps/s", round(d["value"],3), "ms", round(d["ms_per_step"],3), " ".join(f"{a} {b}" for a,b in k.items()), "| roofline", d["roofline"]["frac"] if d.get("roofline") else None)
except Exception as e: print(sys.argv[1], "ERR", e, open("/tmp/b.err").read()[-600:])
PY
}
run sdxl_bf16
run sd15 --workload sd15-512-bs1
run sdxl_bf16
run sd15 --workload sd15-512-bs1
cat $O/r03_s10_steps.txt
