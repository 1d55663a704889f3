#!/bin/bash
# round-3 GPU session 9: the -m gpu suite on the current code (without the two slow subprocess files), then the step on every
# workload + the residual-stream modes
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity_loops.py --deselect tests/test_gpu_gemm_variants.py 2>&1 | tail -12 > $O/r03_s9_tests.txt
cat $O/r03_s9_tests.txt
: > $O/r03_s9_steps.txt
run() {   # label, bench args..., (env via ENVV)
  local label=$1; shift
  env $ENVV timeout 150 python bench.py --no-cpu-baseline --no-parity-mode --steps 20 "$@" > /tmp/b.json 2>/tmp/b.err
  python - "$label" >> $O/r03_s9_steps.txt <<'PY'
import json,sys
try:
    d=json.load(open("/tmp/b.json")); k=d["kernel_breakdown_ms"]; print(sys.argv[1], "| steps/s", round(d["value"],3), "ms", round(d["ms_per_step"],3), " ".join(f"{a} {b}" for a,b in k.items()), "| roofline", d["roofline"]["frac"] if d.get("roofline") else None)
except Exception as e: print(sys.argv[1], "ERR", e, open("/tmp/b.err").read()[-600:])
PY
}
ENVV="X=0"
run sdxl_bf16
run sdxl_bf16_resid32 --residual fp32
run sdxl_fp16 --dtype fp16
run sdxl_fp16_resid32 --dtype fp16 --residual fp32
run sd15 --workload sd15-512-bs1
run sd3 --workload sd3-1024-bs8
run sd3_fp8w --workload sd3-1024-bs8-fp8w
run sd3_w8a8 --workload sd3-1024-bs8-w8a8
run sdxl_bf16
ENVV="MI355X_SD_GEMM_NO_PRE=1 MI355X_SD_GEMM_NO_EPI_BATCH=1 MI355X_SD_GEMM_NO_BIAS_ACC=1 MI355X_SD_ATTN_NO_SHORT=1 MI355X_SD_GEMM_PERSIST=0 MI355X_SD_ATTN_NO_WIDE=1"
run sdxl_bf16_round2_kernels
run sdxl_bf16_resid32_round2_kernels --residual fp32
cat $O/r03_s9_steps.txt
