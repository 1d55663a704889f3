#!/bin/bash
# round-2 GPU session L: LayerNorm rows per wave and GroupNorm pixels per thread inside the step (env knobs), default grid now div=1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
: > $O/r02_l_norm_knobs.txt
run() {
  env "$@" timeout 90 python bench.py --no-cpu-baseline --steps 20 > /tmp/b.json 2>/dev/null
  python - "$*" >> $O/r02_l_norm_knobs.txt <<'PY'
import json,sys
try:
    d=json.load(open("/tmp/b.json")); k=d["kernel_breakdown_ms"]; print(sys.argv[1], "| steps/s", round(d["value"],3), "ms", round(d["ms_per_step"],3), "ln", k.get("ln"), "gn_stats", k.get("gn_stats"), "gn_apply", k.get("gn_apply"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
}
run X=default
run MI355X_SD_LN_ROWS=2
run MI355X_SD_LN_ROWS=2 MI355X_SD_LN_GRID=1,256,1024
run MI355X_SD_GN_ITERS=8
run MI355X_SD_GN_ITERS=32
run MI355X_SD_LN_GRID=1,256,2048
run X=default
timeout 120 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "layernorm or groupnorm or fp32_residual" 2>&1 | tail -2 >> $O/r02_l_norm_knobs.txt
cat $O/r02_l_norm_knobs.txt
