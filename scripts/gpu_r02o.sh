#!/bin/bash
# round-2 GPU session O: GEMM tile rasterisation group inside the step (MI355X_SD_GEMM_GM; 8 = production, <0 = column groups)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
: > $O/r02_o_gemm_gm.txt
for gm in 8 4 16 2 32 -8 -4 -2 8; do
  MI355X_SD_GEMM_GM=$gm timeout 60 python bench.py --no-cpu-baseline --steps 20 > /tmp/b.json 2>/dev/null
  python - "$gm" >> $O/r02_o_gemm_gm.txt <<'PY'
import json,sys
try:
    d=json.load(open("/tmp/b.json")); k=d["kernel_breakdown_ms"]; print("GEMM_GM", sys.argv[1], "| steps/s", round(d["value"],3), "ms", round(d["ms_per_step"],3), "gemm", k.get("gemm"), "conv", k.get("conv"))
except Exception as e: print("GEMM_GM", sys.argv[1], "ERR", e)
PY
done
MI355X_SD_GEMM_GM=-8 timeout 100 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "gemm or linear or conv3x3 or tiles" 2>&1 | tail -2 >> $O/r02_o_gemm_gm.txt
cat $O/r02_o_gemm_gm.txt
