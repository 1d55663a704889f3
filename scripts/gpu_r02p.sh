#!/bin/bash
# round-2 GPU session P: GEMM rasterisation default -4 (column groups of 4): neighbours, GEMM / conv kernel tests, small UNet tests
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
: > $O/r02_p_gemm_gm.txt
for gm in default -3 -5 -6 -16 8 default; do
  if [ $gm = default ]; then unset MI355X_SD_GEMM_GM; else export MI355X_SD_GEMM_GM=$gm; fi
  timeout 60 python bench.py --no-cpu-baseline --steps 20 > /tmp/b.json 2>/dev/null
  python - "$gm" >> $O/r02_p_gemm_gm.txt <<'PY'
import json,sys
try:
    d=json.load(open("/tmp/b.json")); k=d["kernel_breakdown_ms"]; print("GEMM_GM", sys.argv[1], "| steps/s", round(d["value"],3), "ms", round(d["ms_per_step"],3), "gemm", k.get("gemm"), "conv", k.get("conv"), "frac", round(d["roofline"]["frac"],4))
except Exception as e: print("GEMM_GM", sys.argv[1], "ERR", e)
PY
done
unset MI355X_SD_GEMM_GM
timeout 60 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "gemm or linear or conv3x3 or tiles" 2>&1 | tail -2 >> $O/r02_p_gemm_gm.txt
timeout 60 python -m pytest tests/test_gpu_unet.py tests/test_gpu_cexec.py -m gpu -q -k "small_unet or independ or plain_c" 2>&1 | tail -2 >> $O/r02_p_gemm_gm.txt
cat $O/r02_p_gemm_gm.txt
