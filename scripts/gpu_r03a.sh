#!/bin/bash
# round-3 GPU session A (prepared at the end of round 2): the persistent-block streaming GEMM (csrc/gemm_persist.hip) --
# bit-for-bit probe against the one-tile-per-block kernel, then the A/B inside the SDXL step, then the kernel tests with it on.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/persist_probe.py > $O/r03_a_persist_probe.txt 2>&1; echo "probe rc=$?" >> $O/r03_a_persist_probe.txt
cat $O/r03_a_persist_probe.txt
: > $O/r03_a_persist_step.txt
for mode in 0 1 0 1; do
  MI355X_SD_GEMM_PERSIST=$mode timeout 60 python bench.py --no-cpu-baseline --steps 20 > /tmp/b.json 2>/dev/null
  python - "$mode" >> $O/r03_a_persist_step.txt <<'PY'
import json,sys
try:
    d=json.load(open("/tmp/b.json")); k=d["kernel_breakdown_ms"]; print("PERSIST", sys.argv[1], "| steps/s", round(d["value"],3), "ms", round(d["ms_per_step"],3), "gemm", k.get("gemm"), "conv", k.get("conv"))
except Exception as e: print("PERSIST", sys.argv[1], "ERR", e)
PY
done
MI355X_SD_GEMM_PERSIST=1 timeout 120 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "gemm or linear or conv3x3 or tiles or geglu" 2>&1 | tail -2 >> $O/r03_a_persist_step.txt
cat $O/r03_a_persist_step.txt
