#!/bin/bash
# round-3 GPU session 1: (1) changed-code checks (attention guard in both builds, C handle life cycle), (2) where a K = 1280 GEMM
# launch's time goes (per-block time stamps), (3) the persistent-block streaming GEMM probe + A/B inside the step, (4) loader waves
# selected by shape, A/B inside the step.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp16.py tests/test_gpu_cexec.py -m gpu -q -x -k "sdpa or kernels or cexec or handle or linear_plain" 2>&1 | tail -5 > $O/r03_s1_tests.txt
cat $O/r03_s1_tests.txt
timeout 200 python scripts/gemm_timeline.py > $O/r03_s1_gemm_timeline.txt 2>&1; echo "timeline rc=$?"
cat $O/r03_s1_gemm_timeline.txt
timeout 300 python scripts/persist_probe.py > $O/r03_s1_persist_probe.txt 2>&1; echo "probe rc=$?" >> $O/r03_s1_persist_probe.txt
tail -30 $O/r03_s1_persist_probe.txt
: > $O/r03_s1_step_ab.txt
run() {   # label, env assignments...
  local label=$1; shift
  env "$@" timeout 90 python bench.py --no-cpu-baseline --no-parity-mode --steps 20 > /tmp/b.json 2>/tmp/b.err
  python - "$label" >> $O/r03_s1_step_ab.txt <<'PY'
import json,sys
try:
    d=json.load(open("/tmp/b.json")); k=d["kernel_breakdown_ms"]; print(sys.argv[1], "| steps/s", round(d["value"],3), "ms", round(d["ms_per_step"],3), "gemm", k.get("gemm"), "conv", k.get("conv"), "attn", k.get("attn"))
except Exception as e: print(sys.argv[1], "ERR", e, open("/tmp/b.err").read()[-400:])
PY
}
run base X=0
run persist MI355X_SD_GEMM_PERSIST=1
run lw_auto MI355X_SD_GEMM_LOADERS=-1
run base X=0
run persist MI355X_SD_GEMM_PERSIST=1
run lw_auto MI355X_SD_GEMM_LOADERS=-1
cat $O/r03_s1_step_ab.txt
