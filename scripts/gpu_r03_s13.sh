#!/bin/bash
# round-3 GPU session 13: single-launch GroupNorm for small (batch, group) chunks: correctness + A/B on SD-1.5 bs 1 / SDXL bs 8
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_cexec.py tests/test_gpu_unet.py tests/test_gpu_vae.py -m gpu -q -k "groupnorm or handle or plain_c or unet or vae or norms" 2>&1 | tail -8 > $O/r03_s13_tests.txt
cat $O/r03_s13_tests.txt
: > $O/r03_s13_step_ab.txt
run() {
  local label=$1; local wl=$2; shift; shift
  env "$@" timeout 120 python bench.py --workload $wl --no-cpu-baseline --no-parity-mode --steps 30 > /tmp/b.json 2>/tmp/b.err
  python - "$label" "$wl" >> $O/r03_s13_step_ab.txt <<'PY'
import json,sys
try:
    d=json.load(open("/tmp/b.json")); k=d["kernel_breakdown_ms"]; print(sys.argv[2], sys.argv[1], "| steps/s", round(d["value"],3), "ms", round(d["ms_per_step"],3), " ".join(f"{a} {b}" for a,b in k.items()))
except Exception as e: print(sys.argv[2], sys.argv[1], "ERR", e, open("/tmp/b.err").read()[-400:])
PY
}
for wl in sd15-512-bs1 sdxl-1024-bs8; do
  run gn_pair $wl MI355X_SD_NO_GN_FUSED=1
  run gn_fused $wl X=0
  run gn_pair $wl MI355X_SD_NO_GN_FUSED=1
  run gn_fused $wl X=0
done
cat $O/r03_s13_step_ab.txt
