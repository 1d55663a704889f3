#!/bin/bash
# round-3 GPU session 7: self-attention experiments (16-byte O stores, priority around P.V, staggered blocks), tile cost model
# (phased 256x256 penalised) on every bench workload
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "sdpa" 2>&1 | tail -4 > $O/r03_s7_tests.txt
cat $O/r03_s7_tests.txt
: > $O/r03_s7_attn.txt
a() { env "$@" timeout 100 python scripts/attn_self_probe.py 2>&1 | grep -v amdgpu.ids >> $O/r03_s7_attn.txt; }
a MI355X_SD_ATTN_NO_WIDE=1
a X=0
a MI355X_SD_ATTN_DBG=4
a MI355X_SD_ATTN_DBG=16
a MI355X_SD_ATTN_DBG=20
a MI355X_SD_ATTN_NO_WIDE=1
a X=0
cat $O/r03_s7_attn.txt
: > $O/r03_s7_step_ab.txt
run() {   # label, workload, env assignments...
  local label=$1; local wl=$2; shift; shift
  env "$@" timeout 120 python bench.py --workload $wl --no-cpu-baseline --no-parity-mode --steps 20 > /tmp/b.json 2>/tmp/b.err
  python - "$label" "$wl" >> $O/r03_s7_step_ab.txt <<'PY'
import json,sys
try:
    d=json.load(open("/tmp/b.json")); k=d["kernel_breakdown_ms"]; print(sys.argv[2], sys.argv[1], "| steps/s", round(d["value"],3), "ms", round(d["ms_per_step"],3), " ".join(f"{a} {b}" for a,b in k.items()))
except Exception as e: print(sys.argv[2], sys.argv[1], "ERR", e, open("/tmp/b.err").read()[-400:])
PY
}
for wl in sdxl-1024-bs8 sd15-512-bs1 sd3-1024-bs8 sd3-1024-bs8-w8a8; do
  run p257_1.00_nopersist $wl MI355X_SD_GEMM_P257=1.0 MI355X_SD_GEMM_PERSIST=0
  run p257_1.00 $wl MI355X_SD_GEMM_P257=1.0
  run p257_1.25 $wl X=0
  run p257_1.00 $wl MI355X_SD_GEMM_P257=1.0
  run p257_1.25 $wl X=0
done
run attn_prio sdxl-1024-bs8 MI355X_SD_ATTN_DBG=4
run attn_stagger sdxl-1024-bs8 MI355X_SD_ATTN_DBG=16
run attn_nowide sdxl-1024-bs8 MI355X_SD_ATTN_NO_WIDE=1
run default sdxl-1024-bs8 X=0
cat $O/r03_s7_step_ab.txt
