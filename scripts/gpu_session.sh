#!/bin/bash
# Template of a measurement session on the GPU box (round 3 ran thirteen of these; their outputs are under profiles/r03_s*):
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash scripts/gpu_session.sh'
# Every step under `timeout`; A/B variants of one library through the MI355X_SD_* switches (each read once per process):
#   MI355X_SD_GEMM_NO_PRE / _NO_EPI_BATCH / _NO_BIAS_ACC   round-2 epilogue forms          MI355X_SD_GEMM_PERSIST=0   one block per tile
#   MI355X_SD_GEMM_TILE=<id> / _TILE_MAP=from:to,...       tile families (128 129 160 256 257 320)   MI355X_SD_NO_PIPE=1   generic loop
#   MI355X_SD_ATTN_NO_SHORT / _NO_QT / _NO_WIDE     MI355X_SD_NO_GN_FUSED / _NO_SPLITK / _NO_WIDEN_F8      (the 15 switches left after
#   round 4; each is exercised by tests/test_gpu_gemm_variants.py or tests/test_gpu_switches.py. The one-off session scripts of rounds 3-5 are in the history only; round 6: scripts/r06/)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
TAG=${1:-session}
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q 2>&1 | tail -4 > $O/${TAG}_tests.txt
cat $O/${TAG}_tests.txt
: > $O/${TAG}_variants.txt
v() { local label=$1; shift; env "$@" timeout 120 python scripts/gemm_variants.py --label "$label" 2>&1 | grep -v "amdgpu.ids" >> $O/${TAG}_variants.txt; }
v round2_epilogues MI355X_SD_GEMM_NO_PRE=1 MI355X_SD_GEMM_NO_EPI_BATCH=1 MI355X_SD_GEMM_NO_BIAS_ACC=1 MI355X_SD_GEMM_PERSIST=0
v default X=0
grep -v VARIANT_TIMES $O/${TAG}_variants.txt
timeout 200 python scripts/gemm_timeline.py 2>&1 | grep -v "amdgpu.ids" > $O/${TAG}_gemm_timeline.txt
cat $O/${TAG}_gemm_timeline.txt
: > $O/${TAG}_step_ab.txt
run() {   # label, env assignments...
  local label=$1; shift
  env "$@" timeout 90 python bench.py --no-cpu-baseline --no-parity-mode --steps 20 > /tmp/b.json 2>/tmp/b.err
  python - "$label" >> $O/${TAG}_step_ab.txt <<'PY'
import json,sys
try:
    d=json.load(open("/tmp/b.json")); k=d["kernel_breakdown_ms"]; print(sys.argv[1], "| steps/s", round(d["value"],3), "ms", round(d["ms_per_step"],3), " ".join(f"{a} {b}" for a,b in k.items()))
except Exception as e: print(sys.argv[1], "ERR", e, open("/tmp/b.err").read()[-400:])
PY
}
run round2_kernels MI355X_SD_GEMM_NO_PRE=1 MI355X_SD_GEMM_NO_EPI_BATCH=1 MI355X_SD_GEMM_NO_BIAS_ACC=1 MI355X_SD_GEMM_PERSIST=0 MI355X_SD_ATTN_NO_SHORT=1 MI355X_SD_ATTN_NO_WIDE=1 MI355X_SD_NO_GN_FUSED=1
run default X=0
run round2_kernels MI355X_SD_GEMM_NO_PRE=1 MI355X_SD_GEMM_NO_EPI_BATCH=1 MI355X_SD_GEMM_NO_BIAS_ACC=1 MI355X_SD_GEMM_PERSIST=0 MI355X_SD_ATTN_NO_SHORT=1 MI355X_SD_ATTN_NO_WIDE=1 MI355X_SD_NO_GN_FUSED=1
run default X=0
cat $O/${TAG}_step_ab.txt
