#!/bin/bash
# round-3 GPU session 5: the early-residual form with its landing zone outside the register allocator, the short-key attention
# kernel: per-launch times, the launch timeline, A/B inside the SDXL bs-8 step
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "sdpa" 2>&1 | tail -4 > $O/r03_s5_tests.txt
cat $O/r03_s5_tests.txt
: > $O/r03_s5_attn.txt
a() { env "$@" timeout 100 python scripts/attn_short_probe.py 2>&1 | grep -v amdgpu.ids >> $O/r03_s5_attn.txt; }
a MI355X_SD_ATTN_NO_SHORT=1
a X=0
a MI355X_SD_ATTN_SHORT_QT=1
a MI355X_SD_ATTN_SHORT_QT=2
a MI355X_SD_ATTN_SHORT_QT=4
a MI355X_SD_ATTN_SHORT_QT=8
cat $O/r03_s5_attn.txt
: > $O/r03_s5_variants.txt
v() { local label=$1; shift; env "$@" timeout 120 python scripts/gemm_variants.py --label "$label" 2>&1 | grep -v "amdgpu.ids" >> $O/r03_s5_variants.txt; }
v old        MI355X_SD_GEMM_NO_PRE=1 MI355X_SD_GEMM_NO_EPI_BATCH=1 MI355X_SD_GEMM_NO_BIAS_ACC=1
v new        X=0
v lw0        MI355X_SD_GEMM_LOADERS=0
v t320to160  MI355X_SD_GEMM_TILE_MAP=320:160
grep -v VARIANT_TIMES $O/r03_s5_variants.txt
timeout 200 python scripts/gemm_timeline.py > $O/r03_s5_gemm_timeline.txt 2>&1; echo "timeline rc=$?"
grep -v "amdgpu.ids" $O/r03_s5_gemm_timeline.txt
: > $O/r03_s5_step_ab.txt
run() {   # label, env assignments...
  local label=$1; shift
  env "$@" timeout 90 python bench.py --no-cpu-baseline --no-parity-mode --steps 20 > /tmp/b.json 2>/tmp/b.err
  python - "$label" >> $O/r03_s5_step_ab.txt <<'PY'
import json,sys
try:
    d=json.load(open("/tmp/b.json")); k=d["kernel_breakdown_ms"]; print(sys.argv[1], "| steps/s", round(d["value"],3), "ms", round(d["ms_per_step"],3), "gemm", k.get("gemm"), "conv", k.get("conv"), "attn", k.get("attn"), "ln", k.get("layernorm"), "gn", k.get("groupnorm"))
except Exception as e: print(sys.argv[1], "ERR", e, open("/tmp/b.err").read()[-400:])
PY
}
run old MI355X_SD_GEMM_NO_PRE=1 MI355X_SD_GEMM_NO_EPI_BATCH=1 MI355X_SD_GEMM_NO_BIAS_ACC=1 MI355X_SD_ATTN_NO_SHORT=1
run new X=0
run new_noshort MI355X_SD_ATTN_NO_SHORT=1
run new_lw0 MI355X_SD_GEMM_LOADERS=0
run old MI355X_SD_GEMM_NO_PRE=1 MI355X_SD_GEMM_NO_EPI_BATCH=1 MI355X_SD_GEMM_NO_BIAS_ACC=1 MI355X_SD_ATTN_NO_SHORT=1
run new X=0
run new_t129 MI355X_SD_GEMM_TILE_MAP=160:129
run new_t320to160 MI355X_SD_GEMM_TILE_MAP=320:160
cat $O/r03_s5_step_ab.txt
