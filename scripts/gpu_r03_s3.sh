#!/bin/bash
# round-3 GPU session 3: the epilogue rework (bias in the accumulators, residual fetched during the last K iterations, row-tile
# batches) and the four-wave tiles for two blocks per CU: (1) correctness of every variant, (2) time per launch of the step's
# dominant shapes per variant, (3) the launch timeline of the new default, (4) A/B inside the SDXL bs-8 step.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 420 python -m pytest tests/test_gpu_gemm_variants.py -m gpu -q -x 2>&1 | tail -15 > $O/r03_s3_tests.txt
cat $O/r03_s3_tests.txt
timeout 240 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "linear or conv3x3 or pipelined or geglu or layernorm_folded" 2>&1 | tail -6 >> $O/r03_s3_tests.txt
tail -6 $O/r03_s3_tests.txt
: > $O/r03_s3_variants.txt
v() {   # label, env assignments...
  local label=$1; shift
  env "$@" timeout 120 python scripts/gemm_variants.py --label "$label" 2>&1 | grep -v "amdgpu.ids" >> $O/r03_s3_variants.txt
}
v old        MI355X_SD_GEMM_NO_PRE=1 MI355X_SD_GEMM_NO_EPI_BATCH=1 MI355X_SD_GEMM_NO_BIAS_ACC=1
v new        X=0
v no_pre     MI355X_SD_GEMM_NO_PRE=1
v no_batch   MI355X_SD_GEMM_NO_EPI_BATCH=1
v no_biasacc MI355X_SD_GEMM_NO_BIAS_ACC=1
v t160to129  MI355X_SD_GEMM_TILE_MAP=160:129
v t160to161  MI355X_SD_GEMM_TILE_MAP=160:161,257:161
v t257to160  MI355X_SD_GEMM_TILE_MAP=257:160
v t257to129  MI355X_SD_GEMM_TILE_MAP=257:129,320:129
v t320to192  MI355X_SD_GEMM_TILE_MAP=320:192,257:192
v t320to160  MI355X_SD_GEMM_TILE_MAP=320:160
v old2       MI355X_SD_GEMM_NO_PRE=1 MI355X_SD_GEMM_NO_EPI_BATCH=1 MI355X_SD_GEMM_NO_BIAS_ACC=1
v new2       X=0
grep -v VARIANT_TIMES $O/r03_s3_variants.txt
timeout 200 python scripts/gemm_timeline.py > $O/r03_s3_gemm_timeline.txt 2>&1; echo "timeline rc=$?"
grep -v "amdgpu.ids" $O/r03_s3_gemm_timeline.txt
: > $O/r03_s3_step_ab.txt
run() {   # label, env assignments...
  local label=$1; shift
  env "$@" timeout 90 python bench.py --no-cpu-baseline --no-parity-mode --steps 20 > /tmp/b.json 2>/tmp/b.err
  python - "$label" >> $O/r03_s3_step_ab.txt <<'PY'
import json,sys
try:
    d=json.load(open("/tmp/b.json")); k=d["kernel_breakdown_ms"]; print(sys.argv[1], "| steps/s", round(d["value"],3), "ms", round(d["ms_per_step"],3), "gemm", k.get("gemm"), "conv", k.get("conv"), "attn", k.get("attn"))
except Exception as e: print(sys.argv[1], "ERR", e, open("/tmp/b.err").read()[-400:])
PY
}
run old MI355X_SD_GEMM_NO_PRE=1 MI355X_SD_GEMM_NO_EPI_BATCH=1 MI355X_SD_GEMM_NO_BIAS_ACC=1
run new X=0
run t160to129 MI355X_SD_GEMM_TILE_MAP=160:129
run t320to192 MI355X_SD_GEMM_TILE_MAP=320:192
run old MI355X_SD_GEMM_NO_PRE=1 MI355X_SD_GEMM_NO_EPI_BATCH=1 MI355X_SD_GEMM_NO_BIAS_ACC=1
run new X=0
run t257to160 MI355X_SD_GEMM_TILE_MAP=257:160
cat $O/r03_s3_step_ab.txt
