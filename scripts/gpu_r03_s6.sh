#!/bin/bash
# round-3 GPU session 6: persistent blocks in the pipelined GEMM (multi-round launches): correctness, per-launch A/B, timeline, step A/B
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "linear or conv3x3 or pipelined or geglu or layernorm_folded or sdpa" 2>&1 | tail -4 > $O/r03_s6_tests.txt
timeout 400 python -m pytest tests/test_gpu_gemm_variants.py -m gpu -q -k "epilogue_operand" 2>&1 | tail -6 >> $O/r03_s6_tests.txt
cat $O/r03_s6_tests.txt
: > $O/r03_s6_variants.txt
v() { local label=$1; shift; env "$@" timeout 120 python scripts/gemm_variants.py --label "$label" 2>&1 | grep -v "amdgpu.ids" >> $O/r03_s6_variants.txt; }
v one_block_per_tile MI355X_SD_GEMM_PERSIST=0
v persistent         X=0
v persist_t257to160  MI355X_SD_GEMM_TILE_MAP=257:160
v one_block_per_tile MI355X_SD_GEMM_PERSIST=0
v persistent         X=0
grep -v VARIANT_TIMES $O/r03_s6_variants.txt
timeout 200 python scripts/gemm_timeline.py > $O/r03_s6_gemm_timeline.txt 2>&1; echo "timeline rc=$?"
grep -v "amdgpu.ids" $O/r03_s6_gemm_timeline.txt
: > $O/r03_s6_step_ab.txt
run() {   # label, env assignments...
  local label=$1; shift
  env "$@" timeout 90 python bench.py --no-cpu-baseline --no-parity-mode --steps 20 > /tmp/b.json 2>/tmp/b.err
  python - "$label" >> $O/r03_s6_step_ab.txt <<'PY'
import json,sys
try:
    d=json.load(open("/tmp/b.json")); k=d["kernel_breakdown_ms"]; print(sys.argv[1], "| steps/s", round(d["value"],3), "ms", round(d["ms_per_step"],3), " ".join(f"{a} {b}" for a,b in k.items()))
except Exception as e: print(sys.argv[1], "ERR", e, open("/tmp/b.err").read()[-400:])
PY
}
run one_block_per_tile MI355X_SD_GEMM_PERSIST=0
run persistent X=0
run persist_t257to160 MI355X_SD_GEMM_TILE_MAP=257:160
run one_block_per_tile MI355X_SD_GEMM_PERSIST=0
run persistent X=0
run persist_t257to160 MI355X_SD_GEMM_TILE_MAP=257:160
cat $O/r03_s6_step_ab.txt
