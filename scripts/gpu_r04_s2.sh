#!/bin/bash
# Round 4, session 1: the interleaved K loop (MI355X_SD_GEMM_IL) -- bit-identity on every tile family, isolated per-shape times,
# and the in-step per-shape tables (BENCH_SHAPES=1) of both loops and of every forced tile family: the data the tile rule is rebuilt from.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
T=r04_s2
( timeout 900 python -m pytest tests/test_gpu_gemm_variants.py -m gpu -q -x -k "interleaved" 2>&1 | tail -15 ) > $O/${T}_tests.txt
cat $O/${T}_tests.txt
: > $O/${T}_variants.txt
v() { local label=$1; shift; env "$@" timeout 150 python scripts/gemm_variants.py --label "$label" 2>&1 | grep -v "amdgpu.ids" >> $O/${T}_variants.txt; }
v il0 MI355X_SD_GEMM_IL=0
v il1 MI355X_SD_GEMM_IL=1
grep -v VARIANT_TIMES $O/${T}_variants.txt
: > $O/${T}_step_ab.txt
run() {   # label, env assignments...
  local label=$1; shift
  env BENCH_SHAPES=1 "$@" timeout 150 python bench.py --no-cpu-baseline --no-parity-mode --steps 20 > /tmp/b.json 2>/tmp/b.err
  grep -E "^  (gemm|conv|attn):" /tmp/b.err > $O/${T}_shapes_${label}.txt
  python - "$label" >> $O/${T}_step_ab.txt <<'PY'
import json,sys
try:
    d=json.load(open("/tmp/b.json")); k=d["kernel_breakdown_ms"]; print(sys.argv[1], "| steps/s", round(d["value"],3), "ms", round(d["ms_per_step"],3), " ".join(f"{a} {b}" for a,b in k.items()))
except Exception as e: print(sys.argv[1], "ERR", e, open("/tmp/b.err").read()[-400:])
PY
}
run il0_a MI355X_SD_GEMM_IL=0
run il1_a MI355X_SD_GEMM_IL=1
run il0_b MI355X_SD_GEMM_IL=0
run il1_b MI355X_SD_GEMM_IL=1
run il1_t160 MI355X_SD_GEMM_IL=1 MI355X_SD_GEMM_TILE=160
run il1_t257 MI355X_SD_GEMM_IL=1 MI355X_SD_GEMM_TILE=257
run il1_t320 MI355X_SD_GEMM_IL=1 MI355X_SD_GEMM_TILE=320
run il1_p160_1 MI355X_SD_GEMM_IL=1 MI355X_SD_GEMM_P160=1.0
run il1_noloaders MI355X_SD_GEMM_IL=1 MI355X_SD_GEMM_LOADERS=0
cat $O/${T}_step_ab.txt
timeout 200 python scripts/gemm_timeline.py 2>&1 | grep -v "amdgpu.ids" > $O/${T}_gemm_timeline_il1.txt
MI355X_SD_GEMM_IL=0 timeout 200 python scripts/gemm_timeline.py 2>&1 | grep -v "amdgpu.ids" > $O/${T}_gemm_timeline_il0.txt
cat $O/${T}_gemm_timeline_il1.txt
