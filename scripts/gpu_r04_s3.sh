#!/bin/bash
# Round 4, session 3: schedules of the interleaved loop (A/B isolated + in the step), matrix-pipe / LDS counters of the
# 256x160 kernel (old loop vs interleaved), the round-3 K limit of the 256x160 weight, the bench line with its live parity legs,
# the RCCL code path with one rank, full-depth parity loops.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
T=r04_s3
( timeout 600 python -m pytest tests/test_gpu_gemm_variants.py -m gpu -q -x -k "schedules" 2>&1 | tail -5 ) > $O/${T}_tests.txt
cat $O/${T}_tests.txt
: > $O/${T}_variants.txt
v() { local label=$1; shift; env "$@" timeout 150 python scripts/gemm_variants.py --label "$label" --only to_out,qkv,ff2,out640,ff2_640,qkv640 2>&1 | grep -v "amdgpu.ids" >> $O/${T}_variants.txt; }
for i in 1 2 3 4 1 2 3 4; do v il$i MI355X_SD_GEMM_IL=$i; done
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
run il1 MI355X_SD_GEMM_IL=1
run il2 MI355X_SD_GEMM_IL=2
run il3 MI355X_SD_GEMM_IL=3
run il4 MI355X_SD_GEMM_IL=4
run il1_k1536 MI355X_SD_GEMM_IL=1 MI355X_SD_GEMM_K160=1536
run il1_b MI355X_SD_GEMM_IL=1
run il0_r3rule MI355X_SD_GEMM_IL=0 MI355X_SD_GEMM_K160=1536 MI355X_SD_GEMM_LOADERS=-1
cat $O/${T}_step_ab.txt
# counters of the 256x160 kernel, old loop vs interleaved (plain launch, no residual): two passes each
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
P3="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE"
for il in 0 1; do
  export MI355X_SD_GEMM_IL=$il GEMM_SHAPE=8192x3840x1280
  ( echo "== MI355X_SD_GEMM_IL=$il  gemm 8192x3840x1280 (256x160 tile, 768 tiles on 256 persistent blocks)"; timeout 400 bash scripts/pmc.sh scripts/gemm_one.py gemm_pipe_kernel "$P1" "$P2" "$P3" ) > $O/${T}_pmc_il$il.txt 2>&1
  cat $O/${T}_pmc_il$il.txt
done
unset MI355X_SD_GEMM_IL GEMM_SHAPE
# the bench line, with its live parity legs; then the multi-rank code path with one rank
timeout 900 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err; tail -c 3000 $O/${T}_bench.json
timeout 600 python bench.py --gpus 1 --force-dist --no-parity-mode --no-roofline > $O/${T}_bench_force_dist.json 2> $O/${T}_bench_force_dist.err; tail -c 1500 $O/${T}_bench_force_dist.json; tail -3 $O/${T}_bench_force_dist.err
timeout 1700 python scripts/parity_loops.py --out $O/r04_parity.json > $O/${T}_parity_loops.log 2>&1; tail -5 $O/${T}_parity_loops.log | cut -c1-1500
