#!/bin/bash
# Round 5, session 16: attention accuracy on the IEEE-half build (guard at 2^15); weight-only fp8 on SD3 with the context stream's
# large launches on widened matrices too (MI355X_SD_WIDEN_F8_MIN_M, debug build) -- in-step A/B; then the whole GPU suite once more
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Lpaddlemix_amd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/attn_check.c $L -lmi355x_sd_f16 -o /tmp/attn_check16 || exit 1
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/paddlemix_amd
{ echo "== accuracy vs float64, IEEE-half library"; timeout 120 /tmp/attn_check16; } > $O/r05_s16_attn_check_f16.txt 2>&1
tail -2 $O/r05_s16_attn_check_f16.txt
{
  for r in 1 2; do for m in 4096 1024; do
    echo "== sd3-1024-bs8-fp8w, debug-switch library, MI355X_SD_WIDEN_F8_MIN_M=$m (round $r)"
    MI355X_SD_LIB=dbg MI355X_SD_WIDEN_F8_MIN_M=$m python bench.py --workload sd3-1024-bs8-fp8w --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'steps/s', d['ms_per_step'], 'ms')"
  done; done
  echo "== sd3-1024-bs8 (16-bit weights), same session"; python bench.py --workload sd3-1024-bs8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'steps/s', d['ms_per_step'], 'ms')"
} > $O/r05_s16_sd3_fp8w_min_m.txt 2>&1
cat $O/r05_s16_sd3_fp8w_min_m.txt
timeout 1500 python -m pytest tests -q -m gpu > $O/r05_s16_pytest_gpu.txt 2>&1
tail -6 $O/r05_s16_pytest_gpu.txt
python bench.py > $O/r05_g_bench.json 2> $O/r05_g_bench.err
python - <<PY
import json
d = json.load(open("$O/r05_g_bench.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "meets_target", "value_meeting_target", "board_during_timed_region")})
print("roofline", {k: d["roofline"].get(k) for k in ("achieved", "frac", "traffic", "traffic_source")})
for leg in ("parity", "parity_mode"):
    print(leg, {k: d[leg].get(k) for k in ("dtype", "steps_per_s", "end_latents_rel_l2", "pred_rel_bs8", "pred_rel_bs8_per_prompt_max", "meets_target", "error")})
PY
