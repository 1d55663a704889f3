#!/bin/bash
# Round 6, session 7: the four-wave tile in the picker -- whole-step A/B (plain-C step bench, debug build: MI355X_SD_NO_W4), tests, bench line
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Iscripts/c -Lpaddlemix_amd -lmi355x_sd_dbg -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/step_bench.c $L -o /tmp/step_bench || exit 1
export LD_LIBRARY_PATH=paddlemix_amd
R=$O/r06_s7_step_ab.txt; : > $R
for round in 1 2 3; do
  echo "round $round  picker with the four-wave tile:" >> $R
  timeout 200 /tmp/step_bench scripts/c/sdxl_unet_config.json 8 128 128 77 60 5 2>&1 | tail -1 | cut -c1-220 >> $R
  echo "round $round  MI355X_SD_NO_W4=1:" >> $R
  MI355X_SD_NO_W4=1 timeout 200 /tmp/step_bench scripts/c/sdxl_unet_config.json 8 128 128 77 60 5 2>&1 | tail -1 | cut -c1-220 >> $R
done
cat $R
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gemm_variants.py tests/test_gpu_cexec.py tests/test_gpu_unet.py tests/test_gpu_sd3.py -m gpu -q -x 2>&1 | tail -6 > $O/r06_s7_tests.txt
cat $O/r06_s7_tests.txt
timeout 900 python bench.py > $O/r06_s7_bench.json 2> $O/r06_s7_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06_s7_bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["kernel_breakdown_ms"], d.get("parity",{}).get("end_latents_rel_l2"), d.get("value_meeting_target"), d.get("board_during_timed_region"))
PY
BENCH_SHAPES=1 timeout 300 python bench.py --no-cpu-baseline --no-parity-mode --steps 10 2> $O/r06_s7_per_shape.txt > /dev/null
grep "TFLOP/s" $O/r06_s7_per_shape.txt | head -16
