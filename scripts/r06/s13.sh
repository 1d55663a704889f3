#!/bin/bash
# Round 6, session 13: small-tile ring kernel with convs; SD3 with / without the four-wave tile (debug build, switch set properly)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_gemm_variants.py tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -4 > $O/r06_s13_tests.txt
cat $O/r06_s13_tests.txt
L="-I/opt/rocm/include -Iinclude -Iscripts/c -Lpaddlemix_amd -lmi355x_sd_dbg -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/step_bench.c $L -o /tmp/step_bench || exit 1
R=$O/r06_s13_sd15_step_ab.txt; : > $R
for round in 1 2 3; do
  echo "round $round  small-tile ring (linear + conv) in the picker:" >> $R
  LD_LIBRARY_PATH=paddlemix_amd timeout 200 /tmp/step_bench scripts/c/sd15_unet_config.json 1 64 64 77 300 20 2>&1 | tail -1 | cut -c1-200 >> $R
  echo "round $round  MI355X_SD_NO_SMALL=1:" >> $R
  LD_LIBRARY_PATH=paddlemix_amd MI355X_SD_NO_SMALL=1 timeout 200 /tmp/step_bench scripts/c/sd15_unet_config.json 1 64 64 77 300 20 2>&1 | tail -1 | cut -c1-200 >> $R
done
cat $R
for wl in sd3-1024-bs8 sd3-1024-bs8-w8a8; do
  for round in 1 2; do
  MI355X_SD_LIB=dbg timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-parity-mode --steps 15 > /tmp/b.json 2>/tmp/b.err
  python - "$wl with the four-wave tile" <<'PY'
import json,sys
try:
    d=json.load(open("/tmp/b.json")); print(sys.argv[1], round(d["value"],3), "steps/s", round(d["ms_per_step"],2), d["kernel_breakdown_ms"])
except Exception as e: print(sys.argv[1], "ERR", e, open("/tmp/b.err").read()[-300:])
PY
  MI355X_SD_LIB=dbg MI355X_SD_NO_W4=1 timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-parity-mode --steps 15 > /tmp/b.json 2>/tmp/b.err
  python - "$wl MI355X_SD_NO_W4=1" <<'PY'
import json,sys
try:
    d=json.load(open("/tmp/b.json")); print(sys.argv[1], round(d["value"],3), "steps/s", round(d["ms_per_step"],2), d["kernel_breakdown_ms"])
except Exception as e: print(sys.argv[1], "ERR", e, open("/tmp/b.err").read()[-300:])
PY
  done
done 2>&1 | tee $O/r06_s13_sd3_w4_ab.txt
BENCH_SHAPES=1 timeout 300 python bench.py --workload sd15-512-bs1 --no-cpu-baseline --no-parity-mode --steps 50 2> $O/r06_s13_sd15_per_shape.txt > $O/r06_s13_sd15_bench.json
grep "TFLOP/s" $O/r06_s13_sd15_per_shape.txt | head -24
