#!/bin/bash
# Round 6, session 10: split-K finished inside the launch -- bit-identity to the two-pass form, SD-1.5 bs-1 step A/B
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_gemm_variants.py tests/test_gpu_switches.py tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -5 > $O/r06_s10_tests.txt
cat $O/r06_s10_tests.txt
L="-I/opt/rocm/include -Iinclude -Iscripts/c -Lpaddlemix_amd -lmi355x_sd_dbg -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/step_bench.c $L -o /tmp/step_bench || exit 1
export LD_LIBRARY_PATH=paddlemix_amd
R=$O/r06_s10_sd15_step_ab.txt; : > $R
for round in 1 2 3; do
  echo "round $round  split-K finished inside the launch:" >> $R
  timeout 200 /tmp/step_bench scripts/c/sd15_unet_config.json 1 64 64 77 300 20 2>&1 | tail -1 | cut -c1-200 >> $R
  echo "round $round  MI355X_SD_SPLITK_2PASS=1:" >> $R
  MI355X_SD_SPLITK_2PASS=1 timeout 200 /tmp/step_bench scripts/c/sd15_unet_config.json 1 64 64 77 300 20 2>&1 | tail -1 | cut -c1-200 >> $R
done
cat $R
timeout 600 python -m pytest tests/test_gpu_cexec.py tests/test_gpu_export.py tests/test_gpu_unet.py -m gpu -q -x 2>&1 | tail -4
