#!/bin/bash
# Round 4, session 12: the step and the step's linear shapes from plain C through the C ABI (scripts/c/step_bench.c, gemm_probe.c):
# no torch in any process of this session.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
T=r04_s12
L="-I/opt/rocm/include -Iinclude -Lpaddlemix_amd -lmi355x_sd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/step_bench.c $L -o /tmp/step_bench || exit 1
gcc -std=c11 -O2 scripts/c/gemm_probe.c $L -o /tmp/gemm_probe || exit 1
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/paddlemix_amd:$LD_LIBRARY_PATH
{
  timeout 170 /tmp/step_bench scripts/c/sdxl_unet_config.json 8 128 128 77 30 3
  timeout 60 /tmp/step_bench scripts/c/sd15_unet_config.json 1 64 64 77 200 10
} > $O/${T}_c_step_bench.txt 2>&1
cat $O/${T}_c_step_bench.txt
{
  timeout 120 /tmp/gemm_probe 20
  echo "# MI355X_SD_NO_PIPE=1 (generic loop; its hashes differ: bias added last instead of first, see gemm_probe.c)"
  MI355X_SD_NO_PIPE=1 timeout 120 /tmp/gemm_probe 3
} > $O/${T}_c_gemm_probe.txt 2>&1
cat $O/${T}_c_gemm_probe.txt
cd /tmp && rm -rf /tmp/p12
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p12 -o r -- /tmp/step_bench $GRAFT_REPO_ROOT/scripts/c/sdxl_unet_config.json 8 128 128 77 10 2 > /tmp/p12.log 2>&1
DB=$(find /tmp/p12 -name '*.db' | head -1)
if [ -n "$DB" ]; then
  python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $DB $O/${T}_c_step_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- step_bench sdxl_unet_config.json 8 128 128 77 10 2   ($(grep c_abi_step /tmp/p12.log | cut -c1-160))" | head -25
else
  tail -20 /tmp/p12.log
fi
