#!/bin/bash
# Round 5, session 22: persistent GEMM blocks issue the next tile's prologue LDS-DMA ahead of the current tile's epilogue stores
# (gemm_pipe.hip `carried`). Bit-identity tests first; then A/B against the round-4 order (debug build, MI355X_SD_GEMM_NO_CARRY=1):
# linear shapes and conv shapes isolated (300 launches each), the SDXL bs-8 step (30 steps), interleaved twice.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_gemm_variants.py -q -m gpu -x > $O/r05_s22_pytest_variants.txt 2>&1
tail -5 $O/r05_s22_pytest_variants.txt | cut -c1-400
L="-I/opt/rocm/include -Iinclude -Iscripts/c -Lpaddlemix_amd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
for p in gemm_probe conv_probe step_bench; do gcc -std=c11 -O2 scripts/c/$p.c $L -lmi355x_sd_dbg -o /tmp/${p}_dbg || exit 1; done
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/paddlemix_amd:$LD_LIBRARY_PATH
{
  for r in 1 2; do for m in 1 0; do
    if [ $m = 1 ]; then export MI355X_SD_GEMM_NO_CARRY=1; else unset MI355X_SD_GEMM_NO_CARRY; fi
    echo "== linear shapes, MI355X_SD_GEMM_NO_CARRY=${MI355X_SD_GEMM_NO_CARRY:-unset} (round $r)"; timeout 100 /tmp/gemm_probe_dbg 300 | grep -v "^#" | cut -c1-100
    echo "== conv shapes, MI355X_SD_GEMM_NO_CARRY=${MI355X_SD_GEMM_NO_CARRY:-unset} (round $r)"; timeout 100 /tmp/conv_probe_dbg 100 | grep -v "^#" | cut -c1-110
  done; done
  for r in 1 2; do for m in 1 0; do
    if [ $m = 1 ]; then export MI355X_SD_GEMM_NO_CARRY=1; else unset MI355X_SD_GEMM_NO_CARRY; fi
    echo "== step, MI355X_SD_GEMM_NO_CARRY=${MI355X_SD_GEMM_NO_CARRY:-unset} (round $r)"; timeout 100 /tmp/step_bench_dbg scripts/c/sdxl_unet_config.json 8 128 128 77 30 3 | sed 's/"launches.*//'
  done; done
} > $O/r05_s22_carry_ab.txt 2>&1
cat $O/r05_s22_carry_ab.txt | cut -c1-200
