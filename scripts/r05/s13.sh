#!/bin/bash
# Round 5, session 13: the shipped state after the clean-up -- accuracy of the attention paths vs float64, the kernel / switch / variant
# GPU tests that touch attention and the debug-switch build, the shader clock of the 16x16x32 attention kernel, sustained attention rates
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Lpaddlemix_amd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/attn_check.c $L -lmi355x_sd -o /tmp/attn_check || exit 1
gcc -std=c11 -O2 scripts/c/attn_probe.c $L -lmi355x_sd -o /tmp/attn_probe || exit 1
gcc -std=c11 -O2 scripts/c/attn_probe.c $L -lmi355x_sd_dbg -o /tmp/attn_probe_dbg || exit 1
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/paddlemix_amd
{ echo "== accuracy vs float64, production library (bf16)"; timeout 120 /tmp/attn_check; } > $O/r05_s13_attn_check.txt 2>&1
{
  for r in 1 2; do
    echo "== production library (16x16x32 kernel), 3000 launches per shape (round $r)"; timeout 100 /tmp/attn_probe 3000
    echo "== debug-switch library, MI355X_SD_ATTN_NO_M16=1 (32x32x16 kernel), 3000 launches per shape (round $r)"; MI355X_SD_ATTN_NO_M16=1 timeout 100 /tmp/attn_probe_dbg 3000
  done
  echo "== debug-switch library, MI355X_SD_ATTN_STAMP=1: shader clock of the 16x16x32 kernel, 3000 launches per shape"; MI355X_SD_ATTN_STAMP=1 timeout 100 /tmp/attn_probe_dbg 3000
} > $O/r05_s13_attn_sustained.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_switches.py tests/test_gpu_gemm_variants.py tests/test_gpu_seams.py -x -q -m gpu > $O/r05_s13_pytest.txt 2>&1
tail -3 $O/r05_s13_attn_check.txt
grep -h "==\|self\|clock\|launches of one" $O/r05_s13_attn_sustained.txt | cut -c1-170
tail -15 $O/r05_s13_pytest.txt
