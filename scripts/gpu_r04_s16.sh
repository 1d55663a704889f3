#!/bin/bash
# Round 4, session 16: ablation of the GEMM K loops (timing only, results are garbage): libraries built from the working tree with
# -DSD_ABL_NODMA (no LDS-DMA transfers), -DSD_ABL_NOREAD (fragments read once per tile), -DSD_ABL_NOBAR (no in-loop barriers), all three.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Lpaddlemix_amd -lmi355x_sd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/gemm_probe.c $L -o /tmp/gemm_probe || exit 1
{
  for v in base nodma noread nobar mfmaonly base; do
    echo "== $v"
    LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/build_exp/$v timeout 60 /tmp/gemm_probe 20
  done
} > $O/r04_s16_gemm_loop_ablation.txt 2>&1
grep "^==\|^linear   8192\|^linear  32768x  640x 640\|shapes of" $O/r04_s16_gemm_loop_ablation.txt | cut -c1-110
