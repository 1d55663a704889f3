#!/bin/bash
# Round 4, session 17: ablation of the GEMM K loops (timing only, results are garbage): libraries built from the working tree with
# -DSD_ABL_NOWAIT (no counted vmcnt waits), -DSD_ABL_DMAL2 (every LDS-DMA piece re-reads the first K-tile), -DSD_ABL_NOEPI (no epilogue), MFMA-only + NOEPI.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Lpaddlemix_amd -lmi355x_sd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/gemm_probe.c $L -o /tmp/gemm_probe || exit 1
{
  for v in base nowait dmal2 noepi mfmanoepi; do
    echo "== $v"
    LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/build_exp/$v timeout 60 /tmp/gemm_probe 20
  done
} > $O/r04_s17_gemm_loop_ablation2.txt 2>&1
grep "^==\|^linear   8192\|^linear  32768x  640x 640\|shapes of" $O/r04_s17_gemm_loop_ablation2.txt | cut -c1-110
