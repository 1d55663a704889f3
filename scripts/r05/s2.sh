#!/bin/bash
# Round 5, session 2: (a) do MFMA and exp / VALU work of one SIMD overlap (instruction-mix probe), (b) ablation of the pipelined attention loop
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Lpaddlemix_amd -lmi355x_sd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/attn_probe.c $L -o /tmp/attn_probe || exit 1
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/paddlemix_amd
timeout 120 build_exp/ovl > $O/r05_s2_overlap_probe.txt 2>&1
{
  for abl in 0 1 2 4 8 12 14 15 16 17 30 0; do echo "== MI355X_SD_ATTN_ABL=$abl"; MI355X_SD_ATTN_ABL=$abl timeout 100 /tmp/attn_probe 20 | grep "self\|#"; done
} > $O/r05_s2_attn_ablation.txt 2>&1
cat $O/r05_s2_overlap_probe.txt
cut -c1-120 $O/r05_s2_attn_ablation.txt
