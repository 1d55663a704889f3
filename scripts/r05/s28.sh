#!/bin/bash
# Round 5, session 28: conv shapes on the alternative tiles, per shape, isolated (picker check for the conv class)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Iscripts/c -Lpaddlemix_amd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/conv_probe.c $L -lmi355x_sd_dbg -o /tmp/conv_probe_dbg || exit 1
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/paddlemix_amd:$LD_LIBRARY_PATH
{
  for r in 1 2; do for m in "" "320:160,257:160" "320:257" "257:320"; do
    echo "== MI355X_SD_GEMM_TILE_MAP=$m (round $r)"; MI355X_SD_GEMM_TILE_MAP=$m timeout 100 /tmp/conv_probe_dbg 100 | grep -v "^#" | cut -c1-100
  done; done
} > $O/r05_s28_conv_tile_alt_per_shape.txt 2>&1
cat $O/r05_s28_conv_tile_alt_per_shape.txt
