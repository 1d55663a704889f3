#!/bin/bash
# Round 5, session 20: which shapes gain / lose on the co-resident four-wave 128x160 tiles (id 129)? In the step the whole 160
# family on them costs +1.9 ms (r05_s12_step_tile129.txt); per shape, isolated, 300 launches each, interleaved twice.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Iscripts/c -Lpaddlemix_amd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/gemm_probe.c $L -lmi355x_sd_dbg -o /tmp/gemm_probe_dbg || exit 1
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/paddlemix_amd:$LD_LIBRARY_PATH
{
  for r in 1 2; do for m in "" "160:129" "160:129,320:129"; do
    echo "== MI355X_SD_GEMM_TILE_MAP=$m (round $r)"; MI355X_SD_GEMM_TILE_MAP=$m timeout 100 /tmp/gemm_probe_dbg 300
  done; done
} > $O/r05_s20_gemm_tile129_per_shape.txt 2>&1
cut -c1-150 $O/r05_s20_gemm_tile129_per_shape.txt
