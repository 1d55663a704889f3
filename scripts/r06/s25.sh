#!/bin/bash
# Round 6, session 25: the four-wave 256 x 160 tile per shape (isolated, sustained, interleaved): does it win where K is long?
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Iscripts/c -Lpaddlemix_amd -lmi355x_sd_dbg -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/gemm_probe.c $L -o /tmp/gemm_probe || exit 1
export LD_LIBRARY_PATH=paddlemix_amd
R=$O/r06_s25_w4_160_probe.txt; : > $R
MASK=0x2AA
for round in 1 2; do
echo "== round $round: picker (eight-wave 256 x 160)" >> $R
timeout 300 /tmp/gemm_probe 1500 $MASK 2>&1 | grep -v "^#" >> $R
echo "-- four-wave 256 x 160 (MI355X_SD_GEMM_TILE_MAP=160:259)" >> $R
MI355X_SD_GEMM_TILE_MAP="160:259" timeout 300 /tmp/gemm_probe 1500 $MASK 2>&1 | grep -v "^#" >> $R
done
cat $R
