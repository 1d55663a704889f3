#!/bin/bash
# Round 6, session 8: attention16 with the row sums on the matrix pipe (MI355X_SD_ATTN_RS, debug build): sustained A/B + accuracy
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Iscripts/c -Lpaddlemix_amd -lmi355x_sd_dbg -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/attn_probe.c $L -o /tmp/attn_probe || exit 1
gcc -std=c11 -O2 scripts/c/attn_check.c $L -o /tmp/attn_check || exit 1
export LD_LIBRARY_PATH=paddlemix_amd
R=$O/${1:-r06_s8}_attn_rs.txt; : > $R
for round in 1 2 3; do
  echo "== round $round: shipped" >> $R
  timeout 300 /tmp/attn_probe 1500 2>&1 | grep -v "^#" >> $R
  echo "-- MI355X_SD_ATTN_RS=1" >> $R
  MI355X_SD_ATTN_RS=1 timeout 300 /tmp/attn_probe 1500 2>&1 | grep -v "^#" >> $R
done
echo "== accuracy vs host float64 (shipped)" >> $R
timeout 600 /tmp/attn_check >> $R 2>&1
echo "== accuracy vs host float64 (MI355X_SD_ATTN_RS=1)" >> $R
MI355X_SD_ATTN_RS=1 timeout 600 /tmp/attn_check >> $R 2>&1
cat $R
