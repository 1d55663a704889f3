#!/bin/bash
# Round 5, session 9: board power and clock while ONLY the self-attention launch runs (old kernel, 16x16x32 kernel), rocm-smi at 4 Hz
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Lpaddlemix_amd -lmi355x_sd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 -DONLY_FIRST_SHAPE scripts/c/attn_probe.c $L -o /tmp/attn_probe1 || exit 1
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/paddlemix_amd
( for i in $(seq 120); do echo "t=$i $(rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Current Socket" | sed 's/.*(\([0-9]*Mhz\)).*/\1/; s/.*(W): //' | tr '\n' ' ')"; sleep 0.25; done ) > $O/r05_s9_smi.txt 2>&1 &
{
  for il in 0 8 0 8; do echo "== MI355X_SD_ATTN_IL=$il, 6000 launches of the S = 4096 self-attention shape"; MI355X_SD_ATTN_IL=$il timeout 100 /tmp/attn_probe1 6000 | grep "self"; sleep 1; done
} > $O/r05_s9_attn_long.txt 2>&1
wait
cat $O/r05_s9_attn_long.txt | cut -c1-130
cat $O/r05_s9_smi.txt | tr '\n' ';' | cut -c1-3000
