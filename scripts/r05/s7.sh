#!/bin/bash
# Round 5, session 7: work per joule of the two MFMA shapes under the board's power cap (dense MFMA streams, random operands)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
( for i in $(seq 40); do echo "t=$i $(rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Average Graphics Package Power\|Current Socket" | tr '\n' ' ' | cut -c1-200)"; sleep 0.5; done ) > $O/r05_s7_smi.txt 2>&1 &
timeout 120 build_exp/shp 1.5 > $O/r05_s7_mfma_shape_energy.txt 2>&1
wait
cat $O/r05_s7_mfma_shape_energy.txt
sed -n 1,45p $O/r05_s7_smi.txt | cut -c1-160
