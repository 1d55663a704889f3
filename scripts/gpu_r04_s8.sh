#!/bin/bash
# Round 4, session 8: the exported-program tests three times over (the CPU-planned VAE program failed once in session 6: the
# emulator had under-sized its GroupNorm workspace), then the full-batch CPU baseline (2 timed steps of SDXL 8x4x128x128).
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
T=r04_s8
for i in 1 2 3; do ( timeout 300 python -m pytest tests/test_gpu_export.py -m gpu -q 2>&1 | tail -2 | cut -c1-200 ); done > $O/${T}_export_x3.txt 2>&1; cat $O/${T}_export_x3.txt
timeout 1000 python -u scripts/cpu_baseline.py --steps 2 --chunk 1 --out $O/r04_cpu_baseline_sdxl-1024-bs8.json > $O/${T}_cpu_baseline.log 2>&1; tail -5 $O/${T}_cpu_baseline.log | cut -c1-400
