#!/bin/bash
# Round 6, session 21: SD3 bs 8 in its three modes, interleaved three times on one box (60 timed steps each; no CPU / parity legs)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
: > $O/r06_s21_sd3_modes.txt
for r in 1 2 3; do
  for w in sd3-1024-bs8 sd3-1024-bs8-fp8w sd3-1024-bs8-w8a8; do
    python bench.py --workload $w --steps 60 --warmup 5 --no-cpu-baseline --no-parity-mode --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $r  %-20s %6.2f steps/s  %6.2f ms/step  board %s' % ('$w', d['value'], d['ms_per_step'], d.get('board_during_timed_region')))" >> $O/r06_s21_sd3_modes.txt
  done
done
cat $O/r06_s21_sd3_modes.txt
