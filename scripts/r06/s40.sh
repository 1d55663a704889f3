#!/bin/bash
# Round 6, session 40: LayerNorm folded into the consuming GEMMs (MI355X_SD_LNFOLD=1, a builder option since round 1) on today's kernels
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
R=$O/r06_s40_lnfold_ab.txt; : > $R
for r in 1 2; do
  for arm in default lnfold; do
    if [ $arm = lnfold ]; then export MI355X_SD_LNFOLD=1; else unset MI355X_SD_LNFOLD; fi
    python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-parity-mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $r  %-8s %7.3f ms/step  %s' % ('$arm', d['ms_per_step'], d.get('kernel_breakdown_ms')))" >> $R
  done
done
unset MI355X_SD_LNFOLD
cat $R
