#!/bin/bash
# round-2 GPU session K: LayerNorm grid A/B inside the step (MI355X_SD_LN_GRID="div,min,max"; default 2,256,512)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
: > $O/r02_k_ln_grid.txt
for cfg in 2,256,512 1,256,512 1,256,2048 2,256,1024 2,256,512; do
  MI355X_SD_LN_GRID=$cfg timeout 90 python bench.py --no-cpu-baseline --steps 20 > /tmp/b.json 2>/dev/null
  python - "$cfg" >> $O/r02_k_ln_grid.txt <<'PY'
import json,sys
try:
    d=json.load(open("/tmp/b.json")); print("LN_GRID", sys.argv[1], "steps/s", round(d["value"],3), "ms", round(d["ms_per_step"],3), "ln_ms", d["kernel_breakdown_ms"].get("ln"))
except Exception as e: print("LN_GRID", sys.argv[1], "ERR", e)
PY
done
cat $O/r02_k_ln_grid.txt
