#!/bin/bash
# Round 5, session 23: the whole GPU suite and the default bench line on the final tree (after the weight-scale kernel instantiations)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu > $O/r05_h_pytest_gpu.txt 2>&1
tail -6 $O/r05_h_pytest_gpu.txt
python bench.py > $O/r05_h_bench.json 2> $O/r05_h_bench.err
python - <<PY
import json
d = json.load(open("$O/r05_h_bench.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "meets_target", "value_meeting_target", "board_during_timed_region")})
print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "achieved", "frac", "traffic")})
for leg in ("parity", "parity_mode"):
    print(leg, {k: d[leg].get(k) for k in ("dtype", "steps_per_s", "pred_rel_bs8", "meets_target", "error")})
PY
