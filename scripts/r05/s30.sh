#!/bin/bash
# Round 5, session 30: the default bench command on the final tree (stdout must be the JSON line alone)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
python bench.py > $O/r05_i_bench.json 2> $O/r05_i_bench.err
echo "stdout lines: $(wc -l < $O/r05_i_bench.json)"
python - <<PY
import json
d = json.load(open("$O/r05_i_bench.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "meets_target", "value_meeting_target", "board_during_timed_region")})
print("roofline", {k: d["roofline"].get(k) for k in ("achieved", "frac")}, "cpu_baseline", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "kind")})
for leg in ("parity", "parity_mode"):
    print(leg, {k: d[leg].get(k) for k in ("dtype", "steps_per_s", "pred_rel_bs8", "meets_target", "error")})
PY
