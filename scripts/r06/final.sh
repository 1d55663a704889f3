#!/bin/bash
# Round 6, final measurement set on the final kernels: smoke, bench lines of every workload, rocprofv3 kernel table, PMC traffic
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_f_smoke.txt 2>&1; tail -3 gpurun_out/r06_f_smoke.txt
bash scripts/final_profile.sh r06_f
for f in gpurun_out/r06_f_bench*.json; do echo $f; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step", "meets_target", "value_meeting_target")}, "frac", d.get("roofline", {}).get("frac"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "pred_rel_bs8", d.get("pred_rel_bs8"))
except Exception as e:
    print("unreadable", e)
PY
done
