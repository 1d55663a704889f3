#!/bin/bash
# HBM traffic of the dominant kernel class via rocprofv3 PMC, collected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE --pmc passes (TCC slots), no other tracing; FETCH_SIZE is doubled (gfx950 counts 128-B requests
# as 64 B for wide coalesced reads), both are in KiB.  usage: traffic.sh <workload> [seconds per pass]
# Every pass runs under `timeout`: in round 3 one PMC pass of the whole bench did not come back within 20 minutes.
cd /tmp && export TMPDIR=/tmp
WL=${1:-sdxl-1024-bs8}
LIM=${2:-240}
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tr_$C
  timeout $LIM rocprofv3 --pmc $C --kernel-trace -d /tmp/tr_$C -o r -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-graph --no-parity-mode > /tmp/tr_$C.log 2>&1
  echo "pass $C rc=$?" >&2
done
python - <<PY
import sqlite3, glob, json
res = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    dbs = glob.glob(f"/tmp/tr_{C}/**/*.db", recursive=True)
    if not dbs:
        print(open(f"/tmp/tr_{C}.log").read()[-600:]); raise SystemExit(1)
    c = sqlite3.connect(dbs[0])
    q = "select kernel_name, count(*), sum(value) from counters_collection where counter_name = '%s' group by kernel_name" % C
    for name, n, tot in c.execute(q):
        if "sd::" not in name:
            continue
        key = ("gemm" if ("gemm" in name and "<true" not in name) else "conv" if "gemm" in name else
               "attn" if "attention" in name else "norm" if ("layernorm" in name or "gn_" in name or "scale_shift" in name or "adaln" in name) else "misc")
        d = res.setdefault(key, {"launches": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
        d[C] += tot
        if C == "FETCH_SIZE":
            d["launches"] += n
for k, d in res.items():
    rd = 2.0 * d["FETCH_SIZE"] * 1024      # gfx950 correction: x2, KiB -> B
    wr = d["WRITE_SIZE"] * 1024            # uncalibrated per the guide
    d["hbm_read_GB_total"] = rd / 1e9
    d["hbm_write_GB_total"] = wr / 1e9
    d["hbm_bytes_per_launch"] = (rd + wr) / max(d["launches"], 1)
print(json.dumps({"workload": "$WL", "forwards_profiled": "warm-up + 1 step (eager)", "classes": res}, indent=1))
PY
