#!/bin/bash
# usage: prof_gemm.sh <tile>   -- per-shape GEMM kernel durations via rocprofv3 (no python launch overhead in the numbers)
cd /tmp && export TMPDIR=/tmp
T=$1
rm -rf /tmp/p$T
MI355X_SD_GEMM_TILE=$T rocprofv3 --kernel-trace --stats -d /tmp/p$T -o r -- python $GRAFT_REPO_ROOT/scripts/gemm_probe.py > /tmp/p$T.log 2>&1
python - <<PY
import sqlite3,glob
dbs=glob.glob("/tmp/p$T/**/*.db", recursive=True)
if not dbs:
    print(open("/tmp/p$T.log").read()[-2000:])
    raise SystemExit
c=sqlite3.connect(dbs[0])
for r in c.execute("select name, grid_x, workgroup_x, count(*), avg(end-start)/1e3, min(end-start)/1e3 from kernels where name like '%gemm%' group by name, grid_x order by grid_x"):
    print("tile$T", r[0][20:80], "grid", r[1], "wg", r[2], "n", r[3], "avg_us", round(r[4],1), "min_us", round(r[5],1))
PY
