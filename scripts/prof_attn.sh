#!/bin/bash
# per-shape attention kernel durations via rocprofv3 (scripts/attn_probe.py)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pa
rocprofv3 --kernel-trace --stats -d /tmp/pa -o r -- python $GRAFT_REPO_ROOT/scripts/attn_probe.py > /tmp/pa.log 2>&1
python - <<PY
import sqlite3,glob
dbs=glob.glob("/tmp/pa/**/*.db", recursive=True)
c=sqlite3.connect(dbs[0])
for r in c.execute("select name, grid_x, count(*), avg(end-start)/1e3, min(end-start)/1e3 from kernels where name like '%attention%' group by name, grid_x order by grid_x"):
    print("attn", r[0][:50], "grid", r[1], "n", r[2], "avg_us", round(r[3],1), "min_us", round(r[4],1))
PY
