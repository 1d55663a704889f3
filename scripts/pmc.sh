#!/bin/bash
# usage: pmc.sh <script.py> <kernel-name-substring> "<counters pass 1>" ["<counters pass 2>" ...]
# one rocprofv3 --pmc pass per counter list (never combined with sys/hip tracing); prints per-grid averages
cd /tmp && export TMPDIR=/tmp
S=$1; shift; KN=$1; shift
i=0
for C in "$@"; do
  i=$((i+1)); rm -rf /tmp/pmc$i
  rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc$i -o r -- python $GRAFT_REPO_ROOT/$S > /tmp/pmc$i.log 2>&1
  python - <<PY
import sqlite3,glob
dbs=glob.glob("/tmp/pmc$i/**/*.db", recursive=True)
if not dbs:
    print(open("/tmp/pmc$i.log").read()[-800:]); raise SystemExit
c=sqlite3.connect(dbs[0])
q="select grid_size_x, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%$KN%' group by grid_size_x, counter_name"
for r in c.execute(q): print("pass$i grid", r[0], r[1], f"{r[2]:.5g}", "n", r[3])
PY
done
