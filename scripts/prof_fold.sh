#!/bin/bash
# rocprofv3 kernel trace of bench.py in hipGraph mode with / without LayerNorm folding; per (kernel, grid) durations
cd /tmp && export TMPDIR=/tmp
for e in 0 1; do
  if [ $e = 0 ]; then export MI355X_SD_LNFOLD=1; else unset MI355X_SD_LNFOLD; fi
  rm -rf /tmp/pf$e
  rocprofv3 --kernel-trace --stats -d /tmp/pf$e -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > /tmp/pf$e.log 2>&1
  tail -1 /tmp/pf$e.log | cut -c1-200
  python - <<PY
import sqlite3,glob
dbs=glob.glob("/tmp/pf$e/**/*.db", recursive=True)
c=sqlite3.connect(dbs[0])
tot=c.execute("select sum(end-start)/1e6, count(*) from kernels").fetchone()
print("fold" if $e==0 else "nofold", "total kernel ms", round(tot[0],1), "dispatches", tot[1])
rows=c.execute("select name, grid_x, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels group by name, grid_x order by 4 desc limit 26").fetchall()
for r in rows:
    print(f"  {r[0][:70]:70s} grid {r[1]:8d} n {r[2]:5d} total {r[3]:9.2f} ms avg {r[4]:8.1f} us")
PY
done
