#!/bin/bash
# usage: prof_ksweep.sh <MxN> [tile]  -- per-K kernel durations (min of 5) via rocprofv3
cd /tmp && export TMPDIR=/tmp
export KSWEEP_MN=$1
[ -n "$2" ] && export MI355X_SD_GEMM_TILE=$2
D=/tmp/ks_$1_$2
rm -rf $D
rocprofv3 --kernel-trace --stats -d $D -o r -- python $GRAFT_REPO_ROOT/scripts/gemm_ksweep.py > $D.log 2>&1
python - <<PY
import sqlite3,glob
dbs=glob.glob("$D/**/*.db", recursive=True)
if not dbs:
    print(open("$D.log").read()[-2000:]); raise SystemExit
c=sqlite3.connect(dbs[0])
rows=list(c.execute("select name, grid_x, (end-start)/1e3 from kernels where name like '%gemm%' and name not like '%reduce%' order by start"))
Ks=(64,128,256,512,1024,1280,2560,5120)
for i,K in enumerate(Ks):
    ch=rows[5*i:5*i+5]
    if ch: print("$1 tile=${2:-auto} K=%5d  %-40s grid %6d  min_us %8.1f" % (K, ch[0][0][4:44], ch[0][1], min(r[2] for r in ch)))
PY
