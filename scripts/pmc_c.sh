#!/bin/bash
# rocprofv3 counters for one of the plain-C probes (scripts/c/*.c): no Python in the profiled process, so a pass costs seconds.
#   usage: pmc_c.sh <probe: gemm_probe|conv_probe|attn_probe|step_bench> "<probe args>" <kernel-name-substring> "<counters pass 1>" ["<pass 2>" ...]
#   e.g.   scripts/pmc_c.sh gemm_probe 3 gemm_pipe "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"
# One --pmc pass per counter list, never combined with sys / hip tracing (the pool's rule); FETCH_SIZE and WRITE_SIZE in separate
# passes (MI355X_MICROARCH.md). Prints, per kernel and grid size, the average of every counter over the launches.
# PMC_LIBDIR decides which directory's build of the library is profiled (default: the shipped one); PMC_LIB=mi355x_sd_dbg links the
# debug-switch build (the only one that reads the MI355X_SD_* A/B switches).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
P=$1; A=$2; KN=$3; shift 3
L="-I/opt/rocm/include -I$R/include -L$R/paddlemix_amd -l${PMC_LIB:-mi355x_sd} -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 $R/scripts/c/$P.c $L -o /tmp/$P || exit 1
export LD_LIBRARY_PATH=${PMC_LIBDIR:-$R/paddlemix_amd}${LD_LIBRARY_PATH:+:$LD_LIBRARY_PATH}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "$@"; do
  i=$((i+1)); rm -rf /tmp/pmcc$i
  timeout 120 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmcc$i -o r -- /tmp/$P $A > /tmp/pmcc$i.log 2>&1
  python3 - <<PY
import sqlite3, glob
dbs = glob.glob("/tmp/pmcc$i/**/*.db", recursive=True)
if not dbs:
    print(open("/tmp/pmcc$i.log").read()[-800:]); raise SystemExit
c = sqlite3.connect(dbs[0])
q = ("select kernel_name, grid_size_x, counter_name, avg(value), count(*) from counters_collection "
     "where kernel_name like '%$KN%' group by kernel_name, grid_size_x, counter_name order by kernel_name, grid_size_x")
for name, grid, cn, v, n in c.execute(q):
    print(f"pass$i {name[:70]:70s} grid {grid:7d} {cn:28s} {v:14.6g}  n {n}")
PY
done
