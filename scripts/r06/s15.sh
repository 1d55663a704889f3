#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp && rm -rf /tmp/pw && MI355X_SD_LIB=dbg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw -o w -- python $GRAFT_REPO_ROOT/scripts/r06/w4_which.py > /tmp/pw.log 2>&1
python - <<'PY'
import csv, glob, collections
agg = collections.OrderedDict()
for f in glob.glob('/tmp/pw/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'gemm' not in n: continue
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        a = agg.setdefault(n[:150], [0, 0.0]); a[0] += 1; a[1] += d
for n, (c, t) in agg.items(): print(f"n={c} avg_us={t/c:.1f} {n}")
PY
