#!/bin/bash
# Round 6, session 11: vendor yardstick on the batch-1 SD-1.5 linear shapes (+ the names of the kernels it picks)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 400 python scripts/blas_yardstick.py --sd15 --seconds 0.5 --rounds 3 --out $O/r06_s11_blas_yardstick_sd15.txt 2>&1 | grep -v amdgpu.ids | tail -12
cd /tmp && rm -rf /tmp/prof_y && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_y -o y -- python $GRAFT_REPO_ROOT/scripts/blas_yardstick.py --sd15 --seconds 0.02 --rounds 1 > /tmp/prof_y.log 2>&1
ls -R /tmp/prof_y | head -20
python - <<'PY' > $O/r06_s11_blas_kernels_sd15.txt 2>&1
import csv, glob, collections
fs = glob.glob('/tmp/prof_y/**/*kernel_trace.csv', recursive=True)
agg = collections.OrderedDict()
for f in fs:
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        k = (n[:300], r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Workgroup_Size_X') or r.get('Workgroup_Size'))
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        a = agg.setdefault(k, [0, 0.0, 1e9])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d)
for (n, g, w), (c, t, mn) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"n={c:6d} avg_us={t/c:9.1f} min_us={mn:9.1f} grid={g} wg={w} {n}")
PY
head -40 $O/r06_s11_blas_kernels_sd15.txt | cut -c1-260
