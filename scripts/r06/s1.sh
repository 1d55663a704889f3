#!/bin/bash
# Round 6, session 1: same-box vendor yardstick (sustained), SD3 config-5 geometry fixtures on the device, bench lines with cpu_baseline.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 400 python scripts/blas_yardstick.py --seconds 1.0 --rounds 3 --out $O/r06_s1_blas_yardstick.txt 2>&1 | grep -v amdgpu.ids | tail -12
# hipBLASLt's kernel per shape (names only)
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_y -o y -- python $GRAFT_REPO_ROOT/scripts/blas_yardstick.py --seconds 0.05 --rounds 1 > /dev/null 2>&1 )
python - <<'PY' > $O/r06_s1_blas_kernels.txt 2>&1
import csv, glob, collections
fs = glob.glob('/tmp/prof_y/**/*kernel_trace.csv', recursive=True)
agg = collections.OrderedDict()
for f in fs:
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'Cijk' in n or 'gemm' in n.lower():
            k = (n[:400], r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Workgroup_Size_X') or r.get('Workgroup_Size'))
            d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            a = agg.setdefault(k, [0, 0.0, 1e9])
            a[0] += 1; a[1] += d; a[2] = min(a[2], d)
for (n, g, w), (c, t, mn) in agg.items():
    print(f"n={c:6d} avg_us={t/c:9.1f} min_us={mn:9.1f} grid={g} wg={w} {n}")
PY
cat $O/r06_s1_blas_kernels.txt | cut -c1-330
timeout 1500 python -m pytest tests/test_gpu_parity_loops.py -m gpu -q -x --durations=5 2>&1 | tail -15 > $O/r06_s1_parity_tests.txt
cat $O/r06_s1_parity_tests.txt
for wl in sd15-512-bs1 sd3-1024-bs8 sd3-1024-bs8-fp8w sd3-1024-bs8-w8a8; do
  timeout 600 python bench.py --workload $wl > $O/r06_s1_bench_$wl.json 2> $O/r06_s1_bench_$wl.err
  python - $O/r06_s1_bench_$wl.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(d["config"]["workload"], round(d["value"],2), "steps/s", round(d["ms_per_step"],3), "ms; cpu:", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("cores"), "parity:", d.get("parity"), "roofline:", d.get("roofline",{}).get("frac"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
