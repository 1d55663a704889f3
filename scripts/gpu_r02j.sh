#!/bin/bash
# round-2 GPU session J: the whole -m gpu suite on the final kernels, then the bench line + rocprofv3 kernel stats
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 640 python -m pytest tests -m gpu -q --durations=8 > $O/r02_j_pytest_gpu.log 2>&1
echo "rc=$?" >> $O/r02_j_pytest_gpu.log
tail -14 $O/r02_j_pytest_gpu.log | cut -c1-200
timeout 200 python bench.py > $O/r02_j_bench.json 2> $O/r02_j_bench.err
cd /tmp
rm -rf /tmp/pfin
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pfin -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline > /tmp/pfin.log 2>&1
DB=$(find /tmp/pfin -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $DB $O/r02_j_sdxl_bs8_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-roofline   ($(tail -1 /tmp/pfin.log | cut -c1-160))" > /dev/null
cd $GRAFT_REPO_ROOT
timeout 100 python bench.py --workload sd15-512-bs1 --steps 50 --warmup 5 --no-cpu-baseline > $O/r02_j_bench_sd15_bs1.json 2>/dev/null
python - <<'PY'
import json,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/"
for f in ("r02_j_bench.json","r02_j_bench_sd15_bs1.json"):
    try:
        d=json.load(open(O+f)); print(f, d["value"], d["ms_per_step"], d.get("kernel_breakdown_ms"), d.get("parity",{}).get("forward_rel_l2"))
    except Exception as e: print(f, "ERR", e)
PY
grep -E "scale_shift|gn_partial|gn_final|conv_in|conv_out|layernorm" $O/r02_j_sdxl_bs8_kernel_stats.txt | cut -c1-150
