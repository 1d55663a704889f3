#!/bin/bash
# Round 5, session 31: the other workloads' bench lines on the last tree
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
for w in sd3-1024-bs8 sd3-1024-bs8-fp8w sd3-1024-bs8-w8a8; do
  timeout 120 python bench.py --workload $w --no-cpu-baseline --no-parity-mode > $O/r05_i_bench_$w.json 2>/dev/null
  python -c "import json;d=json.load(open('$O/r05_i_bench_$w.json'));print('$w', round(d['value'],3), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), (d.get('parity') or {}).get('end_latents_rel_l2'))"
done
timeout 100 python bench.py --workload sd15-512-bs1 --steps 50 --warmup 5 --no-cpu-baseline --no-parity-mode > $O/r05_i_bench_sd15-512-bs1.json 2>/dev/null
python -c "import json;d=json.load(open('$O/r05_i_bench_sd15-512-bs1.json'));print('sd15', round(d['value'],2), round(d['ms_per_step'],3))"
