#!/bin/bash
# Round 6, session 22: where weight-only fp8 loses 6 ms to 16-bit weights on SD3 bs 8: per-shape tables of both + kernel names
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
for w in sd3-1024-bs8 sd3-1024-bs8-fp8w; do
  BENCH_SHAPES=1 timeout 300 python bench.py --workload $w --no-cpu-baseline --no-parity-mode --steps 10 2> $O/r06_s22_per_shape_$w.txt > /tmp/b.json
  echo "== $w"; grep "TFLOP/s" $O/r06_s22_per_shape_$w.txt | head -10
done
cd /tmp; rm -rf /tmp/p22
rocprofv3 --kernel-trace --stats -d /tmp/p22 -o r -- python $GRAFT_REPO_ROOT/bench.py --workload sd3-1024-bs8-fp8w --no-cpu-baseline --no-parity-mode --no-roofline --steps 10 > /tmp/p22.log 2>&1
DB=$(find /tmp/p22 -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $DB $O/r06_s22_sd3_fp8w_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --workload sd3-1024-bs8-fp8w --steps 10" > /dev/null
head -16 $O/r06_s22_sd3_fp8w_kernel_stats.txt | cut -c1-200
