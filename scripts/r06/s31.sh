#!/bin/bash
# Round 6, session 31: rocprofv3 kernel table of the SD-1.5 bs-1 step (graph replay: durations without the per-launch event overhead)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp; rm -rf /tmp/p31
rocprofv3 --kernel-trace --stats -d /tmp/p31 -o r -- python $GRAFT_REPO_ROOT/bench.py --workload sd15-512-bs1 --steps 50 --warmup 5 --no-cpu-baseline --no-roofline --no-parity-mode > /tmp/p31.log 2>&1
DB=$(find /tmp/p31 -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $DB $O/r06_s31_sd15_bs1_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --workload sd15-512-bs1 --steps 50 ($(tail -1 /tmp/p31.log | cut -c1-120))" > /dev/null
head -40 $O/r06_s31_sd15_bs1_kernel_stats.txt | cut -c1-190
