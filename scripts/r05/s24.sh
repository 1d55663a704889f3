#!/bin/bash
# Round 5, session 24: rocprofv3 kernel table of SD3 bs 8 with e4m3 weights (what the widening pass and the weight-scale kernels cost)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp; rm -rf /tmp/p24
rocprofv3 --kernel-trace --stats -d /tmp/p24 -o r -- python $GRAFT_REPO_ROOT/bench.py --workload sd3-1024-bs8-fp8w --no-cpu-baseline --no-roofline > /tmp/p24.log 2>&1
DB=$(find /tmp/p24 -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $DB $O/r05_s24_sd3_fp8w_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --workload sd3-1024-bs8-fp8w --no-cpu-baseline --no-roofline   ($(tail -1 /tmp/p24.log | cut -c1-200))" > /dev/null
head -16 $O/r05_s24_sd3_fp8w_kernel_stats.txt | cut -c1-200
