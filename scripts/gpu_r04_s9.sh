#!/bin/bash
# Round 4, session 9 (last): the bench line of the final tree (with the board clock / power sampled inside the timed region) and the
# rocprofv3 kernel statistics of the same command.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
T=r04_g
BENCH_SHAPES=1 timeout 900 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err; grep -E "^  (gemm|conv|attn):" $O/${T}_bench.err > $O/${T}_per_shape_ms.txt; tail -c 2200 $O/${T}_bench.json
cd /tmp && rm -rf /tmp/prof && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity-mode --no-roofline > /tmp/prof.log 2>&1
cd $GRAFT_REPO_ROOT; DB=$(find /tmp/prof -name "*.db" | head -1)
python scripts/rocprof_summary.py $DB $O/${T}_sdxl_bs8_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-parity-mode --no-roofline   ($(grep '^{' /tmp/prof.log | tail -1 | cut -c1-200))" | head -14
