#!/bin/bash
# Round 5, session 17: the lazy-maximum guard test at the ABI; per-shape table of SD3 bs 8 with 16-bit and with e4m3 weights
# (where do the 4-5 ms of the weight-only mode sit?)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "sdpa" > $O/r05_s17_pytest_sdpa.txt 2>&1
tail -4 $O/r05_s17_pytest_sdpa.txt
for w in sd3-1024-bs8 sd3-1024-bs8-fp8w; do
  BENCH_SHAPES=1 python bench.py --workload $w --no-cpu-baseline 2> $O/r05_s17_per_shape_$w.txt > /dev/null
  echo "== $w"; grep -E "n= " $O/r05_s17_per_shape_$w.txt | head -24
done
