#!/bin/bash
# round-2 GPU session H: LayerNorm load-issue fix (one s_waitcnt per row group again), the fused-op entry points on hardware,
# the default bench line + rocprofv3 kernel stats on the final code
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 240 python -m pytest tests/test_gpu_sd3.py tests/test_gpu_kernels.py -m gpu -x -q -k "fused_adaln or split_concat or layernorm or norm or adaln" > $O/r02_h_pytest_fused_norm.log 2>&1
tail -3 $O/r02_h_pytest_fused_norm.log
timeout 240 python bench.py > $O/r02_h_bench.json 2> $O/r02_h_bench.err
timeout 120 python bench.py --residual fp32 --no-cpu-baseline > $O/r02_h_bench_resid_fp32.json 2>/dev/null
cd /tmp
rm -rf /tmp/pfin
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pfin -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline > /tmp/pfin.log 2>&1
DB=$(find /tmp/pfin -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $DB $O/r02_h_sdxl_bs8_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-roofline   ($(tail -1 /tmp/pfin.log | cut -c1-160))" > /dev/null
cd $GRAFT_REPO_ROOT
BENCH_SHAPES=1 timeout 120 python bench.py --no-cpu-baseline 2> $O/r02_h_per_shape_ms.txt > /dev/null
tail -c 900 $O/r02_h_bench.json; echo; tail -c 300 $O/r02_h_bench_resid_fp32.json; echo; head -16 $O/r02_h_sdxl_bs8_kernel_stats.txt | cut -c1-170
