#!/bin/bash
# Round 6, session 36: split-K reduce kernel with wave-sized blocks on small grids + 8 slab loads in flight: SD-1.5 step, tests
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Iscripts/c -Lpaddlemix_amd -lmi355x_sd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/step_bench.c $L -o /tmp/step_bench || exit 1
export LD_LIBRARY_PATH=paddlemix_amd
R=$O/r06_s36_reduce_blocks.txt; : > $R
for round in 1 2 3; do
  timeout 100 /tmp/step_bench scripts/c/sd15_unet_config.json 1 64 64 77 200 20 2>&1 | tail -1 | cut -c1-200 >> $R
done
cat $R
( timeout 1200 python -m pytest tests/test_gpu_gemm_variants.py tests/test_gpu_kernels.py tests/test_gpu_switches.py -x -q -m gpu ) > $O/r06_s36_tests.txt 2>&1; tail -4 $O/r06_s36_tests.txt
