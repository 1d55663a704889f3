#!/bin/bash
# Round 5, session 29: mi355x_sd_set_workspace refuses host memory -- the kernel tests (incl. the refusal), the paths that bind a
# workspace around hipGraph capture (UNet, C executor, exported programs)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_cexec.py tests/test_gpu_export.py tests/test_gpu_unet.py -q -m gpu -x > $O/r05_s29_pytest.txt 2>&1
tail -5 $O/r05_s29_pytest.txt | cut -c1-400
L="-I/opt/rocm/include -Iinclude -Iscripts/c -Lpaddlemix_amd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/step_bench.c $L -lmi355x_sd -o /tmp/step_bench && LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/paddlemix_amd:$LD_LIBRARY_PATH timeout 100 /tmp/step_bench scripts/c/sdxl_unet_config.json 8 128 128 77 10 2 | cut -c1-200
