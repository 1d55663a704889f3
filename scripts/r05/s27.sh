#!/bin/bash
# Round 5, session 27: does the HIP runtime's kernel-argument placement matter for the 926-launch hipGraph of the step?
# HIP_FORCE_DEV_KERNARG=0 / 1 (kernel arguments in host-coherent vs device memory), plain-C step bench, interleaved twice
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Iscripts/c -Lpaddlemix_amd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/step_bench.c $L -lmi355x_sd -o /tmp/step_bench || exit 1
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/paddlemix_amd:$LD_LIBRARY_PATH
{
  for r in 1 2; do for v in unset 0 1; do
    if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
    echo "== HIP_FORCE_DEV_KERNARG=$v (round $r)"; timeout 100 /tmp/step_bench scripts/c/sdxl_unet_config.json 8 128 128 77 30 3 | sed 's/"launches.*//'
  done; done
} > $O/r05_s27_kernarg_ab.txt 2>&1
cat $O/r05_s27_kernarg_ab.txt | cut -c1-220
