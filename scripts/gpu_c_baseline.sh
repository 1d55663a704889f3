#!/bin/bash
# Reference run of the plain-C harness on the shipped library: the hashes and times later A/B sessions compare against (the probes'
# data generation changed at the end of round 4 -- hashes from profiles/r04_s12..s17 no longer apply to gemm_probe).
#   usage (on the GPU box, via gpurun):  bash scripts/gpu_c_baseline.sh <tag>      e.g. r05_s1     (~60 s of GPU-box time)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
T=${1:-c_baseline}
L="-I/opt/rocm/include -Iinclude -Lpaddlemix_amd -lmi355x_sd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
for p in gemm_probe conv_probe attn_probe step_bench; do gcc -std=c11 -O2 scripts/c/$p.c $L -o /tmp/$p || exit 1; done
export LD_LIBRARY_PATH=$R/paddlemix_amd:$LD_LIBRARY_PATH
{
  timeout 60 /tmp/gemm_probe 20
  timeout 60 /tmp/conv_probe 10
  timeout 60 /tmp/attn_probe 20
  timeout 100 /tmp/step_bench scripts/c/sdxl_unet_config.json 8 128 128 77 30 3
  timeout 60 /tmp/step_bench scripts/c/sd15_unet_config.json 1 64 64 77 200 10
} > $O/${T}_c_baseline.txt 2>&1
cat $O/${T}_c_baseline.txt
