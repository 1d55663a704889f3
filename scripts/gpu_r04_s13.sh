#!/bin/bash
# Round 4, sessions 13+: A/B of builds of the library from plain C (no torch): build_exp/<name>/libmi355x_sd.so, in turn.
#   usage: gpu_r04_s13.sh <tag> <rounds> <name> [<name> ...]
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
T=$1; R=$2; shift 2
L="-I/opt/rocm/include -Iinclude -Lpaddlemix_amd -lmi355x_sd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/step_bench.c $L -o /tmp/step_bench || exit 1
gcc -std=c11 -O2 scripts/c/gemm_probe.c $L -o /tmp/gemm_probe || exit 1
{
  for v in "$@"; do
    echo "== $v: linear shapes"
    LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/build_exp/$v timeout 100 /tmp/gemm_probe 20
  done
  for r in $(seq $R); do
    for v in "$@"; do
      echo "== $v: SDXL 1024^2 bs 8 step (round $r)"
      LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/build_exp/$v timeout 100 /tmp/step_bench scripts/c/sdxl_unet_config.json 8 128 128 77 30 3
    done
  done
} > $O/${T}_ab.txt 2>&1
grep -v "^linear   \|^linear  3\|^linear 1\|^#" $O/${T}_ab.txt | cut -c1-250
