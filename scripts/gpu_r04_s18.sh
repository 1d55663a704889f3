#!/bin/bash
# Round 4, session 18: first run of the torch-free attention and convolution probes (scripts/c/attn_probe.c, conv_probe.c).
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
L="-I/opt/rocm/include -Iinclude -Lpaddlemix_amd -lmi355x_sd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,/opt/rocm/lib"
gcc -std=c11 -O2 scripts/c/attn_probe.c $L -o /tmp/attn_probe || exit 1
gcc -std=c11 -O2 scripts/c/conv_probe.c $L -o /tmp/conv_probe || exit 1
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/paddlemix_amd:$LD_LIBRARY_PATH
{ timeout 12 /tmp/attn_probe 10; timeout 20 /tmp/conv_probe 5; } > $O/r04_s18_c_attn_conv_probe.txt 2>&1
cat $O/r04_s18_c_attn_conv_probe.txt
