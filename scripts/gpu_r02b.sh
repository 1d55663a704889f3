#!/bin/bash
# round-2 GPU session B: handle API (C executor + plain-C client), attention8 probe, conv K-order, bench A/B
set -x
O=gpurun_out/r02b; mkdir -p $O
export TMPDIR=/tmp
timeout 420 python -m pytest tests/test_gpu_cexec.py -q -x -s > $O/cexec.log 2>&1; echo "rc=$?" >> $O/cexec.log
timeout 300 python scripts/attn8_probe.py --iters 10 > $O/attn8.log 2>&1; echo "rc=$?" >> $O/attn8.log
MI355X_SD_ATTN8=1 timeout 240 python -m pytest tests/test_gpu_kernels.py -q -x -k "sdpa" > $O/sdpa_attn8.log 2>&1; echo "rc=$?" >> $O/sdpa_attn8.log
MI355X_SD_ATTN8=9 timeout 240 python -m pytest tests/test_gpu_kernels.py -q -x -k "sdpa" > $O/sdpa_attn8_lazy.log 2>&1; echo "rc=$?" >> $O/sdpa_attn8_lazy.log
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv3x3 or sdpa or fp32_residual" > $O/kernels.log 2>&1; echo "rc=$?" >> $O/kernels.log
timeout 200 python bench.py --no-cpu-baseline --steps 20 > $O/bench_kb64.json 2> $O/bench_kb64.err
MI355X_SD_NO_KB64=1 timeout 200 python bench.py --no-cpu-baseline --steps 20 > $O/bench_nokb64.json 2> $O/bench_nokb64.err
tail -3 $O/cexec.log $O/sdpa_attn8.log $O/sdpa_attn8_lazy.log $O/kernels.log; grep ATTN8_JSON -v $O/attn8.log | tail -50
