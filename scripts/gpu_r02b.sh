#!/bin/bash
# round-2 GPU session B: handle API (C executor + plain-C client), attention8 probe, conv K-order A/B, fp16 parity
set -x
O=gpurun_out/r02b; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python scripts/attn8_probe.py --iters 10 > $O/attn8.log 2>&1; echo "rc=$?" >> $O/attn8.log
timeout 400 python -m pytest tests/test_gpu_cexec.py -q -x -s > $O/cexec.log 2>&1; echo "rc=$?" >> $O/cexec.log
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "kb64 or ragged_kv_tail or sdpa" > $O/kernels.log 2>&1; echo "rc=$?" >> $O/kernels.log
timeout 150 python bench.py --no-cpu-baseline --steps 20 > $O/bench_kb64.json 2> $O/bench_kb64.err
MI355X_SD_NO_KB64=1 timeout 150 python bench.py --no-cpu-baseline --steps 20 > $O/bench_nokb64.json 2> $O/bench_nokb64.err
PARITY_SKIP_SD15=1 timeout 420 python scripts/parity_report.py --out $O/parity.json > $O/parity.log 2>&1; echo "rc=$?" >> $O/parity.log
tail -3 $O/cexec.log $O/kernels.log; grep -v ATTN8_JSON $O/attn8.log | tail -60
