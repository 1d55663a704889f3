#!/bin/bash
# Round 4, session 4: the cleaned-up library (interleaved loop only, no loader waves, 14 switches) -- variant / switch / kernel /
# model tests, the bench line, board power and clocks while the step runs, rocprofv3 kernel statistics, counters and traffic of the
# final kernels, the full-batch CPU baseline.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
T=r04_s4
( timeout 1500 python -m pytest tests/test_gpu_gemm_variants.py tests/test_gpu_switches.py tests/test_gpu_kernels.py -m gpu -q 2>&1 | tail -15 ) > $O/${T}_tests.txt
cat $O/${T}_tests.txt
( timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_seams.py tests/test_gpu_cexec.py -m gpu -q 2>&1 | tail -8 ) > $O/${T}_tests_models.txt
cat $O/${T}_tests_models.txt
# board power / clocks while the headline step runs (rocm-smi sampled next to a 200-step bench)
( timeout 200 python bench.py --no-cpu-baseline --no-parity-mode --no-roofline --steps 300 > $O/${T}_bench_300.json 2>/dev/null & )
sleep 45
for i in $(seq 1 12); do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|edge)" ; sleep 1; done > $O/${T}_power.txt 2>&1
sleep 20
cat $O/${T}_power.txt | head -40
tail -c 600 $O/${T}_bench_300.json
# the bench line
BENCH_SHAPES=1 timeout 900 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err; grep -E "^  (gemm|conv|attn):" $O/${T}_bench.err > $O/${T}_shapes.txt; tail -c 2500 $O/${T}_bench.json
# rocprofv3 kernel statistics of the same command (no cpu baseline / parity legs: kernels of the timed loop + roofline pass)
cd /tmp && rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity-mode --no-roofline > /tmp/prof.log 2>&1
cd $GRAFT_REPO_ROOT; DB=$(find /tmp/prof -name "*.db" | head -1)
python scripts/rocprof_summary.py $DB $O/${T}_sdxl_bs8_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-parity-mode --no-roofline   ($(grep '^{' /tmp/prof.log | tail -1 | cut -c1-200))" | head -30
# HBM traffic per kernel class (FETCH_SIZE / WRITE_SIZE in separate passes)
timeout 900 bash scripts/traffic.sh sdxl-1024-bs8 > $O/${T}_traffic_sdxl-1024-bs8.json 2> $O/${T}_traffic.err; tail -30 $O/${T}_traffic_sdxl-1024-bs8.json
# full-batch CPU baseline (2 timed steps of SDXL 8x4x128x128 on the host cores)
timeout 1500 python scripts/cpu_baseline.py --steps 2 --out $O/r04_cpu_baseline_sdxl-1024-bs8.json > $O/${T}_cpu_baseline.log 2>&1; tail -3 $O/${T}_cpu_baseline.log
