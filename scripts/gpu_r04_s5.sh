#!/bin/bash
# Round 4, session 5: which kernel test aborts (session 4 lost the head of that log), the variant / switch tests incl. the fp8
# weight widening, the SD3 lines (16-bit / weight-only fp8 / W8A8), SD-1.5 bs 1, board power + clocks sampled WHILE the step runs.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
T=r04_s5
timeout 600 python -X faulthandler -m pytest tests/test_gpu_kernels.py -m gpu -x -v 2>&1 | grep -v "amdgpu.ids" > /tmp/k.log
( head -150 /tmp/k.log | cut -c1-200; echo ...; tail -40 /tmp/k.log | cut -c1-200 ) > $O/${T}_tests_kernels.txt
grep -E "PASSED|FAILED|ERROR|Fatal|fault|passed|failed" $O/${T}_tests_kernels.txt | tail -12
( timeout 900 python -m pytest tests/test_gpu_switches.py tests/test_gpu_gemm_variants.py -m gpu -q 2>&1 | tail -30 | cut -c1-300 ) > $O/${T}_tests_variants.txt
tail -12 $O/${T}_tests_variants.txt
# power / clocks while 200 steps run: the sampler starts first and outlives the bench
( for i in $(seq 1 150); do echo "t=$i $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power \(W\)|sclk' | sed 's/.*: //' | tr '\n' ' ')"; sleep 0.5; done > $O/${T}_power.txt ) &
timeout 200 python bench.py --no-cpu-baseline --no-parity-mode --no-roofline --steps 400 > $O/${T}_bench_400.json 2>/dev/null
wait
awk '{print}' $O/${T}_power.txt | sort -t= -k2 -n | awk 'NR%6==0' | head -30
for wl in sd3-1024-bs8 sd3-1024-bs8-fp8w sd3-1024-bs8-w8a8 sd15-512-bs1; do
  timeout 240 python bench.py --workload $wl --no-cpu-baseline --no-parity-mode --steps 20 > $O/${T}_bench_$wl.json 2> $O/${T}_bench_$wl.err
  python - $O/${T}_bench_$wl.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(d["config"]["workload"], round(d["value"],3), "steps/s", round(d["ms_per_step"],3), "ms", d.get("kernel_breakdown_ms"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
