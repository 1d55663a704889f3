#!/bin/bash
# Round 5, session 18: weight-scale kernel instantiations (gemm_epilogue.h WS), 16-byte widening pass, phased 256x256 kernel on
# widened matrices, batched W8A8 epilogue -- tests first, then SD3 bs 8 in its three modes, interleaved, and the per-shape tables
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_gemm_variants.py tests/test_gpu_sd3.py tests/test_gpu_switches.py -q -m gpu > $O/r05_s18_pytest.txt 2>&1
tail -15 $O/r05_s18_pytest.txt | cut -c1-600
one() { python bench.py --workload $1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], 'steps/s', d['ms_per_step'], 'ms')"; }
{ for r in 1 2; do for w in sd3-1024-bs8 sd3-1024-bs8-fp8w sd3-1024-bs8-w8a8; do one $w; done; done; } > $O/r05_s18_sd3_modes.txt 2>&1
cat $O/r05_s18_sd3_modes.txt
for w in sd3-1024-bs8-fp8w sd3-1024-bs8-w8a8; do
  BENCH_SHAPES=1 python bench.py --workload $w --no-cpu-baseline 2> $O/r05_s18_per_shape_$w.txt > $O/r05_s18_bench_$w.json
  echo "== $w"; grep -E "n= " $O/r05_s18_per_shape_$w.txt | head -12
done
