#!/bin/bash
# Round 5, session 32: persistent blocks in the phased 256x256 kernel (16-bit, weight-scale and W8A8 forms): tests, then SD3 bs 8 in
# its three modes against MI355X_SD_GEMM_PERSIST=0 (debug build: one block per tile, also in the pipelined loops), interleaved
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 700 python -m pytest tests/test_gpu_sd3.py "tests/test_gpu_gemm_variants.py::test_epilogue_operand_variants" -q -m gpu -x > $O/r05_s32_pytest.txt 2>&1
tail -4 $O/r05_s32_pytest.txt | cut -c1-400
one() { python bench.py --workload $1 --no-cpu-baseline --no-roofline --no-parity-mode 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', d['value'], 'steps/s', d['ms_per_step'], 'ms')"; }
{ for r in 1 2; do for w in sd3-1024-bs8-w8a8 sd3-1024-bs8; do
    one $w "shipped"
    MI355X_SD_LIB=dbg one $w "dbg-persistent"
    MI355X_SD_LIB=dbg MI355X_SD_GEMM_PERSIST=0 one $w "dbg-one-block-per-tile"
  done; done; } > $O/r05_s32_persist256_ab.txt 2>&1
cat $O/r05_s32_persist256_ab.txt
