#!/bin/bash
# round-3 GPU session 2: new tests (masks, 16-bit modulation vectors, GEMM variants), the GEMM launch timeline, two concurrent
# half-batch chains, the full-depth parity loops
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_unet.py tests/test_gpu_sd3.py tests/test_gpu_gemm_variants.py -m gpu -q -k "self_attention_mask or modulation or encoder_attention_mask or fused_adaln or loader" 2>&1 | tail -8 > $O/r03_s2_tests.txt
cat $O/r03_s2_tests.txt
timeout 200 python scripts/gemm_timeline.py > $O/r03_s2_gemm_timeline.txt 2>&1; echo "timeline rc=$?"
cat $O/r03_s2_gemm_timeline.txt
timeout 300 python scripts/two_stream_probe.py > $O/r03_s2_two_streams.txt 2>&1; echo "two-stream rc=$?"
cat $O/r03_s2_two_streams.txt
timeout 1500 python scripts/parity_loops.py --out $O/r03_parity.json 2>&1 | grep -v "^PARITY_JSON" | tail -40
