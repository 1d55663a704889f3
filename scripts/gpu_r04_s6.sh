#!/bin/bash
# Round 4, session 6 (final code): smoke(), the whole -m gpu suite in ONE process the way the driver runs it, the bench line, the
# HBM-traffic passes, matrix-pipe counters of the FF1 launch, the fp16 line, the SD3 16-bit / weight-only-fp8 pair on one box.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
T=r04_f
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -3 > $O/${T}_smoke.txt; cat $O/${T}_smoke.txt
( time timeout 2700 python -X faulthandler -m pytest tests -m gpu -x -q 2>&1 | grep -v "amdgpu.ids" > /tmp/suite.log ) 2> /tmp/suite.time
( head -60 /tmp/suite.log | cut -c1-220; echo ...; tail -60 /tmp/suite.log | cut -c1-400; cat /tmp/suite.time ) > $O/${T}_pytest_gpu.log
tail -6 $O/${T}_pytest_gpu.log
BENCH_SHAPES=1 timeout 900 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err; grep -E "^  (gemm|conv|attn):" $O/${T}_bench.err > $O/${T}_per_shape_ms.txt; tail -c 1800 $O/${T}_bench.json
timeout 600 bash scripts/traffic.sh sdxl-1024-bs8 > $O/${T}_traffic_sdxl-1024-bs8.json 2> $O/${T}_traffic.err; grep -A8 '"gemm"' $O/${T}_traffic_sdxl-1024-bs8.json
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
( echo "== gemm 8192x10240x1280 GEGLU (FF1; 256x320 streaming loop, 1024 tiles on 256 persistent blocks)"; GEMM_SHAPE=8192x10240x1280 GEGLU=1 timeout 300 bash scripts/pmc.sh scripts/gemm_one.py gemm_pipe_kernel "$P1" "$P2" ) > $O/${T}_pmc_ff1.txt 2>&1; cat $O/${T}_pmc_ff1.txt
timeout 240 python bench.py --dtype fp16 --no-cpu-baseline --no-parity-mode > $O/${T}_bench_fp16.json 2>/dev/null
for wl in sd3-1024-bs8 sd3-1024-bs8-fp8w; do timeout 240 python bench.py --workload $wl --no-cpu-baseline --no-parity-mode --steps 20 > $O/${T}_bench_$wl.json 2>/dev/null; done
python - $O/${T}_bench_fp16.json $O/${T}_bench_sd3-1024-bs8.json $O/${T}_bench_sd3-1024-bs8-fp8w.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.load(open(f)); print(d["config"]["workload"], d["dtype"], round(d["value"],3), "steps/s", round(d["ms_per_step"],3), "ms", d.get("kernel_breakdown_ms"))
    except Exception as e: print(f, "ERR", e)
PY
