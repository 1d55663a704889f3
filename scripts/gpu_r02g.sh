#!/bin/bash
# round-2 GPU session G: the round-end measurement set (bench lines, rocprofv3 kernel stats, PMC traffic, attention PMC,
# residual-stream / fp16 / text-encoder bench variants, the full-SDXL 30-step parity loop)
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 bash scripts/final_profile.sh r02_g > $O/r02_g_final_profile.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 150 python bench.py --residual fp32 --no-cpu-baseline > $O/r02_g_bench_resid_fp32.json 2>/dev/null
timeout 150 python bench.py --dtype fp16 --residual fp32 --no-cpu-baseline > $O/r02_g_bench_fp16_resid_fp32.json 2>/dev/null
timeout 200 python bench.py --text-encoders --no-cpu-baseline --no-roofline > $O/r02_g_bench_text_encoders.json 2> $O/r02_g_bench_text_encoders.err
ATTN_PROBE_LOG2=1 timeout 300 bash scripts/pmc.sh scripts/attn_probe.py attention_kernel "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM" > $O/r02_g_attention_pmc.txt 2>&1
ATTN_PROBE_LOG2=1 timeout 120 bash scripts/prof_attn.sh >> $O/r02_g_attention_pmc.txt 2>&1
PARITY_ONLY_SDXL_LOOP=1 timeout 600 python scripts/parity_report.py --out $O/r02_g_parity_sdxl_loop.json > $O/r02_g_parity_sdxl_loop.log 2>&1
tail -c 1500 $O/r02_g_bench.json; tail -5 $O/r02_g_parity_sdxl_loop.log | cut -c1-500
