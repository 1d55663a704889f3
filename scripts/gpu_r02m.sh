#!/bin/bash
# round-2 GPU session M: final defaults (LayerNorm 2 rows per wave / one trip, GroupNorm apply 8 pixels per thread): norm kernel
# tests, the bench line and its variants, rocprofv3 kernel stats
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
T=r02_m
cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sd3.py tests/test_gpu_t5.py tests/test_gpu_dit.py -m gpu -q -k "not gemm and not linear and not conv3x3 and not attention and not sdpa" 2>&1 | tail -3 > $O/${T}_pytest_norms.log
timeout 120 python -m pytest tests/test_gpu_unet.py tests/test_gpu_cexec.py -m gpu -q -x -k "not euler30 and not headline and not from_pretrained" 2>&1 | tail -3 >> $O/${T}_pytest_norms.log
cat $O/${T}_pytest_norms.log
timeout 100 python bench.py --no-cpu-baseline > $O/${T}_bench.json 2> $O/${T}_bench.err
timeout 100 python bench.py --residual fp32 --no-cpu-baseline > $O/${T}_bench_resid_fp32.json 2>/dev/null
timeout 100 python bench.py --dtype fp16 --no-cpu-baseline > $O/${T}_bench_fp16.json 2>/dev/null
timeout 100 python bench.py --dtype fp16 --residual fp32 --no-cpu-baseline > $O/${T}_bench_fp16_resid_fp32.json 2>/dev/null
timeout 100 python bench.py --workload sd15-512-bs1 --steps 50 --warmup 5 --no-cpu-baseline > $O/${T}_bench_sd15_bs1.json 2>/dev/null
timeout 100 python bench.py --workload sd3-1024-bs8 --no-cpu-baseline > $O/${T}_bench_sd3_bs8.json 2>/dev/null
timeout 100 python bench.py --workload sd3-1024-bs8-w8a8 --no-cpu-baseline > $O/${T}_bench_sd3_bs8_w8a8.json 2>/dev/null
timeout 100 python scripts/vae_bench.py --side 128 --batch 8 > $O/${T}_vae_decode_1024_bs8.json 2>/dev/null
BENCH_SHAPES=1 timeout 100 python bench.py --no-cpu-baseline 2> $O/${T}_per_shape_ms.txt > /dev/null
cd /tmp
rm -rf /tmp/pfin
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pfin -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline > /tmp/pfin.log 2>&1
DB=$(find /tmp/pfin -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $DB $O/${T}_sdxl_bs8_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-roofline   ($(tail -1 /tmp/pfin.log | cut -c1-160))" > /dev/null
python - <<'PY'
import json,os,glob
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/"
for f in sorted(glob.glob(O+"r02_m_bench*.json")):
    try:
        d=json.load(open(f)); print(os.path.basename(f), round(d["value"],3), round(d["ms_per_step"],3), d.get("kernel_breakdown_ms"))
    except Exception as e: print(f, "ERR", e)
PY
head -20 $O/${T}_sdxl_bs8_kernel_stats.txt | cut -c1-150
