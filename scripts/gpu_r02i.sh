#!/bin/bash
# round-2 GPU session I: HBM-bound leftovers (GroupNorm apply / statistics without per-chunk divisions and with four loads in
# flight, conv_in four pixels per thread, conv_out on packed dot products) -- kernel tests, UNet tests, bench line
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sd3.py -m gpu -q -k "not gemm and not linear and not conv3x3 and not attention and not sdpa" > $O/r02_i_pytest_kernels.log 2>&1
tail -4 $O/r02_i_pytest_kernels.log
timeout 300 python -m pytest tests/test_gpu_unet.py tests/test_gpu_cexec.py tests/test_gpu_vae.py -m gpu -q -x -k "not euler30 and not headline and not from_pretrained" > $O/r02_i_pytest_unet.log 2>&1
tail -4 $O/r02_i_pytest_unet.log
timeout 200 python bench.py --no-cpu-baseline > $O/r02_i_bench.json 2> $O/r02_i_bench.err
timeout 100 python bench.py --workload sd15-512-bs1 --steps 50 --warmup 5 --no-cpu-baseline > $O/r02_i_bench_sd15_bs1.json 2>/dev/null
cd /tmp
rm -rf /tmp/pfin
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pfin -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline > /tmp/pfin.log 2>&1
DB=$(find /tmp/pfin -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $DB $O/r02_i_sdxl_bs8_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-roofline   ($(tail -1 /tmp/pfin.log | cut -c1-160))" > /dev/null
python - <<'PY'
import json,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/"
for f in ("r02_i_bench.json","r02_i_bench_sd15_bs1.json"):
    try:
        d=json.load(open(O+f)); print(f, d["value"], d["ms_per_step"], d.get("kernel_breakdown_ms"))
    except Exception as e: print(f, "ERR", e)
PY
grep -E "scale_shift|gn_partial|gn_final|conv_in|conv_out|layernorm" $O/r02_i_sdxl_bs8_kernel_stats.txt | cut -c1-150
