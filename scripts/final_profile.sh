#!/bin/bash
# Round-end measurement set: bench lines, rocprofv3 kernel-trace summary of the default bench, PMC traffic.
# usage (on the GPU box): bash scripts/final_profile.sh <tag>      -> gpurun_out/<tag>_*
TAG=${1:-r01_f}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python bench.py --workload sd15-512-bs1 --steps 50 --warmup 5 > $OUT/${TAG}_bench_sd15-512-bs1.json 2>/dev/null
for w in sd3-1024-bs8 sd3-1024-bs8-fp8w sd3-1024-bs8-w8a8; do   # (each line: cpu_baseline + pred_rel_bs8 at config 5's own geometry)
  python bench.py --workload $w > $OUT/${TAG}_bench_$w.json 2>/dev/null
done
python bench.py --dtype fp16 --no-cpu-baseline > $OUT/${TAG}_bench_fp16.json 2>/dev/null
python scripts/vae_bench.py --side 128 --batch 8 > $OUT/${TAG}_vae_decode_1024_bs8.json 2>/dev/null
BENCH_SHAPES=1 python bench.py --no-cpu-baseline 2> $OUT/${TAG}_per_shape_ms.txt > /dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pfin
rocprofv3 --kernel-trace --stats -d /tmp/pfin -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline > /tmp/pfin.log 2>&1
DB=$(find /tmp/pfin -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $DB $OUT/${TAG}_sdxl_bs8_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-roofline   ($(tail -1 /tmp/pfin.log | cut -c1-160))" > /dev/null
timeout 420 bash $GRAFT_REPO_ROOT/scripts/traffic.sh sdxl-1024-bs8 180 > $OUT/${TAG}_traffic_sdxl-1024-bs8.json 2>/dev/null
tail -c 600 $OUT/${TAG}_bench.json; echo; head -12 $OUT/${TAG}_sdxl_bs8_kernel_stats.txt; head -c 400 $OUT/${TAG}_traffic_sdxl-1024-bs8.json
