#!/bin/bash
# Round 5, final measurement set on the shipped tree: the GPU tests the -x run had not reached, the bench lines, the rocprofv3
# kernel table of the bench command, per-shape table, fabric traffic, the bs-8 forward parity in all four device modes.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
T=r05_f
timeout 900 python -m pytest tests/test_gpu_switches.py tests/test_gpu_t5.py tests/test_gpu_unet.py tests/test_gpu_vae.py -q -m gpu > $O/${T}_pytest_rest.txt 2>&1
python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err
BENCH_SHAPES=1 python bench.py --no-cpu-baseline --no-parity-mode 2> $O/${T}_per_shape_ms.txt > /dev/null
python bench.py --workload sd15-512-bs1 --steps 50 --warmup 5 --no-cpu-baseline > $O/${T}_bench_sd15-512-bs1.json 2>/dev/null
python bench.py --workload sd3-1024-bs8 --no-cpu-baseline > $O/${T}_bench_sd3-1024-bs8.json 2>/dev/null
python bench.py --workload sd3-1024-bs8-fp8w --no-cpu-baseline > $O/${T}_bench_sd3-1024-bs8-fp8w.json 2>/dev/null
python bench.py --workload sd3-1024-bs8-w8a8 --no-cpu-baseline > $O/${T}_bench_sd3-1024-bs8-w8a8.json 2>/dev/null
( cd /tmp; rm -rf /tmp/pfin
  rocprofv3 --kernel-trace --stats -d /tmp/pfin -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-parity-mode > /tmp/pfin.log 2>&1
  DB=$(find /tmp/pfin -name "*.db" | head -1)
  python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $DB $O/${T}_sdxl_bs8_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-roofline --no-parity-mode   ($(tail -1 /tmp/pfin.log | cut -c1-160))" > /dev/null )
timeout 420 bash scripts/traffic.sh sdxl-1024-bs8 180 > $O/${T}_traffic_sdxl-1024-bs8.json 2>/dev/null
timeout 900 python scripts/parity_loops.py --cases sdxl_8x4x128x128_fwd --out $O/r05_parity_bs8.json > $O/r05_parity_bs8.log 2>&1
tail -4 $O/${T}_pytest_rest.txt
python - <<PY
import json
d = json.load(open("$O/${T}_bench.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "meets_target", "value_meeting_target")})
print("roofline", d.get("roofline"))
print("parity", {k: d["parity"].get(k) for k in ("end_latents_rel_l2", "pred_rel_bs8", "pred_rel_bs8_per_prompt_max", "meets_target", "error")})
print("parity_mode", {k: d["parity_mode"].get(k) for k in ("steps_per_s", "end_latents_rel_l2", "pred_rel_bs8", "meets_target", "error")})
print("cpu_baseline", {k: d["cpu_baseline"].get(k) for k in ("value", "unit", "cores", "kind")})
PY
head -14 $O/${T}_sdxl_bs8_kernel_stats.txt
head -c 500 $O/${T}_traffic_sdxl-1024-bs8.json; echo
tail -3 $O/r05_parity_bs8.log | cut -c1-600
for w in sd15-512-bs1 sd3-1024-bs8 sd3-1024-bs8-fp8w sd3-1024-bs8-w8a8; do python -c "import json;d=json.load(open('$O/${T}_bench_$w.json'));print('$w', d['value'], d['ms_per_step'])"; done
