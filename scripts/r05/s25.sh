#!/bin/bash
# Round 5, session 25: widening pass with four loads in flight per lane -- switch test (widened vs in-fragment conversion), the fp8w
# launches of the tile-family test, SD3 bs 8 with 16-bit / e4m3 weights interleaved twice, kernel table of the e4m3 run
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_switches.py tests/test_gpu_sd3.py -q -m gpu > $O/r05_s25_pytest.txt 2>&1
tail -3 $O/r05_s25_pytest.txt | cut -c1-300
one() { python bench.py --workload $1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], 'steps/s', d['ms_per_step'], 'ms')"; }
{ for r in 1 2; do for w in sd3-1024-bs8 sd3-1024-bs8-fp8w; do one $w; done; done; } > $O/r05_s25_sd3_modes.txt 2>&1
cat $O/r05_s25_sd3_modes.txt
cd /tmp; rm -rf /tmp/p25
rocprofv3 --kernel-trace --stats -d /tmp/p25 -o r -- python $GRAFT_REPO_ROOT/bench.py --workload sd3-1024-bs8-fp8w --no-cpu-baseline --no-roofline > /tmp/p25.log 2>&1
DB=$(find /tmp/p25 -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $DB $O/r05_s25_sd3_fp8w_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --workload sd3-1024-bs8-fp8w --no-cpu-baseline --no-roofline" > /dev/null
grep "widen\|calls" $O/r05_s25_sd3_fp8w_kernel_stats.txt | cut -c1-200
