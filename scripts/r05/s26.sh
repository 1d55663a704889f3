#!/bin/bash
# Round 5, session 26: the multi-rank code path of bench.py on the one GPU of the box (RCCL group of one: process group, weight
# broadcast, barriers, all-gather of the latents); stdout must be the JSON line alone (RCCL's banner goes to stderr)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 python bench.py --gpus 1 --force-dist --no-cpu-baseline --no-parity-mode > $O/r05_s26_bench_force_dist.json 2> $O/r05_s26_bench_force_dist.err
echo "stdout lines: $(wc -l < $O/r05_s26_bench_force_dist.json)"
grep -c "RCCL version" $O/r05_s26_bench_force_dist.err
python -c "
import json; d=json.load(open('$O/r05_s26_bench_force_dist.json'))
print({k: d.get(k) for k in ('value','ms_per_step','n_gpus','scaling','weight_broadcast_s','weight_broadcast_gb','gathered_latents','per_rank_ms_per_step')})"
