#!/bin/bash
# One process per GPU of one node: prompt-sharded SDXL denoising (bench.py), RCCL broadcast of UNet + text-encoder weights at start,
# no per-step collective, one all-gather of the latents at the end.   usage: scripts/launch_8gpu.sh [N=8] [bench.py flags...]
set -e
N=${1:-8}; shift || true
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0      # dmabuf IPC: what RCCL needs on this driver
export MASTER_ADDR=127.0.0.1
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "${MASTER_PORT:-29500}" \
     bench.py --gpus "$N" "$@"
