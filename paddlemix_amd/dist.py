"""Multi-GPU plumbing of the denoising path: one process per GPU, prompts sharded across ranks, RCCL only at the edges.

The reference's only inference-time collective is the SD3 batch-parallel CFG split
(ppdiffusers/pipelines/stable_diffusion_3/pipeline_stable_diffusion_3.py:803-839); whole prompts never interact inside the
denoising loop (GroupNorm / LayerNorm / attention are per sample -- tests/test_gpu_unet.py::test_batch_independence), so
the loop itself needs NO collective. What is exchanged:

  * once, at start-up: rank 0's weights (UNet + text encoders), broadcast over xGMI. On the wire a matrix is the 16-bit
    element type the kernels consume (bf16: 5.1 GB for the SDXL UNet instead of 10.3 GB of fp32), 1-D parameters stay fp32.
    xGMI is point-to-point, so a ring broadcast is bound by one link (~150 GB/s): tensors of >= 8 MB go out in place, one
    collective each and no staging copy; the hundreds of small ones (biases, norm parameters) are packed into a few flat
    buckets so they do not cost a launch each;
  * once, at the end: the ranks' latents, all-gathered to every rank (a few MB).
"""
from __future__ import annotations

from typing import Dict, Mapping

import torch

BIG = 4 << 20          # elements: tensors at least this large are broadcast in place
BUCKET_BYTES = 64 << 20


def wire_params(P: Mapping[str, torch.Tensor], wire_dtype: torch.dtype) -> Dict[str, torch.Tensor]:
    """the form the weights travel (and are kept) in: matrices / conv kernels in the kernels' 16-bit element type, the rest fp32"""
    return {k: (v.to(wire_dtype) if v.dim() > 1 else v.float()) for k, v in P.items()}


def empty_wire_params(shapes: Mapping[str, tuple], wire_dtype: torch.dtype, device) -> Dict[str, torch.Tensor]:
    return {k: torch.empty(s, device=device, dtype=wire_dtype if len(s) > 1 else torch.float32) for k, s in shapes.items()}


def broadcast_params(P: Dict[str, torch.Tensor], src: int = 0) -> int:
    """Broadcast every tensor of P from `src` (in place on the receivers). Returns the bytes put on the wire."""
    import torch.distributed as dist
    total = 0
    small: Dict[torch.dtype, list] = {}
    for n in P:     # dict order is the construction order on every rank
        t = P[n]
        total += t.numel() * t.element_size()
        if t.numel() >= BIG:   # (decided by SIZE only: every rank must issue the same sequence of collectives)
            if t.is_contiguous():
                dist.broadcast(t, src=src)
            else:
                c = t.contiguous()
                dist.broadcast(c, src=src)
                t.copy_(c)
        else:
            small.setdefault(t.dtype, []).append(n)

    for dt, names in small.items():
        bucket, size = [], 0

        def flush():
            nonlocal bucket, size
            if not bucket:
                return
            flat = torch.cat([P[n].reshape(-1) for n in bucket])
            dist.broadcast(flat, src=src)
            off = 0
            for n in bucket:
                k = P[n].numel()
                P[n].copy_(flat[off:off + k].view_as(P[n]))
                off += k
            bucket, size = [], 0

        for n in names:
            bucket.append(n)
            size += P[n].numel() * P[n].element_size()
            if size >= BUCKET_BYTES:
                flush()
        flush()
    return total


def gather_latents(latents: torch.Tensor) -> torch.Tensor:
    """[B, C, H, W] on every rank -> [world * B, C, H, W] on every rank (rank-major: global prompt index = rank * B + b)"""
    import torch.distributed as dist
    world = dist.get_world_size()
    out = torch.empty((world * latents.shape[0],) + tuple(latents.shape[1:]), device=latents.device, dtype=latents.dtype)
    dist.all_gather_into_tensor(out, latents.contiguous())
    return out
