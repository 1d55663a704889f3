"""Experiment: the bs-8 SDXL step as TWO independent half-batch chains (4 prompts each) replayed concurrently on two streams, against
the single bs-8 chain. Prompts never interact, so the split is exact. Why it might pay: in one chain every kernel boundary is a
chip-wide bubble (all 256 CUs leave their K loops together, hammer HBM with their epilogues together, wait out the launch gap and
the next prologue together: ~17 of a K = 1280 GEMM's 38 us, scripts/gemm_timeline.py); two desynchronised chains put one chain's
memory phases beside the other's MFMA phases (and a power-limited chip clocks the busy half higher).

    python scripts/two_stream_probe.py [--steps 20] [--offset-us 0]

Prints ms per bs-8 step for: one chain of 8; two chains of 4 back to back on ONE stream (what the split alone costs); two chains of
4 on two streams."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from paddlemix_amd import _lib  # noqa: E402
from paddlemix_amd.dist import wire_params  # noqa: E402
from paddlemix_amd.unet import UNet2DConditionModel, synth_unet_params  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
cfg = bench.SDXL
H = W = 128
L = 77
P = wire_params(synth_unet_params(cfg, seed=1234, device=dev), _lib.elem_dtype())


def inputs(B, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    lat = torch.randn(B, 4, H, W, generator=g, device=dev)
    enc = torch.randn(B, L, cfg["cross_attention_dim"], generator=g, device=dev)
    td = cfg["projection_class_embeddings_input_dim"] - 6 * cfg["addition_time_embed_dim"]
    added = dict(text_embeds=torch.randn(B, td, generator=g, device=dev),
                 time_ids=torch.tensor([[1024., 1024., 0., 0., 1024., 1024.]], device=dev).repeat(B, 1))
    return lat, enc, added


def prepare(model, B, seed):
    lat, enc, added = inputs(B, seed)
    plan = model._get_plan(B, H, W, L)
    with torch.cuda.stream(model._stream):
        model.stage_inputs(plan, lat, 500.0, enc, added, in_scale=0.5)
        model.run(plan)      # eager warm-up + capture + first replay
        model.run(plan)
    torch.cuda.synchronize()
    return plan


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


full = UNet2DConditionModel(cfg, P, device=dev)
p8 = prepare(full, 8, 0)
ms8 = timed(lambda: full.run(p8), args.steps)
print(f"one chain of 8           : {ms8:7.3f} ms per bs-8 step", flush=True)

a = UNet2DConditionModel(cfg, P, device=dev)
b = UNet2DConditionModel(cfg, P, device=dev)
pa, pb = prepare(a, 4, 1), prepare(b, 4, 2)


def serial():
    a.run(pa)
    b._stream.wait_stream(a._stream)
    b.run(pb)
    a._stream.wait_stream(b._stream)


def concurrent():
    a.run(pa)
    b.run(pb)


ms_ser = timed(serial, args.steps)
print(f"two chains of 4, serial  : {ms_ser:7.3f} ms per bs-8 step", flush=True)
for rep in range(2):
    ms_con = timed(concurrent, args.steps)
    print(f"two chains of 4, 2 streams: {ms_con:7.3f} ms per bs-8 step  ({100 * (ms8 / ms_con - 1):+.1f} % steps/s vs one chain)", flush=True)
ms8b = timed(lambda: full.run(p8), args.steps)
print(f"one chain of 8 (again)   : {ms8b:7.3f} ms per bs-8 step", flush=True)
