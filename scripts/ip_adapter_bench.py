"""IP-Adapter cost on one MI355X: OpenCLIP ViT-H/14 image encoder (once per image prompt) and the UNet step with and without the
image-token attention (SDXL geometry, bs 8, 1024^2, random-init weights, inputs resident in HBM).
  python scripts/ip_adapter_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlemix_amd.clip import CLIPVisionModelWithProjection, synth_clip_vision_params  # noqa: E402
from paddlemix_amd.unet import UNet2DConditionModel, synth_unet_params  # noqa: E402
from tests.configs import CLIP_VIT_H14, SDXL  # noqa: E402


def timeit(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    enc = CLIPVisionModelWithProjection(CLIP_VIT_H14, synth_clip_vision_params(CLIP_VIT_H14, 1, device=dev))
    for B in (1, 8):
        px = torch.randn(B, 3, 224, 224, device=dev, generator=g)
        print(f"CLIP ViT-H/14 image encoder, bs {B}: {timeit(lambda: enc(px).image_embeds, 10):.2f} ms")
    del enc
    torch.cuda.empty_cache()
    cfg = dict(SDXL, encoder_hid_dim_type="ip_image_proj", encoder_hid_dim=1024)
    unet = UNet2DConditionModel(cfg, synth_unet_params(cfg, 2, device=dev))
    B = 8
    x = torch.randn(B, 4, 128, 128, device=dev, generator=g)
    ctx = torch.randn(B, 77, 2048, device=dev, generator=g)
    added = dict(text_embeds=torch.randn(B, 1280, device=dev, generator=g),
                 time_ids=torch.tensor([[1024., 1024., 0., 0., 1024., 1024.]], device=dev).repeat(B, 1),
                 image_embeds=torch.randn(B, 1024, device=dev, generator=g))
    for sc in (0.0, 1.0):
        unet.set_ip_adapter_scale(sc)
        ms = timeit(lambda: unet(x, 500, ctx, added_cond_kwargs=added, return_dict=False)[0], 10)
        print(f"SDXL UNet step bs 8 1024^2, IP-Adapter scale {sc}: {ms:.2f} ms")


if __name__ == "__main__":
    main()
