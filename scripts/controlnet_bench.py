"""ControlNet cost on one MI355X (random-init weights, inputs resident in HBM): one ControlNetModel forward and one UNet forward
that takes its residuals, SD-1.5 geometry 512^2 at the CFG batch of one prompt (bs 2) and at bs 16.
  python scripts/controlnet_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlemix_amd.unet import ControlNetModel, UNet2DConditionModel, synth_controlnet_params, synth_unet_params  # noqa: E402
from tests.configs import SD15  # noqa: E402


def timeit(fn, n=10):
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
    net = ControlNetModel(SD15, synth_controlnet_params(SD15, 1, device=dev))
    unet = UNet2DConditionModel(SD15, synth_unet_params(SD15, 2, device=dev))
    for B in (2, 16):
        x = torch.randn(B, 4, 64, 64, device=dev, generator=g)
        ctx = torch.randn(B, 77, 768, device=dev, generator=g)
        hint = torch.rand(B, 3, 512, 512, device=dev, generator=g)
        d, m = net(x, 500, ctx, hint, return_dict=False)
        t_net = timeit(lambda: net(x, 500, ctx, hint, return_dict=False))
        t_plain = timeit(lambda: unet(x, 500, ctx, return_dict=False))
        t_ctrl = timeit(lambda: unet(x, 500, ctx, down_block_additional_residuals=d, mid_block_additional_residual=m, return_dict=False))
        print(f"SD-1.5 512^2 bs {B}: ControlNet forward {t_net:.2f} ms | UNet {t_plain:.2f} ms | UNet with residuals {t_ctrl:.2f} ms "
              f"| controlled step {t_net + t_ctrl:.2f} ms")


if __name__ == "__main__":
    main()
