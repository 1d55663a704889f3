"""AutoencoderKL encode / decode wall time on one GPU (SD VAE geometry, random-init weights, inputs resident in HBM).
Usage: python scripts/vae_encode_bench.py [size ...]   (image side in pixels, default 512 1024)"""
import sys
import time

import torch

sys.path.insert(0, ".")
from paddlemix_amd.vae import AutoencoderKL, synth_vae_params  # noqa: E402

SD_VAE = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
              layers_per_block=2, norm_num_groups=32, scaling_factor=0.13025)


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [512, 1024]
    vae = AutoencoderKL(SD_VAE, synth_vae_params(SD_VAE, 1, device="cuda"))
    for px in sizes:
        x = torch.rand(1, 3, px, px, device="cuda") * 2 - 1
        for name, fn, arg in (("encode", lambda t: vae.encode(t).latent_dist.mean, x),
                              ("decode", lambda t: vae.decode(t).sample, torch.randn(1, 4, px // 8, px // 8, device="cuda"))):
            for _ in range(2):
                fn(arg)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 5
            for _ in range(n):
                fn(arg)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
            print(f"vae {name} {px}x{px} bs1: {ms:.2f} ms  ({1e3 / ms:.1f} img/s)")


if __name__ == "__main__":
    main()
