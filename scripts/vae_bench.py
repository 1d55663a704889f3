"""AutoencoderKL decode timing on one MI355X: python scripts/vae_bench.py [--side 128] [--batch 8]
Prints one JSON line: images/s, ms per batch, algorithmic TFLOP/s and the per-kernel-class breakdown."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from paddlemix_amd.vae import AutoencoderKL, synth_decoder_params  # noqa: E402

SD_VAE = dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, scaling_factor=0.13025)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", type=int, default=128, help="latent side (128 -> 1024x1024 images)")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    P = synth_decoder_params(SD_VAE, 1234, device=dev)
    z = torch.randn(a.batch, 4, a.side, a.side, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    prof = AutoencoderKL(SD_VAE, P, device=dev, profile=True)
    prof.decode(z, in_scale=1 / 0.13025)
    prof.kernel_times.clear()
    prof.decode(z, in_scale=1 / 0.13025)
    brk, flops = {}, 0.0
    for kind, lst in prof.kernel_times.items():
        k = kind.split(":")[0]
        brk[k] = brk.get(k, 0.0) + 1e3 * sum(t for t, _ in lst)
        flops += sum(f for _, f in lst)
    if os.environ.get("BENCH_SHAPES"):
        for kind, lst in sorted(prof.kernel_times.items(), key=lambda kv: -sum(t for t, _ in kv[1]))[:30]:
            t = sum(x for x, _ in lst)
            f = sum(x for _, x in lst)
            print(f"  {kind:40s} n={len(lst):4d} {1e3 * t:9.3f} ms {f / t / 1e12 if t else 0:8.1f} TFLOP/s", file=sys.stderr)
    del prof
    torch.cuda.empty_cache()
    vae = AutoencoderKL(SD_VAE, P, device=dev)
    for _ in range(2):
        vae.decode(z, in_scale=1 / 0.13025)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(a.iters):
        out = vae.decode(z, in_scale=1 / 0.13025).sample
    torch.cuda.synchronize()
    dt = (time.time() - t0) / a.iters
    print(json.dumps({"metric": "AutoencoderKL decode images/sec", "value": a.batch / dt, "unit": "images/s",
                      "ms_per_batch": 1e3 * dt, "batch": a.batch, "image": [3, 8 * a.side, 8 * a.side],
                      "algorithmic_tflop_per_batch": flops / 1e12, "tflops_effective": flops / dt / 1e12,
                      "kernel_breakdown_ms": {k: round(v, 3) for k, v in sorted(brk.items(), key=lambda kv: -kv[1])},
                      "finite": bool(torch.isfinite(out).all()), "hbm_peak_GB": torch.cuda.max_memory_allocated() / 1e9}))


if __name__ == "__main__":
    main()
