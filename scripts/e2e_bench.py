"""End-to-end text -> image timing the way the reference's deploy README measures it (ppdiffusers/deploy/sd15/
infer_dygraph_paddle.py:239-265): it/s = inference_steps / mean wall time of one whole pipeline call (text encoders, CFG
denoising loop, VAE decode), 512x512, batch 1, 50 steps, after a warm-up call. Random-init weights of the real
architectures (no checkpoints offline), synthetic token ids. ORIENTATION ONLY next to BASELINE.md section 1 (other
hardware, other metric than bench.py's); never used as `vs_baseline`.

  python scripts/e2e_bench.py [--model sd15|sdxl] [--calls 5] [--steps 50] [--side 512]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paddlemix_amd.clip import CLIPTextModel, CLIPTextModelWithProjection, synth_clip_params  # noqa: E402
from paddlemix_amd.pipeline import StableDiffusionDenoiser  # noqa: E402
from paddlemix_amd.schedulers import DDIMScheduler, EulerDiscreteScheduler  # noqa: E402
from paddlemix_amd.unet import UNet2DConditionModel, synth_unet_params  # noqa: E402
from paddlemix_amd.vae import AutoencoderKL, synth_decoder_params  # noqa: E402
from tests.configs import CLIP_BIGG, CLIP_L, SD15, SD_VAE, SDXL  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="sd15", choices=["sd15", "sdxl"])
    ap.add_argument("--calls", type=int, default=5)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--side", type=int, default=512)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    xl = a.model == "sdxl"
    ucfg = SDXL if xl else SD15
    unet = UNet2DConditionModel(ucfg, synth_unet_params(ucfg, seed=1, device=dev), device=dev)
    vcfg = dict(SD_VAE, scaling_factor=0.13025) if xl else SD_VAE
    vae = AutoencoderKL(vcfg, synth_decoder_params(vcfg, seed=2, device=dev), device=dev)
    te = CLIPTextModel(CLIP_L, synth_clip_params(CLIP_L, seed=3, device=dev), device=dev)
    te2 = None
    if xl:
        c2 = dict(CLIP_BIGG, with_projection=True)
        te2 = CLIPTextModelWithProjection(c2, synth_clip_params(c2, seed=4, device=dev), device=dev)
        sched = EulerDiscreteScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                       timestep_spacing="leading", steps_offset=1)
    else:
        sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                              set_alpha_to_one=False, steps_offset=1)
    pipe = StableDiffusionDenoiser(unet, sched, vae=vae, text_encoder=te, text_encoder_2=te2)
    g = torch.Generator(device=dev).manual_seed(0)
    ids = torch.randint(3, 40000, (1, 77), generator=g, device=dev)
    ids[:, 0], ids[:, 20:] = 49406, 49407
    neg = torch.full((1, 77), 49407, device=dev)
    neg[:, 0] = 49406
    kw = dict(prompt_ids=ids, negative_prompt_ids=neg, height=a.side, width=a.side, num_inference_steps=a.steps,
              guidance_scale=7.5, output_type="pt", generator=g)
    if xl:
        kw.update(prompt_ids_2=ids, negative_prompt_ids_2=neg, guidance_scale=5.0)
    img = pipe(**kw)   # warm-up call (plans, graphs)
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.calls):
        t0 = time.perf_counter()
        img = pipe(**kw)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    img = img[0] if isinstance(img, (tuple, list)) else getattr(img, "images", img)
    mean = sum(ts) / len(ts)
    print(json.dumps({"what": f"{a.model} text2img end to end, {a.side}x{a.side}, bs 1, {a.steps} steps, CFG, "
                              f"{'2 CLIP' if xl else 'CLIP'} + UNet + VAE decode, random-init weights",
                      "it_per_s": a.steps / mean, "s_per_image": mean, "calls": a.calls,
                      "image_shape": list(img.shape), "finite": bool(torch.isfinite(img.float()).all()),
                      "orientation": "reference deploy README: SD15 47.22 / SDXL 31.98 it/s on A100-80G TensorRT fp16 "
                                     "(ppdiffusers/deploy/README.md:44,47); not the same hardware or weights"}))


if __name__ == "__main__":
    main()
