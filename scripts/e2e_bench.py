"""End-to-end text -> image timing the way the reference's deploy README measures it (ppdiffusers/deploy/sd15/
infer_dygraph_paddle.py:239-265): it/s = inference_steps / mean wall time of one whole pipeline call (text encoders, CFG
denoising loop, VAE decode), 512x512, batch 1, 50 steps, after a warm-up call. Random-init weights of the real
architectures (no checkpoints offline), synthetic token ids. ORIENTATION ONLY next to BASELINE.md section 1 (other
hardware, other metric than bench.py's); never used as `vs_baseline`.

  python scripts/e2e_bench.py [--model sd15|sdxl|sd3|dit|lcm] [--img2img] [--calls 5] [--steps 50] [--side 512]
(lcm: the SD-1.5 UNet with the guidance embedding of LCM-distilled checkpoints, LCMScheduler, 4 steps, no doubled batch;
--img2img: VAE encode -> re-noise -> the last 75 % of the steps, pipeline_stable_diffusion_img2img.py)
(sd3: the reference quotes seconds per image, ppdiffusers/deploy/sd3/README.md:27-31: 1.2 s Paddle-Inference on A100-40G)
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
from paddlemix_amd.schedulers import DDIMScheduler, EulerDiscreteScheduler, LCMScheduler  # noqa: E402
from paddlemix_amd.unet import UNet2DConditionModel, synth_unet_params  # noqa: E402
from paddlemix_amd.vae import AutoencoderKL, synth_decoder_params, synth_vae_params  # noqa: E402
from tests.configs import CLIP_BIGG, CLIP_L, SD15, SD_VAE, SDXL  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="sd15", choices=["sd15", "sdxl", "sd3", "dit", "lcm"])
    ap.add_argument("--controlnet", action="store_true", help="sd15: ControlNet residuals every step (512^2 hint image)")
    ap.add_argument("--img2img", action="store_true", help="sd15 / sdxl / lcm: start from an encoded image, strength 0.75")
    ap.add_argument("--act-dtype", default="bf16", choices=["bf16", "fp8"], help="sd3 only: W8A8 block GEMMs")
    ap.add_argument("--calls", type=int, default=5)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--side", type=int, default=512)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if a.model == "sd3":
        return sd3(a, dev)
    if a.model == "dit":
        return dit(a, dev)
    xl = a.model == "sdxl"
    lcm = a.model == "lcm"
    ucfg = SDXL if xl else (dict(SD15, time_cond_proj_dim=256) if lcm else SD15)
    if lcm and a.steps == 50:
        a.steps = 4
    unet = UNet2DConditionModel(ucfg, synth_unet_params(ucfg, seed=1, device=dev), device=dev)
    vcfg = dict(SD_VAE, scaling_factor=0.13025) if xl else SD_VAE
    vae = AutoencoderKL(vcfg, (synth_vae_params if a.img2img else synth_decoder_params)(vcfg, seed=2, device=dev), device=dev)
    te = CLIPTextModel(CLIP_L, synth_clip_params(CLIP_L, seed=3, device=dev), device=dev)
    te2 = None
    if xl:
        c2 = dict(CLIP_BIGG, with_projection=True)
        te2 = CLIPTextModelWithProjection(c2, synth_clip_params(c2, seed=4, device=dev), device=dev)
        sched = EulerDiscreteScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                       timestep_spacing="leading", steps_offset=1)
    elif lcm:
        sched = LCMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear")
    else:
        sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                              set_alpha_to_one=False, steps_offset=1)
    cn = None
    if a.controlnet:
        from paddlemix_amd.unet import ControlNetModel, synth_controlnet_params
        cn = ControlNetModel(ucfg, synth_controlnet_params(ucfg, seed=9, device=dev), device=dev)
    pipe = StableDiffusionDenoiser(unet, sched, vae=vae, text_encoder=te, text_encoder_2=te2, controlnet=cn)
    g = torch.Generator(device=dev).manual_seed(0)
    ids = torch.randint(3, 40000, (1, 77), generator=g, device=dev)
    ids[:, 0], ids[:, 20:] = 49406, 49407
    neg = torch.full((1, 77), 49407, device=dev)
    neg[:, 0] = 49406
    kw = dict(prompt_ids=ids, negative_prompt_ids=neg, height=a.side, width=a.side, num_inference_steps=a.steps,
              guidance_scale=7.5, output_type="pt", generator=g)
    if xl:
        kw.update(prompt_ids_2=ids, negative_prompt_ids_2=neg, guidance_scale=5.0)
    if lcm:
        kw.update(guidance_scale=8.0)
    if a.img2img:
        kw.update(image=torch.rand(1, 3, a.side, a.side, generator=g, device=dev) * 2 - 1, strength=0.75)
    if a.controlnet:
        kw.update(control_image=torch.rand(1, 3, a.side, a.side, generator=g, device=dev))
    ran = int(a.steps * 0.75) if a.img2img else a.steps
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
    print(json.dumps({"what": f"{a.model} {'img2img (strength 0.75)' if a.img2img else 'text2img'} end to end, {a.side}x{a.side}, bs 1, "
                              f"{ran} UNet steps, {'guidance embedding (no doubled batch)' if lcm else 'CFG'}, "
                              f"{'2 CLIP' if xl else 'CLIP'} + {'VAE encode + ' if a.img2img else ''}{'ControlNet + ' if a.controlnet else ''}UNet + VAE decode, random-init weights",
                      "it_per_s": ran / mean, "s_per_image": mean, "calls": a.calls,
                      "image_shape": list(img.shape), "finite": bool(torch.isfinite(img.float()).all()),
                      "orientation": "reference deploy README: SD15 47.22 / SDXL 31.98 it/s on A100-80G TensorRT fp16 "
                                     "(ppdiffusers/deploy/README.md:44,47); not the same hardware or weights"}))


def dit(a, dev):
    """DiTPipeline.__call__ as the reference benchmarks it (examples/inference/class_conditional_image_generation-dit.py:82-103):
    DiT-XL/2-256, one class label, 25 DDIM steps, guidance 4.0, VAE decode, wall time of the whole call"""
    from paddlemix_amd.dit import DiTTransformer2DModel, synth_dit_params
    from paddlemix_amd.pipeline import DiTDenoiser
    from tests.configs import DIT_XL2
    tr = DiTTransformer2DModel(DIT_XL2, synth_dit_params(DIT_XL2, seed=1, device=dev), device=dev)
    vae = AutoencoderKL(SD_VAE, synth_decoder_params(SD_VAE, seed=2, device=dev), device=dev)
    sched = DDIMScheduler(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", clip_sample=False)
    pipe = DiTDenoiser(tr, sched, vae=vae)
    g = torch.Generator(device=dev).manual_seed(42)
    labels = torch.tensor([207], device=dev)
    steps = 25 if a.steps == 50 else a.steps
    call = lambda: pipe(labels, guidance_scale=4.0, num_inference_steps=steps, generator=g, output_type="pt", device=dev)  # noqa: E731
    for _ in range(3):
        img = call()
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.calls):
        t0 = time.perf_counter()
        img = call()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    mean = sum(ts) / len(ts)
    print(json.dumps({"what": f"DiT-XL/2-256 class-conditional generation end to end, 1 label, {steps} DDIM steps, guidance 4.0, "
                              "DiT + VAE decode, random-init weights", "ms_per_image": 1e3 * mean, "calls": a.calls,
                      "image_shape": list(img.shape), "finite": bool(torch.isfinite(img.float()).all()),
                      "orientation": "reference: 219 ms (Paddle-Inference) / 242 ms (TensorRT-LLM) / 1200 ms (Paddle dygraph) on "
                                     "A100-SXM4-40GB fp16 (ppdiffusers/examples/class_conditional_image_generation/DiT/README.md:419-421); "
                                     "not the same hardware or weights"}))


def sd3(a, dev):
    """StableDiffusion3Pipeline.__call__ (pipeline_stable_diffusion_3.py:772-886): 2 x CLIP + T5-XXL, 28-or-N-step CFG loop on
    the MMDiT, 16-channel VAE decode"""
    from paddlemix_amd.pipeline import StableDiffusion3Denoiser
    from paddlemix_amd.schedulers import FlowMatchEulerDiscreteScheduler
    from paddlemix_amd.sd3 import SD3Transformer2DModel, synth_sd3_params
    from paddlemix_amd.t5 import T5EncoderModel, synth_t5_params
    from tests.configs import SD3_MEDIUM, T5_XXL
    kw = dict(weight_dtype="fp8", act_dtype="fp8") if a.act_dtype == "fp8" else {}
    tr = SD3Transformer2DModel(SD3_MEDIUM, synth_sd3_params(SD3_MEDIUM, seed=1, device=dev), device=dev, **kw)
    vcfg = dict(SD_VAE, latent_channels=16, use_post_quant_conv=False, scaling_factor=1.5305, shift_factor=0.0609)
    vae = AutoencoderKL(vcfg, synth_decoder_params(vcfg, seed=2, device=dev), device=dev)
    c1, c2 = dict(CLIP_L, with_projection=True), dict(CLIP_BIGG, with_projection=True)
    te = CLIPTextModelWithProjection(c1, synth_clip_params(c1, seed=3, device=dev), device=dev)
    te2 = CLIPTextModelWithProjection(c2, synth_clip_params(c2, seed=4, device=dev), device=dev)
    te3 = T5EncoderModel(T5_XXL, synth_t5_params(T5_XXL, seed=5, device=dev), device=dev)
    pipe = StableDiffusion3Denoiser(tr, FlowMatchEulerDiscreteScheduler(shift=3.0), vae=vae, text_encoder=te,
                                    text_encoder_2=te2, text_encoder_3=te3)
    g = torch.Generator(device=dev).manual_seed(0)
    ids = torch.randint(3, 40000, (1, 77), generator=g, device=dev)
    ids[:, 0], ids[:, 20:] = 49406, 49407
    neg = torch.full((1, 77), 49407, device=dev)
    neg[:, 0] = 49406
    t5_ids = torch.randint(2, 32000, (1, 256), generator=g, device=dev)
    t5_neg = torch.zeros((1, 256), dtype=torch.long, device=dev)

    def call():
        pe, pp = pipe.encode_prompt(ids, ids, t5_ids)
        ne, npool = pipe.encode_prompt(neg, neg, t5_neg)
        return pipe(pe, pp, ne, npool, height=a.side, width=a.side, num_inference_steps=a.steps, guidance_scale=7.0,
                    generator=g, output_type="pt")

    img = call()
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.calls):
        t0 = time.perf_counter()
        img = call()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    mean = sum(ts) / len(ts)
    print(json.dumps({"what": f"sd3-medium text2img end to end, {a.side}x{a.side}, bs 1, {a.steps} steps, CFG, 2 CLIP + T5-XXL "
                              f"(256 tokens) + MMDiT ({a.act_dtype} block GEMMs) + VAE decode, random-init weights",
                      "s_per_image": mean, "it_per_s": a.steps / mean, "calls": a.calls, "image_shape": list(img.shape),
                      "finite": bool(torch.isfinite(img.float()).all()),
                      "orientation": "reference: 1.2 s (Paddle-Inference + Triton fused ops) / 1.78 s (PyTorch) on A100-SXM4-40GB, "
                                     "fp16, 512x512, 50 steps (ppdiffusers/deploy/sd3/README.md:27-31); not the same hardware or weights"}))


if __name__ == "__main__":
    main()
