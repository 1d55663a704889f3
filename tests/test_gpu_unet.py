"""-m gpu whole-model parity: the HIP UNet2DConditionModel against the torch-CPU oracle on identical synthetic
weights (bf16-representable) and inputs.  Stated tolerance for bf16 activations / fp32 accumulation: rel-L2 of the
noise prediction <= 2e-2 per forward (the host-memory emulator, which has the same rounding points but fp32 math,
sits at ~1e-2 against the same oracle; see DESIGN.md "Numerics").  The oracle itself is held to the reference's own
UNet2DConditionModel code by tests/test_reference_modules.py; real checkpoints do not exist in this environment (oracle/__init__.py)."""
import pytest
import torch

from oracle import unet_ref as U
from tests.configs import MINI_XL, SD15, SDXL, TINY, UNET_VARIANTS

pytestmark = pytest.mark.gpu

# Whole-UNet bars = 1.5 x the value MEASURED on the MI355X (bf16 build; first taken in round 2, profiles/r02_parity.json; the same
# quantities re-measured in rounds 3 and 4 sit within 5 % of them: profiles/r03_parity.json, r04_parity.json pred_rel_teacher_forced_max):
# rel-L2 of one noise prediction vs the oracle, (16-bit residual stream, fp32 residual stream). north_star asks for 1e-3 on
# LATENTS: that quantity, at full depth and against the absolute bar, is tests/test_gpu_parity_loops.py; the 30-step loop test below
# measures it on the mini configuration (1.9e-3 / 1.2e-3 in bf16). DESIGN.md section 4 has the tables.
MEASURED = {"tiny": (9.04e-3, 6.23e-3), "mini-xl": (1.289e-2, 8.67e-3), "sd15-1x4x64x64": (1.280e-2, 8.10e-3),
            "sdxl-1x4x128x128": (1.523e-2, 9.53e-3), "euler30-eps": (1.347e-2, 8.48e-3), "euler30-latents": (1.915e-3, 1.229e-3)}
BARS = {k: (1.5 * a, 1.5 * b) for k, (a, b) in MEASURED.items()}
BAR_16 = 2e-2        # other whole-model tests of this file (16-bit stream, small configurations)


def _rel(a, b):
    return ((a - b).norm() / b.norm()).item()


def _inputs(cfg, B, H, W, L=77, seed=0):
    g = torch.Generator().manual_seed(seed)
    sample = torch.randn(B, cfg.get("in_channels", 4), H, W, generator=g)
    cross = cfg["cross_attention_dim"]
    enc = torch.randn(B, L, cross[0] if isinstance(cross, (tuple, list)) else cross, generator=g)
    added = None
    if cfg.get("addition_embed_type") == "text_time":
        td = cfg["projection_class_embeddings_input_dim"] - 6 * cfg["addition_time_embed_dim"]
        added = dict(text_embeds=torch.randn(B, td, generator=g),
                     time_ids=torch.tensor([[1024., 1024., 0., 0., 1024., 1024.]]).repeat(B, 1))
    return sample, enc, added


def _cuda(x):
    if x is None:
        return None
    if isinstance(x, dict):
        return {k: v.cuda() for k, v in x.items()}
    return x.cuda()


def _bf16_params(cfg, device):
    from paddlemix_amd.unet import synth_unet_params
    P = synth_unet_params(cfg, seed=1234, device=device)
    for k, v in P.items():
        if v.dim() > 1:
            P[k] = v.to(torch.bfloat16).float()
    return P


@pytest.mark.parametrize("name,cfg,B,H,W,L", [("tiny", TINY, 2, 16, 16, 7), ("tiny-ragged", TINY, 1, 8, 24, 5),
                                              ("mini-xl", MINI_XL, 2, 32, 32, 77)])
def test_small_unet_vs_oracle(name, cfg, B, H, W, L):
    from paddlemix_amd.unet import UNet2DConditionModel
    P = _bf16_params(cfg, "cpu")
    sample, enc, added = _inputs(cfg, B, H, W, L)
    ref = U.unet_forward(P, cfg, sample, 501, enc, added_cond_kwargs=added)
    model = UNet2DConditionModel(cfg, P, use_graph=False)
    out = model(_cuda(sample), 501, _cuda(enc), added_cond_kwargs=_cuda(added), return_dict=False)[0]
    assert out.is_cuda and out.dtype == torch.float32 and out.shape == ref.shape
    r = _rel(out.cpu(), ref)
    print(f"{name}: rel-L2 vs oracle {r:.3e}")
    assert r < 2e-2, r
    # hipGraph replay == eager launches, bit for bit; and deterministic (reference test_determinism, 5e-4)
    gmodel = UNet2DConditionModel(cfg, P, use_graph=True)
    o1 = gmodel(_cuda(sample), 501, _cuda(enc), added_cond_kwargs=_cuda(added)).sample
    o2 = gmodel(_cuda(sample), 501, _cuda(enc), added_cond_kwargs=_cuda(added)).sample
    assert torch.equal(o1, out) and torch.equal(o1, o2)
    # a different timestep through the captured graph must change the result (inputs are re-staged, not baked in)
    o3 = gmodel(_cuda(sample), 21, _cuda(enc), added_cond_kwargs=_cuda(added)).sample
    ref3 = U.unet_forward(P, cfg, sample, 21, enc, added_cond_kwargs=added)
    assert _rel(o3.cpu(), ref3) < 2e-2 and not torch.equal(o3, o1)


@pytest.mark.parametrize("name,cfg,B,H,W", [("sd15-arch", SD15, 1, 32, 32), ("sdxl-arch", SDXL, 1, 32, 32)])
def test_real_architectures_reduced_resolution(name, cfg, B, H, W):
    """Full SD-1.5 / SDXL parameter sets (0.86 B / 2.57 B params), latent reduced so the CPU oracle finishes in seconds."""
    from paddlemix_amd.unet import UNet2DConditionModel
    Pd = _bf16_params(cfg, "cuda")
    model = UNet2DConditionModel(cfg, Pd, use_graph=True)
    P = {k: v.cpu() for k, v in Pd.items()}
    del Pd
    torch.cuda.empty_cache()
    sample, enc, added = _inputs(cfg, B, H, W)
    ref = U.unet_forward(P, cfg, sample, 301, enc, added_cond_kwargs=added)
    out = model(_cuda(sample), 301, _cuda(enc), added_cond_kwargs=_cuda(added)).sample
    r = _rel(out.cpu(), ref)
    print(f"{name}: rel-L2 vs oracle {r:.3e}")
    assert torch.isfinite(out).all() and r < 2e-2, r


def test_batch_independence_full_width():
    """Reference property test_inference_batch_single_identical (tests/pipelines/test_pipelines_common.py:478, 1e-2):
    the prompts of a batch do not interact -- what makes sharding prompts over GPUs exact."""
    from paddlemix_amd.unet import UNet2DConditionModel
    cfg = MINI_XL
    P = _bf16_params(cfg, "cuda")
    model = UNet2DConditionModel(cfg, P, use_graph=False)
    sample, enc, added = _inputs(cfg, 4, 32, 32)
    full = model(_cuda(sample), 77, _cuda(enc), added_cond_kwargs=_cuda(added)).sample
    for i in range(4):
        one = model(_cuda(sample[i:i + 1]), 77, _cuda(enc[i:i + 1]),
                    added_cond_kwargs={k: v[i:i + 1].cuda() for k, v in added.items()}).sample
        assert torch.allclose(one, full[i:i + 1], atol=1e-5, rtol=1e-5), (one - full[i:i + 1]).abs().max()


def test_encoder_attention_mask_on_device():
    """test_model_xattn_mask semantics (tests/models/test_models_unet_2d_condition.py:486-515) through the HIP path."""
    from paddlemix_amd.unet import UNet2DConditionModel
    cfg = TINY
    P = _bf16_params(cfg, "cpu")
    model = UNet2DConditionModel(cfg, P)
    sample, enc, _ = _inputs(cfg, 2, 16, 16, L=7)
    none = model(sample.cuda(), 10, enc.cuda()).sample
    keep = model(sample.cuda(), 10, enc.cuda(), encoder_attention_mask=torch.ones(2, 7, device="cuda")).sample
    assert torch.allclose(none, keep, rtol=1e-3, atol=1e-5)
    m = torch.ones(2, 7)
    m[:, -1] = 0
    masked = model(sample.cuda(), 10, enc.cuda(), encoder_attention_mask=m.cuda()).sample
    trunc = model(sample.cuda(), 10, enc[:, :-1].cuda()).sample
    assert torch.allclose(masked, trunc, rtol=1e-3, atol=2e-3), (masked - trunc).abs().max()
    ref = U.unet_forward(P, cfg, sample, 10, enc, encoder_attention_mask=m)
    assert _rel(masked.cpu(), ref) < 2e-2


@pytest.mark.parametrize("head_dim64", [False, True], ids=["d16", "d64-folded-scale"])
def test_self_attention_mask_on_device(head_dim64):
    """`attention_mask` (unet_2d_condition.py:916-923: a key mask over the latent tokens, a -10000 bias on every self-attention)
    through the HIP path, on the geometry where the reference can take it (all attention at one resolution); head_dim 64 runs the
    masked kernel on queries that carry the folded softmax scale (bias in base-2 units)."""
    from paddlemix_amd.unet import UNet2DConditionModel
    cfg = dict(TINY, attention_head_dim=2) if head_dim64 else TINY
    P = _bf16_params(cfg, "cpu")
    model = UNet2DConditionModel(cfg, P)
    sample, enc, _ = _inputs(cfg, 2, 16, 16, L=7)
    none = model(sample.cuda(), 10, enc.cuda()).sample
    keep = model(sample.cuda(), 10, enc.cuda(), attention_mask=torch.ones(2, 64, device="cuda")).sample
    assert _rel(keep.cpu(), none.cpu()) < 5e-3        # same arithmetic up to the masked kernel's accumulation order
    g = torch.Generator().manual_seed(3)
    m = (torch.rand(2, 64, generator=g) > 0.4).float()
    m[:, 0] = 1
    masked = model(sample.cuda(), 10, enc.cuda(), attention_mask=m.cuda()).sample
    ref = U.unet_forward(P, cfg, sample, 10, enc, attention_mask=m)
    assert _rel(masked.cpu(), none.cpu()) > 1e-2 and _rel(masked.cpu(), ref) < 2e-2
    with pytest.raises(ValueError, match="key tokens"):
        model(sample.cuda(), 10, enc.cuda(), attention_mask=torch.ones(2, 63, device="cuda"))


def test_controlnet_residuals_on_device():
    """down_block_additional_residuals / mid_block_additional_residual (unet_2d_condition.py:1121-1155) through the
    in-place NCHW-residual kernel on the concat-by-construction skip slots."""
    from paddlemix_amd.unet import UNet2DConditionModel
    from tests.test_host_logic import _controlnet_residuals
    cfg = MINI_XL
    P = _bf16_params(cfg, "cpu")
    model = UNet2DConditionModel(cfg, P)
    sample, enc, added = _inputs(cfg, 2, 32, 32, L=77)
    added_d = {k: v.cuda() for k, v in added.items()}
    down, mid = _controlnet_residuals(cfg, 2, 32, 32)
    out = model(sample.cuda(), 300, enc.cuda(), added_cond_kwargs=added_d,
                down_block_additional_residuals=[d.cuda() for d in down], mid_block_additional_residual=mid.cuda()).sample
    ref = U.unet_forward(P, cfg, sample, 300, enc, added_cond_kwargs=added, down_block_additional_residuals=down,
                         mid_block_additional_residual=mid)
    plain = model(sample.cuda(), 300, enc.cuda(), added_cond_kwargs=added_d).sample
    assert _rel(out.cpu(), ref) < 2e-2 and _rel(out.cpu(), plain.cpu()) > 5e-2


def test_denoise_loop_on_device_config1():
    """BASELINE config: SD pipeline loop, 1 prompt, CFG 7.5, 20 DDIM steps -- device loop vs the oracle loop
    (tiny UNet so the CPU side finishes in seconds). Stated tolerance on the final latents: rel-L2 <= 5e-2."""
    import numpy as np
    from oracle import schedulers_ref as S
    from paddlemix_amd.pipeline import StableDiffusionDenoiser
    from paddlemix_amd.schedulers import DDIMScheduler
    from paddlemix_amd.unet import UNet2DConditionModel
    cfg = TINY
    P = _bf16_params(cfg, "cpu")
    SCHED = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", steps_offset=1)
    g = torch.Generator().manual_seed(0)
    pe, ne = torch.randn(1, 7, 64, generator=g), torch.randn(1, 7, 64, generator=g)
    lat0 = torch.randn(1, 4, 16, 16, generator=g)
    pipe = StableDiffusionDenoiser(UNet2DConditionModel(cfg, P), DDIMScheduler(clip_sample=False, set_alpha_to_one=False, **SCHED))
    out = pipe(pe.cuda(), ne.cuda(), num_inference_steps=20, guidance_scale=7.5, latents=lat0.clone().cuda())
    sch = S.DDIMRef(clip_sample=False, set_alpha_to_one=False, **SCHED)
    sch.set_timesteps(20)
    x = lat0.numpy() * sch.init_noise_sigma
    emb = torch.cat([ne, pe])
    for t in sch.timesteps:
        eps = U.unet_forward(P, cfg, torch.from_numpy(np.concatenate([x, x])), int(t), emb).numpy()
        x = sch.step(eps[:1] + 7.5 * (eps[1:] - eps[:1]), t, x)
    rel = np.linalg.norm(out.cpu().numpy() - x) / np.linalg.norm(x)
    print(f"20-step DDIM CFG latents rel-L2 vs oracle loop: {rel:.3e}")
    assert rel < 5e-2, rel


@pytest.mark.parametrize("name", ["inpaint-9ch", "sd2-layout", "head-dim-tuple", "sin-first-shifted"])
def test_config_variants_on_device(name):
    """configuration switches of the reference's model tests (tests/configs.py UNET_VARIANTS) through the HIP kernels"""
    from paddlemix_amd.unet import UNet2DConditionModel
    cfg = UNET_VARIANTS[name]
    P = _bf16_params(cfg, "cpu")
    sample, enc, _ = _inputs(cfg, 2, 16, 16, 7)
    ref = U.unet_forward(P, cfg, sample, 333, enc)
    out = UNet2DConditionModel(cfg, P)(_cuda(sample), 333, _cuda(enc), return_dict=False)[0]
    assert _rel(out.cpu(), ref) < 2e-2, _rel(out.cpu(), ref)


@pytest.mark.parametrize("kind,concat", [("embedding", False), ("timestep", True), ("identity", True), ("simple_projection", False),
                                         ("projection", True)])
def test_class_embeddings_on_device(kind, concat):
    """class_labels path (unet_2d_condition.py:953-975): gather / sinusoid / projection kernels + residual or
    concatenating GEMM epilogues"""
    from paddlemix_amd.unet import UNet2DConditionModel
    from tests.test_host_logic import CLASS_CASES
    extra, make = CLASS_CASES[kind]
    cfg = dict(TINY, class_embeddings_concat=concat, **extra)
    P = _bf16_params(cfg, "cpu")
    sample, enc, _ = _inputs(cfg, 2, 16, 16, 7)
    labels = make(torch.Generator().manual_seed(5), 2, cfg)
    ref = U.unet_forward(P, cfg, sample, 333, enc, class_labels=labels)
    model = UNet2DConditionModel(cfg, P)
    out = model(_cuda(sample), 333, _cuda(enc), class_labels=labels.cuda(), return_dict=False)[0]
    assert _rel(out.cpu(), ref) < 2e-2, _rel(out.cpu(), ref)
    with pytest.raises(ValueError, match="class_labels should be provided"):
        model(_cuda(sample), 333, _cuda(enc))


def test_timestep_cond_on_device():
    """time_cond_proj_dim / timestep_cond (LCM guidance embedding): bias-free projection added to the sinusoid in place"""
    from paddlemix_amd.unet import UNet2DConditionModel
    cfg = dict(TINY, time_cond_proj_dim=32)
    P = _bf16_params(cfg, "cpu")
    sample, enc, _ = _inputs(cfg, 2, 16, 16, 7)
    w = torch.randn(2, 32, generator=torch.Generator().manual_seed(1))
    model = UNet2DConditionModel(cfg, P)
    out = model(_cuda(sample), 200, _cuda(enc), timestep_cond=w.cuda()).sample.cpu()
    assert _rel(out, U.unet_forward(P, cfg, sample, 200, enc, timestep_cond=w)) < 2e-2
    out0 = model(_cuda(sample), 200, _cuda(enc)).sample.cpu()
    assert _rel(out0, U.unet_forward(P, cfg, sample, 200, enc)) < 2e-2 and not torch.equal(out0, out)


@pytest.mark.parametrize("base", ["TINY", "MINI_XL"])
def test_ip_adapter_on_device(base):
    """IP-Adapter (encoder_hid_dim_type="ip_image_proj", unet_2d_condition.py:1054-1061; IPAdapterAttnProcessor,
    attention_processor.py:1793-1901): image_embeds -> ImageProjection -> per-block K/V of the image tokens -> a second,
    accumulating attention launch in every cross-attention"""
    from paddlemix_amd.unet import UNet2DConditionModel
    from tests import configs
    cfg = dict(getattr(configs, base), encoder_hid_dim_type="ip_image_proj", encoder_hid_dim=96)
    P = _bf16_params(cfg, "cpu")
    sample, enc, added = _inputs(cfg, 2, 16, 16, 7 if base == "TINY" else 77)
    img = torch.randn(2, 96, generator=torch.Generator().manual_seed(9))
    kw = dict(added or {}, image_embeds=img)
    ckw = {k: v.cuda() for k, v in kw.items()}
    model = UNet2DConditionModel(cfg, P)
    outs = {}
    for sc in (1.0, 0.5, 0.0):
        model.set_ip_adapter_scale(sc)
        out = model(_cuda(sample), 400, _cuda(enc), added_cond_kwargs=ckw, return_dict=False)[0].cpu()
        ref = U.unet_forward(P, cfg, sample, 400, enc, added_cond_kwargs=kw, ip_adapter_scale=sc)
        r = _rel(out, ref)
        print(f"ip-adapter {base} scale {sc}: rel-L2 vs oracle {r:.3e}")
        assert r < 2e-2, (sc, r)
        outs[sc] = out
    assert not torch.equal(outs[1.0], outs[0.0])
    with pytest.raises(ValueError, match="image_embeds"):
        model(_cuda(sample), 400, _cuda(enc), added_cond_kwargs={k: v for k, v in ckw.items() if k != "image_embeds"} or None)


def test_img2img_inpaint_lcm_pipelines_on_device():
    """The image-conditioned entries of the SD loop with every tensor on the GPU (VAE encode -> posterior sample -> add_noise ->
    loop -> decode): finite, generator-reproducible, and for the 4-channel inpaint form the kept region comes back. The
    arithmetic of these loops is pinned against oracle loops in tests/test_pipeline.py (CPU, ABI emulator)."""
    from paddlemix_amd.pipeline import StableDiffusionDenoiser
    from paddlemix_amd.schedulers import DDIMScheduler, EulerDiscreteScheduler, LCMScheduler
    from paddlemix_amd.unet import UNet2DConditionModel, synth_unet_params
    from paddlemix_amd.vae import AutoencoderKL, synth_vae_params
    from tests.configs import MINI_VAE
    sk = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear")
    vae = AutoencoderKL(MINI_VAE, synth_vae_params(MINI_VAE, seed=6))
    g = torch.Generator().manual_seed(8)
    pe, ne = torch.randn(2, 7, 64, generator=g).cuda(), torch.randn(2, 7, 64, generator=g).cuda()
    image = (torch.rand(2, 3, 64, 64, generator=g) * 2 - 1).cuda()
    mask = torch.zeros(2, 1, 64, 64, device="cuda")
    mask[:, :, 16:48, 24:64] = 1.0
    gen = lambda s: torch.Generator(device="cuda").manual_seed(s)  # noqa: E731

    unet = UNet2DConditionModel(TINY, synth_unet_params(TINY, seed=1234))
    pipe = StableDiffusionDenoiser(unet, DDIMScheduler(clip_sample=False, set_alpha_to_one=False, steps_offset=1, **sk), vae=vae)
    a = pipe(pe, ne, num_inference_steps=6, guidance_scale=4.0, image=image, strength=0.5, generator=gen(1), output_type="pt")
    b = pipe(pe, ne, num_inference_steps=6, guidance_scale=4.0, image=image, strength=0.5, generator=gen(1), output_type="pt")
    assert a.shape == (2, 3, 64, 64) and torch.isfinite(a).all() and torch.equal(a, b)
    lat = pipe(pe, ne, num_inference_steps=4, guidance_scale=4.0, image=image, mask_image=mask, generator=gen(2))
    dist = vae.encode(image).latent_dist
    keep = (torch.nn.functional.interpolate(mask, size=lat.shape[-2:]) == 0).expand_as(lat)
    redo = pipe(pe, ne, num_inference_steps=4, guidance_scale=4.0, image=image, mask_image=mask, generator=gen(2))
    assert torch.isfinite(lat).all() and torch.equal(lat, redo)
    # kept region = the image posterior sample the call drew (mean +- a few std), not the noise it started from
    z = (lat / MINI_VAE["scaling_factor"] - dist.mean) / dist.std
    assert z[keep].abs().max() < 6.0 and z[~keep].abs().max() > z[keep].abs().max()

    cfg9 = dict(TINY, in_channels=9)
    pipe9 = StableDiffusionDenoiser(UNet2DConditionModel(cfg9, synth_unet_params(cfg9, seed=7)),
                                    EulerDiscreteScheduler(timestep_spacing="leading", steps_offset=1, **sk), vae=vae)
    img9 = pipe9(pe, num_inference_steps=8, guidance_scale=1.0, image=image, mask_image=mask, strength=0.75, generator=gen(3),
                 output_type="pt")
    assert img9.shape == (2, 3, 64, 64) and torch.isfinite(img9).all()

    cfgl = dict(TINY, time_cond_proj_dim=32)
    pipel = StableDiffusionDenoiser(UNet2DConditionModel(cfgl, synth_unet_params(cfgl, seed=5)), LCMScheduler(**sk), vae=vae)
    il = pipel(pe, num_inference_steps=4, guidance_scale=8.0, height=64, width=64, vae_scale_factor=4, generator=gen(4), output_type="pt")
    il2 = pipel(pe, num_inference_steps=4, guidance_scale=8.0, height=64, width=64, vae_scale_factor=4, generator=gen(4), output_type="pt")
    assert il.shape == (2, 3, 64, 64) and torch.isfinite(il).all() and torch.equal(il, il2)


@pytest.mark.parametrize("base", ["TINY", "MINI_XL"])
def test_controlnet_on_device(base):
    """ControlNetModel (controlnet.py:671-877) through the C ABI: SiLU-epilogue convs of the conditioning embedding, the
    UNet's encoder half, zero-convolution GEMMs with conditioning_scale as output scale -> fp32 NCHW residuals; then the
    residuals into the UNet (unet_2d_condition.py:1121-1155)."""
    from paddlemix_amd.unet import ControlNetModel, UNet2DConditionModel, synth_controlnet_params
    from tests import configs
    cfg = getattr(configs, base)
    P = {k: (v.to(torch.bfloat16).float() if v.dim() > 1 else v) for k, v in synth_controlnet_params(cfg, 3).items()}
    sample, enc, added = _inputs(cfg, 2, 16, 16, 7 if base == "TINY" else 77)
    cond = torch.rand(2, 3, 128, 128, generator=torch.Generator().manual_seed(4))
    net = ControlNetModel(cfg, P)
    worst = 0.0
    for sc, gm in ((1.0, False), (0.6, True)):
        out = net(_cuda(sample), 20, _cuda(enc), cond.cuda(), conditioning_scale=sc, guess_mode=gm, added_cond_kwargs=_cuda(added))
        downs, mid = U.controlnet_forward(P, cfg, sample, 20, enc, cond, sc, gm, added_cond_kwargs=added)
        for a, b in zip(out.down_block_res_samples + (out.mid_block_res_sample,), downs + (mid,)):
            assert a.shape == b.shape and a.dtype == torch.float32
            worst = max(worst, _rel(a.cpu(), b))
    print(f"controlnet {base}: worst rel-L2 of a residual vs oracle {worst:.3e}")
    assert worst < 2e-2, worst
    d, m = net(_cuda(sample), 20, _cuda(enc), cond.cuda(), added_cond_kwargs=_cuda(added), return_dict=False)
    d2, m2 = net(_cuda(sample), 20, _cuda(enc), cond.cuda(), added_cond_kwargs=_cuda(added), return_dict=False)
    assert all(torch.equal(a, b) for a, b in zip(d + (m,), d2 + (m2,)))          # graph replay is deterministic
    Pu = _bf16_params(cfg, "cpu")
    unet = UNet2DConditionModel(cfg, Pu)
    got = unet(_cuda(sample), 20, _cuda(enc), added_cond_kwargs=_cuda(added), down_block_additional_residuals=d,
               mid_block_additional_residual=m).sample.cpu()
    rd, rm = U.controlnet_forward(P, cfg, sample, 20, enc, cond, added_cond_kwargs=added)
    ref = U.unet_forward(Pu, cfg, sample, 20, enc, added_cond_kwargs=added, down_block_additional_residuals=rd,
                         mid_block_additional_residual=rm)
    assert _rel(got, ref) < 2e-2, _rel(got, ref)


@pytest.mark.parametrize("name,cfg,B,H,W,L", [("tiny", TINY, 2, 16, 16, 7), ("mini-xl", MINI_XL, 2, 32, 32, 77)])
def test_fp32_residual_stream_on_device(name, cfg, B, H, W, L):
    """residual_dtype="fp32" (fp32 skip slots / hidden state, 16-bit values only as MFMA operands) through the HIP kernels:
    closer to the oracle than the 16-bit stream, deterministic, graph == eager."""
    from paddlemix_amd.unet import UNet2DConditionModel
    P = _bf16_params(cfg, "cpu")
    sample, enc, added = _inputs(cfg, B, H, W, L)
    ref = U.unet_forward(P, cfg, sample, 501, enc, added_cond_kwargs=added)
    o16 = UNet2DConditionModel(cfg, P)(_cuda(sample), 501, _cuda(enc), added_cond_kwargs=_cuda(added)).sample
    m32 = UNet2DConditionModel(cfg, P, residual_dtype="fp32", use_graph=False)
    o32 = m32(_cuda(sample), 501, _cuda(enc), added_cond_kwargs=_cuda(added)).sample
    r16, r32 = _rel(o16.cpu(), ref), _rel(o32.cpu(), ref)
    print(f"{name}: rel-L2 vs oracle, 16-bit stream {r16:.3e}, fp32 stream {r32:.3e}")
    assert r32 < r16 and r16 < BARS[name][0] and r32 < BARS[name][1], (r16, r32)
    g32 = UNet2DConditionModel(cfg, P, residual_dtype="fp32", use_graph=True)
    a = g32(_cuda(sample), 501, _cuda(enc), added_cond_kwargs=_cuda(added)).sample
    b = g32(_cuda(sample), 501, _cuda(enc), added_cond_kwargs=_cuda(added)).sample
    assert torch.equal(a, o32) and torch.equal(a, b)


@pytest.mark.parametrize("name,cfg,H,W", [("sd15-1x4x64x64", SD15, 64, 64), ("sdxl-1x4x128x128", SDXL, 128, 128)])
def test_headline_geometry_vs_oracle(name, cfg, H, W):
    """The geometry the metric is quoted on: full SD-1.5 parameter set at 1x4x64x64 (BASELINE.json config 2) and full SDXL
    parameter set at 1x4x128x128 (one prompt of the bs-8 headline; prompts do not interact -- test_batch_independence) against
    the oracle: S = 4096 / 1024 self-attention, 16384- / 131072-row convolutions, the large-M GEMM tiles. The reference's own
    real-size check is tests/models/test_models_unet_2d_condition.py:745-827 (64x64 latents, real weights: unavailable here)."""
    from paddlemix_amd.unet import UNet2DConditionModel
    Pd = _bf16_params(cfg, "cuda")
    m16 = UNet2DConditionModel(cfg, Pd, use_graph=False)
    m32 = UNet2DConditionModel(cfg, Pd, residual_dtype="fp32", use_graph=False)
    P = {k: v.cpu() for k, v in Pd.items()}
    del Pd
    sample, enc, added = _inputs(cfg, 1, H, W)
    o16 = m16(_cuda(sample), 501, _cuda(enc), added_cond_kwargs=_cuda(added)).sample.cpu()
    o32 = m32(_cuda(sample), 501, _cuda(enc), added_cond_kwargs=_cuda(added)).sample.cpu()
    del m16, m32
    torch.cuda.empty_cache()
    ref = U.unet_forward(P, cfg, sample, 501, enc, added_cond_kwargs=added)
    r16, r32 = _rel(o16, ref), _rel(o32, ref)
    print(f"{name}: rel-L2 vs oracle, 16-bit stream {r16:.3e}, fp32 stream {r32:.3e}")
    assert torch.isfinite(o16).all() and torch.isfinite(o32).all()
    assert r16 < BARS[name][0] and r32 < BARS[name][1], (r16, r32)


def test_euler30_latents_vs_float64_oracle_loop():
    """30 Euler steps (the reference's SDXL test scheduler, tests/pipelines/stable_diffusion_xl/test_stable_diffusion_xl.py:84-90)
    on the SDXL-structured mini UNet: per-step epsilon error with the device fed the oracle's latents, and end-latent error of
    the free-running device loop, against a float64 oracle loop -- the quantity north_star's tolerance is stated on."""
    import numpy as np
    from oracle import schedulers_ref as S
    from paddlemix_amd.unet import UNet2DConditionModel
    cfg = MINI_XL
    P = _bf16_params(cfg, "cpu")
    P64 = {k: v.double() for k, v in P.items()}
    sample, enc, added = _inputs(cfg, 2, 32, 32)
    added64 = {k: v.double() for k, v in added.items()}
    sch = S.EulerRef(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", timestep_spacing="leading", steps_offset=1)
    sch.set_timesteps(30)
    sig = sch.sigmas.astype(np.float64)
    res = {}
    # the float64 oracle loop does not depend on the device mode: computed once
    x_ref = sample.double() * float(sch.init_noise_sigma)
    xins, e_refs = [], []
    for i, t in enumerate(sch.timesteps):
        xins.append(x_ref / (sig[i] * sig[i] + 1.0) ** 0.5)
        e_refs.append(U.unet_forward(P64, cfg, xins[-1], int(t), enc.double(), added_cond_kwargs=added64))
        x_ref = x_ref + e_refs[-1] * (sig[i + 1] - sig[i])
    for rd, bar_eps, bar_lat in (("16", BARS["euler30-eps"][0], BARS["euler30-latents"][0]),
                                 ("fp32", BARS["euler30-eps"][1], BARS["euler30-latents"][1])):
        model = UNet2DConditionModel(cfg, P, residual_dtype=rd)
        x_ref = sample.double() * float(sch.init_noise_sigma)
        x_dev = x_ref.clone()
        worst = 0.0
        for i, t in enumerate(sch.timesteps):
            s = sig[i]
            xin, e_ref = xins[i], e_refs[i]
            e_tf = model(_cuda(xin.float()), int(t), _cuda(enc), added_cond_kwargs=_cuda(added)).sample.cpu().double()
            worst = max(worst, _rel(e_tf, e_ref))
            e_fr = model(_cuda((x_dev / (s * s + 1.0) ** 0.5).float()), int(t), _cuda(enc), added_cond_kwargs=_cuda(added)).sample
            x_ref = x_ref + e_ref * (sig[i + 1] - s)
            x_dev = x_dev + e_fr.cpu().double() * (sig[i + 1] - s)
        res[rd] = (worst, _rel(x_dev, x_ref))
        print(f"30 Euler steps, residual {rd}: worst per-step eps rel-L2 {worst:.3e}, end latents rel-L2 {res[rd][1]:.3e}")
        assert worst < bar_eps and res[rd][1] < bar_lat, (rd, res[rd])
    assert res["fp32"][1] < res["16"][1]


def test_from_pretrained_into_the_hip_path(tmp_path):
    """SURVEY 8f.2 on the device: a model directory (config.json + torch-layout safetensors, the diffusers file name) loaded
    by from_pretrained straight into the HIP program equals the model built from the same parameters in memory, bit for bit."""
    from paddlemix_amd import checkpoint as C
    from paddlemix_amd.unet import UNet2DConditionModel, synth_unet_params, unet_param_shapes
    cfg = dict(MINI_XL)
    P = synth_unet_params(cfg, seed=11)
    C.save_pretrained(str(tmp_path / "unet"), cfg, P, data_format="pt", shapes=unet_param_shapes(cfg))
    sample, enc, added = _inputs(cfg, 1, 16, 16)
    want = UNet2DConditionModel(cfg, P)(_cuda(sample), 77, _cuda(enc), added_cond_kwargs=_cuda(added)).sample
    model = UNet2DConditionModel.from_pretrained(str(tmp_path), subfolder="unet")
    got = model(_cuda(sample), 77, _cuda(enc), added_cond_kwargs=_cuda(added)).sample
    assert torch.equal(got, want)
    ref = U.unet_forward({k: (v.to(torch.bfloat16).float() if v.dim() > 1 else v) for k, v in P.items()}, cfg, sample, 77, enc,
                         added_cond_kwargs=added)
    assert _rel(got.cpu(), ref) < BAR_16


def test_two_models_on_two_streams_match_their_serial_runs_bit_for_bit():
    """ABI 12: the split-K / widening scratch is an argument of every GEMM-class call and belongs to the model that
    plans the call -- until ABI 11 it was ONE process-wide binding that two models on two streams shared (and that a stale pointer
    could survive in). Two models with their own weights and inputs, whose batch-2 16x16 geometry puts most GEMMs on split-K: run one
    after the other, then both at once on their own streams without any synchronisation in between, many times; every concurrent
    result must equal the model's serial result bit for bit. The C++ handle API takes part as a third concurrent client."""
    from paddlemix_amd.cexec import CUNet2DConditionModel
    from paddlemix_amd.unet import UNet2DConditionModel, synth_unet_params
    cfg = MINI_XL
    models, inputs, serial = [], [], []
    for seed in (11, 12):
        P = synth_unet_params(cfg, seed=seed)
        m = UNet2DConditionModel(cfg, P, device="cuda:0")
        s, e, a = _inputs(cfg, 2, 16, 16, seed=seed)
        s, e, a = _cuda(s), _cuda(e), _cuda(a)
        plan = m._get_plan(2, 16, 16, 77)
        m.stage_inputs(plan, s, 500.0, e, a, in_scale=1.0)
        torch.cuda.synchronize()
        m.run(plan)
        torch.cuda.synchronize()                                                          # (the replay runs on the model's own stream)
        serial.append(plan.out.clone())
        for _ in range(2):                                                                # alone: the same bits every time
            m.run(plan)
            torch.cuda.synchronize()
            assert torch.equal(plan.out, serial[-1])
        models.append((m, plan))
        inputs.append((s, e, a))
    assert models[0][0]._gemm_ws[0] != models[1][0]._gemm_ws[0]                      # each model owns its scratch
    assert any(fn.__name__ == "mi355x_sd_linear" and args[-3] == models[0][0]._gemm_ws[0] for fn, args, _, _ in models[0][1].prog)
    cm = CUNet2DConditionModel(cfg, synth_unet_params(cfg, seed=11), device="cuda:0")
    c_serial = cm(inputs[0][0], 500.0, inputs[0][1], added_cond_kwargs=inputs[0][2], return_dict=False)[0].clone()
    torch.cuda.synchronize()
    bad = {}
    for it in range(12):
        outs = [m.run(plan) for m, plan in models]                                       # two replays in flight, two streams
        c_out = cm(inputs[0][0], 500.0, inputs[0][1], added_cond_kwargs=inputs[0][2], return_dict=False)[0]
        torch.cuda.synchronize()
        for k, (o, ref) in enumerate(list(zip(outs, serial)) + [(c_out, c_serial)]):
            if not torch.equal(o, ref):
                bad[(it, k)] = ((o.float() - ref.float()).abs().max().item(), int((o != ref).sum().item()))
    assert not bad, bad
