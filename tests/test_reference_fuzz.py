"""Random UNet configurations inside the supported space -- block types per level, per-level head counts and layer counts, linear /
conv projections, up-cast attention, class / text_time / timestep_cond conditioning, ragged and odd latent sizes -- run three ways: the
reference's own UNet2DConditionModel (over oracle/paddle_shim.py), the oracle, and the MI355X model on the emulated device. The
reference and the oracle must agree to fp32 rounding, the device program to its 16-bit tolerance. Build container only (live
reference); `python tests/test_reference_fuzz.py <seed> <trials>` runs more."""
import random
import sys

import pytest
import torch

from oracle import reference_runner as rr
from oracle import unet_ref as U


def one_trial(trial: int, rng: random.Random):
    from paddlemix_amd.unet import UNet2DConditionModel
    from tests.abi_emulator import Emulator, on_emulator
    n = rng.choice([2, 3])
    boc = tuple(rng.choice([32, 64]) * (2 ** min(i, 1)) for i in range(n))
    down = tuple(rng.choice(["DownBlock2D", "CrossAttnDownBlock2D"]) for _ in range(n))
    up = tuple("CrossAttnUpBlock2D" if d == "CrossAttnDownBlock2D" else "UpBlock2D" for d in reversed(down))
    cfg = dict(block_out_channels=boc, down_block_types=down, up_block_types=up, cross_attention_dim=rng.choice([32, 64]),
               attention_head_dim=rng.choice([4, 8, (4, 8, 8)[:n]]), layers_per_block=rng.choice([1, 2, (1, 2, 1)[:n]]),
               transformer_layers_per_block=rng.choice([1, 2]), use_linear_projection=rng.choice([True, False]),
               upcast_attention=rng.choice([True, False]), norm_num_groups=32, sample_size=16,
               flip_sin_to_cos=rng.choice([True, False]), freq_shift=rng.choice([0, 1]))
    extra = rng.choice([None, "class", "text_time", "tcond"])
    kw = {}
    g = torch.Generator().manual_seed(trial)
    if extra == "class":
        cfg["num_class_embeds"] = 7
        kw["class_labels"] = torch.tensor([2, 5])
    elif extra == "text_time":
        cfg.update(addition_embed_type="text_time", addition_time_embed_dim=16, projection_class_embeddings_input_dim=16 * 6 + 32)
        kw["added_cond_kwargs"] = dict(text_embeds=torch.randn(2, 32, generator=g), time_ids=torch.tensor([[64.0, 64.0, 0.0, 0.0, 64.0, 64.0]]).repeat(2, 1))
    elif extra == "tcond":
        cfg["time_cond_proj_dim"] = 16
        kw["timestep_cond"] = torch.randn(2, 16, generator=g)
    hw = rng.choice([(8, 8), (16, 8), (12, 12), (10, 14)])
    P = U.synth_unet_params(cfg, seed=trial)
    x, enc, t = torch.randn(2, 4, *hw, generator=g), torch.randn(2, 5, cfg["cross_attention_dim"], generator=g), torch.tensor([37.0, 37.0])
    with torch.no_grad():
        ora = U.unet_forward(P, cfg, x, t, enc, **kw)
        ref = rr.from_shim(rr.build_unet(cfg, P)(rr.to_shim(x), rr.to_shim(t), rr.to_shim(enc), **rr.to_shim(kw)).sample)
        orab = U.unet_forward({k: (v.to(torch.bfloat16).float() if v.dim() > 1 else v) for k, v in P.items()}, cfg, x, t, enc, **kw)
    prod = on_emulator(UNet2DConditionModel, cfg, P)(x, 37.0, enc, **kw).sample
    d_ref = float((ora - ref).abs().max() / ref.abs().max())
    d_dev = float((prod - orab).norm() / orab.norm())
    return cfg, extra, hw, d_ref, d_dev


@pytest.mark.skipif(not rr.available(), reason="/root/reference exists only in the build container")
@pytest.mark.parametrize("seed", [0, 1])
def test_random_unet_configurations(seed):
    rng = random.Random(seed)
    for trial in range(6):
        cfg, extra, hw, d_ref, d_dev = one_trial(100 * seed + trial, rng)
        assert d_ref < 5e-5 and d_dev < 2.5e-2, (cfg, extra, hw, d_ref, d_dev)


_REF_SCHED = {"DDIMScheduler": "scheduling_ddim", "EulerDiscreteScheduler": "scheduling_euler_discrete", "PNDMScheduler": "scheduling_pndm",
              "DPMSolverMultistepScheduler": "scheduling_dpmsolver_multistep"}


def one_scheduler_trial(trial: int, rng: random.Random):
    """-> (description, outcome of the reference class, outcome of paddlemix_amd.schedulers' class); an outcome is ("ok", latents,
    timesteps) | ("nan",) | ("raises", type name, message)"""
    import math

    import paddlemix_amd.schedulers as PS
    cls = rng.choice(list(_REF_SCHED))
    kw = dict(beta_start=0.00085, beta_end=0.012, beta_schedule=rng.choice(["linear", "scaled_linear"]),
              prediction_type=rng.choice(["epsilon", "v_prediction"]), timestep_spacing=rng.choice(["leading", "trailing", "linspace"]),
              steps_offset=rng.choice([0, 1]))
    if cls == "DDIMScheduler":
        kw.update(clip_sample=rng.choice([True, False]), set_alpha_to_one=rng.choice([True, False]))
    elif cls == "EulerDiscreteScheduler":
        kw.update(use_karras_sigmas=rng.choice([True, False]))
    elif cls == "PNDMScheduler":
        kw.update(skip_prk_steps=rng.choice([True, False]), set_alpha_to_one=rng.choice([True, False]))
    else:
        kw.update(solver_order=rng.choice([1, 2]), algorithm_type=rng.choice(["dpmsolver++", "dpmsolver"]), solver_type=rng.choice(["midpoint", "heun"]),
                  use_karras_sigmas=rng.choice([True, False]), euler_at_final=rng.choice([True, False]), lower_order_final=rng.choice([True, False]))
    steps = rng.choice([3, 5, 8, 12, 14, 20])
    g = torch.Generator().manual_seed(trial)
    x0, pat = torch.randn(1, 4, 8, 8, generator=g), torch.randn(1, 4, 8, 8, generator=g)
    model = lambda x, t: 0.3 * x * math.cos(0.01 * float(t)) + 0.1 * pat  # noqa: E731

    def loop(sch, wrap, unwrap):
        try:
            sch.set_timesteps(steps)
            ins = getattr(sch, "init_noise_sigma", 1.0)
            x = wrap(x0 * float(ins if isinstance(ins, (int, float)) else unwrap(ins)))
            for t in sch.timesteps:
                x = sch.step(wrap(model(unwrap(sch.scale_model_input(x, t)), unwrap(t))), t, x, return_dict=False)[0]
            x = unwrap(x).float()
            return ("nan",) if torch.isnan(x).any() else ("ok", x, [float(unwrap(t)) for t in sch.timesteps])
        except Exception as e:   # noqa: BLE001
            return ("raises", type(e).__name__, str(e)[:60])

    ref = loop(getattr(rr.ref_module(_REF_SCHED[cls], "schedulers"), cls)(**kw), rr.to_shim, rr.from_shim)
    prod = loop(getattr(PS, cls)(**kw), lambda v: v, lambda v: v)
    return f"{cls}({kw}) x {steps}", ref, prod


@pytest.mark.skipif(not rr.available(), reason="/root/reference exists only in the build container")
def test_random_scheduler_configurations():
    """40 random (class, beta schedule, prediction type, spacing, offsets, solver options, step count) combinations: the product's
    scheduler follows the reference's class to 5e-5 -- and fails the way the reference fails where the reference fails (NaN from
    DPM-Solver's h = inf second-order final step; the 3-step PRK schedule of PNDM that cannot be built)"""
    import warnings
    rng = random.Random(0)
    kinds = set()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for trial in range(40):
            what, ref, prod = one_scheduler_trial(trial, rng)
            assert ref[0] == prod[0], (what, ref[:3], prod[:3])
            kinds.add(ref[0])
            if ref[0] == "raises":
                assert ref[1:] == prod[1:], (what, ref, prod)
            elif ref[0] == "ok":
                assert len(ref[2]) == len(prod[2]) and max(abs(a - b) for a, b in zip(ref[2], prod[2])) < 1e-3, what
                assert float((prod[1] - ref[1]).abs().max() / ref[1].abs().max()) < 5e-5, what
    assert "ok" in kinds


@pytest.mark.skipif(not rr.available(), reason="/root/reference exists only in the build container")
def test_random_pipeline_calls():
    """8 random text-to-image calls -- SD or SDXL (micro-conditioning), DDIM / Euler (Karras) / PNDM / DPM-Solver, leading / trailing
    spacing, 2-5 steps, guidance 1 / 3 / 7.5, guidance_rescale 0 / 0.6, batch 1 / 2: paddlemix_amd.pipeline.StableDiffusionDenoiser with
    the product's scheduler and the UNet on the emulated device against the reference's own pipeline __call__ with the reference's
    model and scheduler. Tolerance: chained 16-bit model evaluations under guidance."""
    import warnings

    import paddlemix_amd.schedulers as PS
    from paddlemix_amd.pipeline import StableDiffusionDenoiser
    from paddlemix_amd.unet import UNet2DConditionModel
    from tests import configs as C
    from tests import reference_cases as RC
    from tests.abi_emulator import Emulator, on_emulator
    rng = random.Random(0)
    warnings.simplefilter("ignore")
    for trial in range(8):
        xl = rng.choice([False, True])
        cfg = C.MINI_XL if xl else C.TINY
        cls = rng.choice(list(_REF_SCHED))
        kw = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", steps_offset=1, timestep_spacing=rng.choice(["leading", "trailing"]))
        if cls == "DDIMScheduler":
            kw.update(clip_sample=False, set_alpha_to_one=False)
        elif cls == "PNDMScheduler":
            kw.update(skip_prk_steps=True)
        elif cls == "EulerDiscreteScheduler":
            kw.update(use_karras_sigmas=rng.choice([True, False]))
        steps, gs, gr, B = rng.choice([2, 3, 5]), rng.choice([1.0, 3.0, 7.5]), rng.choice([0.0, 0.6]), rng.choice([1, 2])
        g = torch.Generator().manual_seed(trial)
        cd = cfg["cross_attention_dim"]
        pe, ne, lat0 = torch.randn(B, 7, cd, generator=g), torch.randn(B, 7, cd, generator=g), torch.randn(B, 4, 8, 8, generator=g)
        P = U.synth_unet_params(cfg, seed=trial)
        sh = rr.to_shim
        common = dict(latents=sh(lat0.clone()), num_inference_steps=steps, guidance_scale=gs, guidance_rescale=gr, output_type="latent", height=64, width=64,
                      return_dict=False)
        akw = {}
        with torch.no_grad():
            if xl:
                pp, npp = torch.randn(B, 64, generator=g), torch.randn(B, 64, generator=g)
                pm = rr.ref_pipeline("pipeline_stable_diffusion_xl", "pipelines.stable_diffusion_xl")
                te2 = type("TextEncoder2", (), {"config": rr.FrozenConfig(projection_dim=64), "dtype": torch.float32})()
                sched = getattr(rr.ref_module(_REF_SCHED[cls], "schedulers"), cls)(**kw)
                pipe = pm.StableDiffusionXLPipeline(vae=RC._FakeVAE(rr, scaling_factor=0.13025, force_upcast=False), text_encoder=None, text_encoder_2=te2,
                                                    tokenizer=None, tokenizer_2=None, unet=rr.build_unet(cfg, P), scheduler=sched)
                ref = pipe(prompt_embeds=sh(pe), negative_prompt_embeds=sh(ne), pooled_prompt_embeds=sh(pp), negative_pooled_prompt_embeds=sh(npp), **common)[0]
                tids = torch.tensor([[64.0, 64.0, 0.0, 0.0, 64.0, 64.0]]).repeat(B, 1)
                akw = dict(added_cond_kwargs=dict(text_embeds=pp, time_ids=tids), negative_added_cond_kwargs=dict(text_embeds=npp, time_ids=tids))
            else:
                pipe = RC._sd_parts(rr, "pipeline_stable_diffusion", "StableDiffusionPipeline", cfg, P, _REF_SCHED[cls], cls, kw)
                ref = pipe(prompt_embeds=sh(pe), negative_prompt_embeds=sh(ne), **common)[0]
        ref = rr.from_shim(ref)
        out = StableDiffusionDenoiser(on_emulator(UNet2DConditionModel, cfg, P), getattr(PS, cls)(**kw))(
            pe, ne, num_inference_steps=steps, guidance_scale=gs, guidance_rescale=gr, latents=lat0.clone(), **akw)
        d = float((out - ref).norm() / ref.norm())
        assert d < 5e-2, (trial, xl, cls, kw, steps, gs, gr, B, d)


if __name__ == "__main__":
    rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
        cfg, extra, hw, d_ref, d_dev = one_trial(trial, rng)
        print(f"trial {trial}: {len(cfg['block_out_channels'])} levels {[d[0] for d in cfg['down_block_types']]} {extra} {hw}: "
              f"oracle vs reference {d_ref:.1e}, device program vs oracle {d_dev:.2e}")
