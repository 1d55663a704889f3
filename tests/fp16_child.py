"""Child process of tests/test_fp16_mode.py / tests/test_gpu_fp16.py: runs with MI355X_SD_DTYPE=fp16 (one element type
per process) and prints one JSON line of parity numbers for the IEEE-half build of the library.

  python tests/fp16_child.py cpu   -- host logic through the ABI emulator (fp32 math, fp16 stores)
  python tests/fp16_child.py gpu   -- the HIP kernels of libmi355x_sd_f16.so on cuda:0
  python tests/fp16_child.py gpu-kernels   -- the kernel-level part of the above only
"""
import json
import math
import os
import sys

os.environ["MI355X_SD_DTYPE"] = "fp16"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import unet_ref as U  # noqa: E402
from paddlemix_amd import _lib  # noqa: E402
from paddlemix_amd.unet import UNet2DConditionModel, synth_unet_params  # noqa: E402
from tests.configs import MINI_XL, TINY  # noqa: E402
from tests.test_host_logic import _inputs, _rel  # noqa: E402


def fp32_residual_cases(make_model, dev, headline: bool):
    """fp16 elements + fp32 residual stream: single forwards, 30 Euler steps on the SDXL-structured mini UNet against a float64
    oracle loop; `headline` adds the full SDXL parameter set at 1x4x128x128 (not run by the suite since round 6:
    tests/test_gpu_parity_loops.py holds that geometry for this build, both stream types, against committed float64 fixtures)"""
    import numpy as np
    from oracle import schedulers_ref as S
    from tests.configs import SDXL
    out = {}
    mv = (lambda t: t.to(dev)) if dev else (lambda t: t)
    mvd = lambda d: None if d is None else {k: mv(v) for k, v in d.items()}  # noqa: E731
    cases = [("tiny", TINY, 2, 16, 16, 7), ("mini_xl", MINI_XL, 2 if dev else 1, 32 if dev else 16, 32 if dev else 16, 77)]
    if headline:
        cases.append(("sdxl_1x4x128x128", SDXL, 1, 128, 128, 77))
    for name, cfg, B, H, W, L in cases:
        P = synth_unet_params(cfg, seed=1234)
        Ph = {k: v.to(torch.float16).float() if v.dim() > 1 else v for k, v in P.items()}
        sample, enc, added = _inputs(cfg, B, H, W, L)
        ref = U.unet_forward(Ph, cfg, sample, 501, enc, added_cond_kwargs=added)
        r = {}
        for rd in ("16", "fp32"):
            model = make_model(cfg, P, rd)
            got = model(mv(sample), 501, mv(enc), added_cond_kwargs=mvd(added), return_dict=False)[0].float().cpu()
            r["resid_" + rd] = _rel(got, ref)
            del model
        out[name] = r
    # 30 Euler steps, float64 oracle loop (the quantity north_star's 1e-3 is stated on: the latents)
    cfg, B, H, W, L = MINI_XL, 2, (32 if dev else 16), (32 if dev else 16), 77
    P = synth_unet_params(cfg, seed=1234)
    Ph = {k: v.to(torch.float16).float() if v.dim() > 1 else v for k, v in P.items()}
    P64 = {k: v.double() for k, v in Ph.items()}
    sample, enc, added = _inputs(cfg, B, H, W, L)
    added64 = {k: v.double() for k, v in added.items()}
    sch = S.EulerRef(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", timestep_spacing="leading", steps_offset=1)
    sch.set_timesteps(30 if dev else 6)
    sig = sch.sigmas.astype(np.float64)
    loop = {}
    # the float64 oracle loop does not depend on the device mode: once (round 6: it used to run per mode, the suite's second-longest item)
    x_ref = sample.double() * float(sch.init_noise_sigma)
    xins, e_refs = [], []
    for i, t in enumerate(sch.timesteps):
        xins.append(x_ref / (sig[i] * sig[i] + 1.0) ** 0.5)
        e_refs.append(U.unet_forward(P64, cfg, xins[-1], int(t), enc.double(), added_cond_kwargs=added64))
        x_ref = x_ref + e_refs[-1] * (sig[i + 1] - sig[i])
    x_ref_end = x_ref
    for rd in ("16", "fp32"):
        model = make_model(cfg, P, rd)
        x_ref = sample.double() * float(sch.init_noise_sigma)
        x_dev = x_ref.clone()
        worst = 0.0
        for i, t in enumerate(sch.timesteps):
            s = sig[i]
            xin, e_ref = xins[i], e_refs[i]
            e_tf = model(mv(xin.float()), int(t), mv(enc), added_cond_kwargs=mvd(added), return_dict=False)[0].cpu().double()
            worst = max(worst, _rel(e_tf, e_ref))
            e_fr = model(mv((x_dev / (s * s + 1.0) ** 0.5).float()), int(t), mv(enc), added_cond_kwargs=mvd(added), return_dict=False)[0]
            x_ref = x_ref + e_ref * (sig[i + 1] - s)
            x_dev = x_dev + e_fr.cpu().double() * (sig[i + 1] - s)
        assert torch.equal(x_ref, x_ref_end)
        loop["resid_" + rd] = dict(eps_worst=worst, end_latents=_rel(x_dev, x_ref), steps=len(sch.timesteps))
        del model
    out["euler_loop_mini_xl"] = loop
    return out


def unet_cases(make_model, dev):
    out = {}
    for name, cfg, B, H, W, L in (("tiny", TINY, 2, 16, 16, 7), ("mini_xl", MINI_XL, 1, 16, 16, 77)):
        P = synth_unet_params(cfg, seed=1234)
        Ph = {k: v.to(torch.float16).float() if v.dim() > 1 else v for k, v in P.items()}   # the weights the device holds
        sample, enc, added = _inputs(cfg, B, H, W, L)
        ref = U.unet_forward(Ph, cfg, sample, 501, enc, added_cond_kwargs=added)
        ref32 = U.unet_forward(P, cfg, sample, 501, enc, added_cond_kwargs=added)
        model = make_model(cfg, P)
        mv = (lambda t: t.to(dev)) if dev else (lambda t: t)
        got = model(mv(sample), 501, mv(enc), added_cond_kwargs=None if added is None else {k: mv(v) for k, v in added.items()},
                    return_dict=False)[0].float().cpu()
        out[name] = dict(vs_oracle_same_weights=_rel(got, ref), vs_oracle_fp32_weights=_rel(got, ref32),
                         finite=bool(torch.isfinite(got).all()))
    return out


def other_models(emulate, dev):
    """SD3 MMDiT (bf16-named paths run on fp16 elements; W8A8 too), VAE decoder, CLIP and T5 text encoders: rel-L2 against
    their oracles on fp16-rounded weights"""
    from oracle import clip_ref, sd3_ref, t5_ref, vae_ref
    from paddlemix_amd.clip import CLIPTextModel, synth_clip_params
    from paddlemix_amd.sd3 import SD3Transformer2DModel, synth_sd3_params
    from paddlemix_amd.t5 import T5EncoderModel, synth_t5_params
    from paddlemix_amd.vae import AutoencoderKL, synth_decoder_params
    from tests.configs import MINI_CLIP, MINI_SD3, MINI_T5, MINI_VAE
    from tests.test_clip_host_logic import _ids
    mv = (lambda t: t.to(dev)) if dev else (lambda t: t)
    if emulate:
        from tests.abi_emulator import on_emulator as mk
    else:
        mk = lambda cls, *a, **k: cls(*a, device=dev, **k)   # noqa: E731
    h = lambda P, skip=(): {k: (v.to(torch.float16).float() if v.dim() > 1 and not any(x in k for x in skip) else v)  # noqa: E731
                            for k, v in P.items()}
    out = {}
    g = torch.Generator().manual_seed(0)
    P = synth_sd3_params(MINI_SD3, seed=1234)
    x, enc, pooled = (torch.randn(2, MINI_SD3["in_channels"], 16, 16, generator=g),
                      torch.randn(2, 10, MINI_SD3["joint_attention_dim"], generator=g),
                      torch.randn(2, MINI_SD3["pooled_projection_dim"], generator=g))
    ref = sd3_ref.sd3_forward(h(P), MINI_SD3, x, enc, pooled, 501.0)
    got = mk(SD3Transformer2DModel, MINI_SD3, P)(mv(x), mv(enc), mv(pooled), 501.0).sample.float().cpu()
    out["sd3"] = _rel(got, ref)
    P = synth_decoder_params(MINI_VAE, seed=7)
    z = torch.randn(2, MINI_VAE["latent_channels"], 8, 8, generator=g)
    out["vae"] = _rel(mk(AutoencoderKL, MINI_VAE, P).decode(mv(z)).sample.float().cpu(), vae_ref.decode(h(P), MINI_VAE, z))
    P = synth_clip_params(MINI_CLIP, seed=5)
    ids = _ids(2, 16, MINI_CLIP["vocab_size"], MINI_CLIP["eos_token_id"])
    ref = clip_ref.clip_text_forward(h(P), MINI_CLIP, ids)["last_hidden_state"]
    out["clip"] = _rel(mk(CLIPTextModel, MINI_CLIP, P)(mv(ids)).last_hidden_state.float().cpu(), ref)
    from oracle import dit_ref
    from paddlemix_amd.dit import DiTTransformer2DModel, synth_dit_params
    from tests.configs import MINI_DIT
    P = synth_dit_params(MINI_DIT, seed=21)
    xd = torch.randn(2, 4, 16, 16, generator=g)
    lab, td = torch.tensor([3, 10]), torch.tensor([999.0, 20.0])
    ref = dit_ref.dit_forward(h(P), MINI_DIT, xd, td, lab)
    out["dit"] = _rel(mk(DiTTransformer2DModel, MINI_DIT, P)(mv(xd), timestep=mv(td), class_labels=mv(lab)).sample.float().cpu(), ref)
    P = synth_t5_params(MINI_T5, seed=11)
    ids = torch.randint(0, MINI_T5["vocab_size"], (2, 24), generator=g)
    ref = t5_ref.t5_encoder_forward(h(P, skip=("relative_attention_bias",)), MINI_T5, ids)
    out["t5"] = _rel(mk(T5EncoderModel, MINI_T5, P)(mv(ids)).last_hidden_state.float().cpu(), ref)
    return out


def main(mode):
    res = dict(elem=_lib.ELEM_NAME, lib=os.path.basename(_lib.LIB_PATH))
    if mode == "cpu":
        from tests.abi_emulator import Emulator, on_emulator
        lib = _lib.load()   # dlopen works without a GPU: every declared symbol present, element type matches
        res["elem_dtype_symbol"] = lib.mi355x_sd_elem_dtype()
        res["unet"] = unet_cases(lambda cfg, P: on_emulator(UNet2DConditionModel, cfg, P), None)
        res["fp32_residual"] = fp32_residual_cases(
            lambda cfg, P, rd: on_emulator(UNet2DConditionModel, cfg, P, residual_dtype=rd), None, False)
        res["models"] = other_models(True, None)
    else:
        from paddlemix_amd import ops
        ops.init(0)
        res["elem_dtype_symbol"] = _lib.load().mi355x_sd_elem_dtype()
        g = torch.Generator().manual_seed(0)
        h = lambda t: t.to(torch.float16)   # noqa: E731
        # GEMM (bias + residual), implicit-GEMM conv, attention, GroupNorm+SiLU against fp32 math on the same fp16 operands
        a, w = h(torch.randn(1000, 1280, generator=g)), h(torch.randn(520, 1280, generator=g) / math.sqrt(1280))
        bias, r = torch.randn(520, generator=g), h(torch.randn(1000, 520, generator=g))
        got = ops.linear(a.cuda(), w.cuda(), bias.cuda(), residual=r.cuda()).float().cpu()
        res["linear"] = _rel(got, a.float() @ w.float().t() + bias + r.float())
        x = h(torch.randn(2, 24, 24, 64, generator=g))
        wc = h(torch.randn(96, 64, 3, 3, generator=g) / math.sqrt(576))
        got = ops.conv3x3(x.cuda(), wc.permute(0, 2, 3, 1).reshape(96, 576).contiguous().cuda(), None).float().cpu()
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), wc.float(), padding=1).permute(0, 2, 3, 1).reshape(-1, 96)
        res["conv3x3"] = _rel(got.reshape(-1, 96), ref)
        q, k, v = (h(torch.randn(2, 300, 5, 64, generator=g)) for _ in range(3))
        got = ops.sdpa(q.cuda(), k.cuda(), v.cuda()).float().cpu()
        ref = F.scaled_dot_product_attention(*(t.float().permute(0, 2, 1, 3) for t in (q, k, v))).permute(0, 2, 1, 3)
        res["sdpa"] = _rel(got, ref)
        # lazy row maximum in the half build: one LATE key (tile 3) scores 20-30 nats above everything the first
        # tile holds for a few queries -- exponentiated against the stale tile-0 reference that is e^20+ > 65504, which P's
        # fp16 cannot hold; the guard (LAZY_PSUM_LIMIT, attention.hip) must send those tiles down the exact path. Plain and
        # base-2 (scale folded into q) flavours.
        qs, ks_, vs = (torch.randn(2, 320, 5, 64, generator=g) for _ in range(3))
        ks_ = ks_ * 0.3
        ks_[:, 200] = 3.0 * qs[:, 17]            # score of query 17 against key 200: 3 |q|^2 / 8 ~ 24 nats
        qs, ks_, vs = h(qs), h(ks_), h(vs)
        ref = F.scaled_dot_product_attention(*(t.float().permute(0, 2, 1, 3) for t in (qs, ks_, vs))).permute(0, 2, 1, 3)
        got = ops.sdpa(qs.cuda(), ks_.cuda(), vs.cuda()).float().cpu()
        res["sdpa_late_spike"] = dict(finite=bool(torch.isfinite(got).all()), rel=_rel(got, ref),
                                      spike_nats=float((qs[:, 17].float() * ks_[:, 200].float()).sum(-1).max() / 8.0))
        q2 = h(qs.float() * (0.125 * 1.4426950408889634))    # what the UNet builder folds into to_q
        ref2 = F.scaled_dot_product_attention(*(t.float().permute(0, 2, 1, 3) for t in (q2, ks_, vs)),
                                              scale=0.6931471805599453).permute(0, 2, 1, 3)
        got2 = ops.sdpa(q2.cuda(), ks_.cuda(), vs.cuda(), log2=True).float().cpu()
        res["sdpa_late_spike_log2"] = dict(finite=bool(torch.isfinite(got2).all()), rel=_rel(got2, ref2))
        xg = h(torch.randn(2, 256, 128, generator=g) * 2 + 0.5)
        gam, bet = torch.randn(128, generator=g), torch.randn(128, generator=g)
        got = ops.group_norm(xg.cuda(), gam.cuda(), bet.cuda(), 32, 1e-5, True).float().cpu()
        ref = F.silu(F.group_norm(xg.float().permute(0, 2, 1), 32, gam, bet, 1e-5)).permute(0, 2, 1)
        res["group_norm_silu"] = _rel(got, ref)
        if mode == "gpu-kernels":   # the kernel-level checks only (seconds)
            print(json.dumps(res))
            return
        res["unet"] = unet_cases(lambda cfg, P: UNet2DConditionModel(cfg, P, device="cuda:0"), "cuda:0")
        res["fp32_residual"] = fp32_residual_cases(
            lambda cfg, P, rd: UNet2DConditionModel(cfg, P, device="cuda:0", residual_dtype=rd), "cuda:0", False)
        res["models"] = other_models(False, "cuda:0")
    print(json.dumps(res))


if __name__ == "__main__":
    main(sys.argv[1])
