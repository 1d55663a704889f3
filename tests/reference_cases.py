"""Cases that put the REFERENCE's own code beside the oracle.

Each case builds seeded inputs and parameters, runs the oracle (oracle/*.py, the torch / numpy restatement every device parity
test is checked against) and -- when asked to, which is only possible in the build container where /root/reference exists --
runs the reference's unmodified module (ppdiffusers/ppdiffusers/models/*.py, schedulers/*.py) over oracle/paddle_shim.py via
oracle/reference_runner.py. scripts/make_reference_golden.py stores the reference's outputs under
tests/golden/reference_modules/; tests/test_reference_modules.py holds the oracle to them (everywhere) and to the live reference
(here).

    case(run_reference: bool) -> {"oracle": {name: tensor}, "reference": {name: tensor} | None}
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch

from tests import configs as C

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_modules")
REL_TOL = 5e-5     # max |oracle - reference| / max |reference|: fp32 rounding of two evaluation orders, nothing structural


def _rr():
    from oracle import reference_runner
    return reference_runner


def _synth(shapes, seed):
    g = torch.Generator().manual_seed(seed)
    P = {}
    for name, shape in shapes.items():
        r = torch.randn(shape, generator=g)
        if name.endswith(".bias"):
            P[name] = 0.02 * r
        elif len(shape) == 1:
            P[name] = 1.0 + 0.05 * r
        elif len(shape) == 2:
            P[name] = r / shape[0] ** 0.5
        else:
            P[name] = r / math.sqrt(shape[1] * shape[2] * shape[3])
    return P


# ------------------------------------------------------------------------------------------------------------------ UNet
def _unet_case(cfg, *, seed=1, masks=None, class_labels=None, controlnet_residuals=False, B=2, hw=16, L=7):
    def run(ref):
        from oracle import unet_ref as U
        P = U.synth_unet_params(cfg, seed=seed)
        g = torch.Generator().manual_seed(seed + 100)
        x = torch.randn(B, cfg.get("in_channels", 4), hw, hw, generator=g)
        cd = cfg["cross_attention_dim"]
        cd = cd[0] if isinstance(cd, (tuple, list)) else cd
        enc = torch.randn(B, L, cd, generator=g)
        t = torch.tensor([10.0, 500.0])[:B]
        kw = {}
        if cfg.get("addition_embed_type") == "text_time":
            n_ids = cfg.get("_n_time_ids", 6)      # 6: SDXL base (sizes, crop, target); 5: the refiner (sizes, crop, aesthetic score)
            td = cfg["projection_class_embeddings_input_dim"] - n_ids * cfg["addition_time_embed_dim"]
            kw["added_cond_kwargs"] = dict(text_embeds=torch.randn(B, td, generator=g),
                                           time_ids=torch.tensor([[1024.0, 1024.0, 0.0, 0.0, 1024.0, 1024.0][:n_ids - 1] + [6.0 if n_ids == 5 else 1024.0]]).repeat(B, 1))
        if masks:
            if masks.get("self_len"):
                kw["attention_mask"] = (torch.rand(B, masks["self_len"], generator=g) > 0.2).float()
            kw["encoder_attention_mask"] = (torch.rand(B, L, generator=g) > 0.3).float()
        if class_labels is not None:
            kw["class_labels"] = class_labels(g)
        if controlnet_residuals:   # TINY at 16x16: the six skip tensors and the mid output
            c0, c1 = cfg["block_out_channels"]
            shapes = [(B, c0, hw, hw)] * 3 + [(B, c0, hw // 2, hw // 2)] + [(B, c1, hw // 2, hw // 2)] * 2
            kw["down_block_additional_residuals"] = tuple(0.3 * torch.randn(s, generator=g) for s in shapes)
            kw["mid_block_additional_residual"] = 0.3 * torch.randn(B, c1, hw // 2, hw // 2, generator=g)
        with torch.no_grad():
            out = {"oracle": {"sample": U.unet_forward(P, cfg, x, t, enc, **kw)}, "reference": None,
                   "inputs": dict(cfg=cfg, P=P, x=x, t=t, enc=enc, kw=kw)}
            if ref:
                rr = _rr()
                net = rr.build_unet(cfg, P)
                out["reference"] = {"sample": rr.from_shim(net(rr.to_shim(x), rr.to_shim(t), rr.to_shim(enc), **rr.to_shim(kw)).sample)}
        return out
    return run


def _ip_adapter_case(ip_scale):
    """IP-Adapter: ImageProjection + IPAdapterAttnProcessor on every cross-attention. The reference wires them in its checkpoint
    loader (loaders/unet.py:754-828, file I/O, not loaded here): the case replays that wiring with the reference's own classes."""
    def run(ref):
        from oracle import unet_ref as U
        cfg = dict(C.TINY, encoder_hid_dim_type="ip_image_proj", encoder_hid_dim=48)
        P = U.synth_unet_params(cfg, seed=6)
        g = torch.Generator().manual_seed(8)
        x, enc = torch.randn(2, 4, 16, 16, generator=g), torch.randn(2, 7, 64, generator=g)
        img, t = torch.randn(2, 48, generator=g), torch.tensor([10.0, 500.0])
        with torch.no_grad():
            out = {"oracle": {"sample": U.unet_forward(P, cfg, x, t, enc, added_cond_kwargs={"image_embeds": img}, ip_adapter_scale=ip_scale)},
                   "reference": None}
            if ref:
                rr = _rr()
                net = rr.ref_module("unet_2d_condition").UNet2DConditionModel(**C.TINY)
                ap, emb = rr.ref_module("attention_processor"), rr.ref_module("embeddings")
                procs = {}
                for name in net.attn_processors.keys():                                   # loaders/unet.py:769-797
                    if name.endswith("attn1.processor"):
                        procs[name] = ap.AttnProcessor()
                        continue
                    boc = net.config.block_out_channels
                    hidden = boc[-1] if name.startswith("mid_block") else (
                        list(reversed(boc))[int(name[len("up_blocks.")])] if name.startswith("up_blocks") else boc[int(name[len("down_blocks.")])])
                    procs[name] = ap.IPAdapterAttnProcessor(hidden_size=hidden, cross_attention_dim=net.config.cross_attention_dim, scale=1.0)
                net.set_attn_processor(procs)                                             # :799
                net.encoder_hid_proj = emb.ImageProjection(cross_attention_dim=64, image_embed_dim=48, num_image_text_embeds=4)   # :808-827
                net.config["encoder_hid_dim_type"] = "ip_image_proj"                      # :828
                for p in net.attn_processors.values():                                    # set_ip_adapter_scale (loaders/ip_adapter.py)
                    if isinstance(p, ap.IPAdapterAttnProcessor):
                        p.scale = ip_scale
                net.eval()
                rr.load_params(net, P)
                out["reference"] = {"sample": rr.from_shim(net(rr.to_shim(x), rr.to_shim(t), rr.to_shim(enc),
                                                                 added_cond_kwargs={"image_embeds": rr.to_shim(img)}).sample)}
        return out
    return run


def _lora_case(ref):
    """LoRA taken the way the reference takes it for inference -- merged into the weights (models/lora.py:312-344 conv, :404-425
    linear). "oracle" here is the PRODUCT's host-side paddlemix_amd.checkpoint.fuse_lora (plain torch on CPU); the reference side
    also runs the unfused forward (base layer + scale * LoRA branch, :364-398, :453-462), which the fused weights must reproduce."""
    from paddlemix_amd.checkpoint import fuse_lora
    g = torch.Generator().manual_seed(11)
    rn = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    params = {"lin.weight": rn(16, 24) / 4, "lin.bias": 0.1 * rn(24), "conv.weight": rn(12, 8, 3, 3) / 8, "conv.bias": 0.1 * rn(12)}
    lora = {"unet.lin.lora.down.weight": rn(16, 4) / 4, "unet.lin.lora.up.weight": rn(4, 24) / 2,
            "unet.conv.lora.down.weight": rn(4, 8, 3, 3) / 8, "unet.conv.lora.up.weight": rn(12, 4, 1, 1) / 2}
    alphas, scale = {"lin": 2.0}, 0.7
    x, img = rn(3, 16), rn(2, 8, 6, 6)
    fused = fuse_lora(params, lora, lora_scale=scale, network_alphas=alphas)
    F = torch.nn.functional
    out = {"oracle": {"lin.weight": fused["lin.weight"], "conv.weight": fused["conv.weight"],
                      "lin(x)": x @ fused["lin.weight"] + fused["lin.bias"],
                      "conv(x)": F.conv2d(img, fused["conv.weight"], fused["conv.bias"], padding=1)}, "reference": None}
    if ref:
        rr = _rr()
        m = rr.ref_module("lora")
        lin = m.LoRACompatibleLinear(16, 24)
        lin.set_lora_layer(m.LoRALinearLayer(16, 24, rank=4, network_alpha=alphas["lin"]))
        rr.load_params(lin, {"weight": params["lin.weight"], "bias": params["lin.bias"],
                             "lora_layer.down.weight": lora["unet.lin.lora.down.weight"], "lora_layer.up.weight": lora["unet.lin.lora.up.weight"]})
        conv = m.LoRACompatibleConv(8, 12, 3, padding=1)
        conv.set_lora_layer(m.LoRAConv2dLayer(8, 12, rank=4, kernel_size=3, padding=1))
        rr.load_params(conv, {"weight": params["conv.weight"], "bias": params["conv.bias"],
                              "lora_layer.down.weight": lora["unet.conv.lora.down.weight"], "lora_layer.up.weight": lora["unet.conv.lora.up.weight"]})
        unfused = {"lin(x)": rr.from_shim(lin(rr.to_shim(x), scale)), "conv(x)": rr.from_shim(conv(rr.to_shim(img), scale))}
        lin._fuse_lora(scale)
        conv._fuse_lora(scale)
        out["reference"] = {"lin.weight": rr.from_shim(lin.weight), "conv.weight": rr.from_shim(conv.weight), **unfused}
        assert torch.allclose(rr.from_shim(lin(rr.to_shim(x))), unfused["lin(x)"], atol=1e-5)     # the reference's own identity
    return out


def _controlnet_case(order, guess):
    def run(ref):
        from oracle import unet_ref as U
        cfg = dict(C.TINY, controlnet_conditioning_channel_order=order)
        P = _synth(U.controlnet_param_shapes(cfg), 3)
        g = torch.Generator().manual_seed(0)
        x, enc = torch.randn(1, 4, 8, 8, generator=g), torch.randn(1, 7, 64, generator=g)
        cond, t = torch.randn(1, 3, 64, 64, generator=g), torch.tensor([20.0])
        with torch.no_grad():
            downs, mid = U.controlnet_forward(P, cfg, x, t, enc, cond, 0.7, guess)
            out = {"oracle": {**{f"down{i}": d for i, d in enumerate(downs)}, "mid": mid}, "reference": None,
                   "inputs": dict(cfg=cfg, P=P, x=x, t=t, enc=enc, cond=cond, scale=0.7, guess=guess)}
            if ref:
                rr = _rr()
                kw = {k: v for k, v in cfg.items() if k not in ("up_block_types", "sample_size")}
                net = rr.ref_module("controlnet").ControlNetModel(**kw)
                net.eval()
                rr.load_params(net, P)
                rd, rm = net(rr.to_shim(x), rr.to_shim(t), rr.to_shim(enc), rr.to_shim(cond), conditioning_scale=0.7,
                             guess_mode=guess, return_dict=False)
                assert len(rd) == len(downs)
                out["reference"] = {**{f"down{i}": rr.from_shim(d) for i, d in enumerate(rd)}, "mid": rr.from_shim(rm)}
        return out
    return run


# ------------------------------------------------------------------------------------------------------------------ DiT / SD3 / VAE
def _dit_case(ref, side=16):
    from oracle import dit_ref as D
    cfg = C.MINI_DIT
    P = D.synth_dit_params(cfg, seed=2)
    g = torch.Generator().manual_seed(0)
    x, t, y = torch.randn(2, 4, side, side, generator=g), torch.tensor([3, 900]), torch.tensor([1, 7])
    with torch.no_grad():
        out = {"oracle": {"sample": D.dit_forward(P, cfg, x, t, y)}, "reference": None, "inputs": dict(cfg=cfg, P=P, x=x, t=t, y=y)}
        if ref:
            rr = _rr()
            full = D.normalize_config(cfg)
            full.pop("inner_dim")
            net = rr.ref_module("transformer_2d").Transformer2DModel(**full)
            net.eval()
            rr.load_params(net, P)
            out["reference"] = {"sample": rr.from_shim(net(rr.to_shim(x), timestep=rr.to_shim(t), class_labels=rr.to_shim(y)).sample)}
    return out


SD3_COMPUTED = ("pos_embed.pos_embed",)                                      # persistable buffer derived from the config
SD3_OPTIONAL = ("norm_out.norm.bias", "norm1_context.norm.bias")             # trainable, zero at construction (normalization.py:182)


def _sd3_case(trained_norm_bias, hw=(16, 16)):
    def run(ref):
        from oracle import sd3_ref as S
        cfg = C.MINI_SD3
        P = S.synth_sd3_params(cfg, seed=3)
        g = torch.Generator().manual_seed(0)
        if trained_norm_bias:
            n, D = cfg["num_layers"], cfg["num_attention_heads"] * cfg["attention_head_dim"]
            P["norm_out.norm.bias"] = 0.5 * torch.randn(D, generator=g)
            P[f"transformer_blocks.{n - 1}.norm1_context.norm.bias"] = 0.5 * torch.randn(D, generator=g)
        x, enc = torch.randn(2, 4, *hw, generator=g), torch.randn(2, 9, 64, generator=g)
        pooled, t = torch.randn(2, 64, generator=g), torch.tensor([3.0, 900.0])
        with torch.no_grad():
            out = {"oracle": {"sample": S.sd3_forward(P, cfg, x, enc, pooled, t)}, "reference": None,
                   "inputs": dict(cfg=cfg, P=P, x=x, enc=enc, pooled=pooled, t=t)}
            if ref:
                rr = _rr()
                net = rr.ref_module("transformer_sd3").SD3Transformer2DModel(**cfg)
                net.eval()
                rr.load_params(net, P, computed=SD3_COMPUTED + (() if trained_norm_bias else SD3_OPTIONAL))
                # the buffer the reference derives is the table the oracle crops from
                full = S.normalize_config(cfg)
                table = rr.from_shim(net.pos_embed.pos_embed)
                mx = cfg["pos_embed_max_size"]
                own = S.cropped_pos_embed(dict(full, sample_size=cfg["sample_size"]), mx * cfg["patch_size"], mx * cfg["patch_size"])
                assert torch.allclose(table.reshape(own.shape), own.float(), atol=1e-6)
                out["reference"] = {"sample": rr.from_shim(net(rr.to_shim(x), encoder_hidden_states=rr.to_shim(enc),
                                                                 pooled_projections=rr.to_shim(pooled), timestep=rr.to_shim(t)).sample)}
        return out
    return run


def _vae_case(ref, zhw=(8, 8), ihw=(32, 32)):
    from oracle import vae_ref as V
    cfg = C.MINI_VAE
    P = V.synth_decoder_params(cfg, seed=2)
    P.update(_synth(V.encoder_param_shapes(cfg), 7))
    g = torch.Generator().manual_seed(0)
    z, img = torch.randn(2, 4, *zhw, generator=g), torch.randn(2, 3, *ihw, generator=g)
    with torch.no_grad():
        mean, logvar, _ = V.encode(P, cfg, img)
        out = {"oracle": {"decode": V.decode(P, cfg, z), "encode_mean": mean, "encode_logvar": logvar}, "reference": None,
               "inputs": dict(cfg=cfg, P=P, z=z, img=img)}
        if ref:
            rr = _rr()
            full = V.normalize_config(cfg)
            n = len(full["block_out_channels"])
            full.update(down_block_types=("DownEncoderBlock2D",) * n, up_block_types=("UpDecoderBlock2D",) * n)
            full.pop("use_post_quant_conv", None)
            full.pop("use_quant_conv", None)
            net = rr.ref_module("autoencoder_kl").AutoencoderKL(**full)
            net.eval()
            rr.load_params(net, P)
            post = net.encode(rr.to_shim(img)).latent_dist
            out["reference"] = {"decode": rr.from_shim(net.decode(rr.to_shim(z)).sample), "encode_mean": rr.from_shim(post.mean),
                                "encode_logvar": rr.from_shim(post.logvar)}
    return out


# ------------------------------------------------------------------------------------------------------------------ text / image encoders
def _transformers_module(rr, name):
    import importlib
    rr.install()
    return importlib.import_module("ppdiffusers.transformers." + name)


def _clip_text_case(act, eos=2):
    def run(ref):
        from oracle import clip_ref as K
        cfg = dict(C.MINI_CLIP, with_projection=True, hidden_act=act, eos_token_id=eos)
        P = K.synth_clip_params(cfg, seed=2)
        g = torch.Generator().manual_seed(0)
        ids = torch.randint(3, 1000, (2, 77), generator=g)
        ids[0, 20], ids[0, 21:], ids[1, 76] = 2, 0, 2         # EOS (the largest id) mid-sequence + padding, and at the end
        if eos != 2:      # configs after the eos fix: pooled = FIRST position holding eos_token_id, not the arg-max id (modeling.py:808-822)
            ids[ids == eos] = eos + 1
            ids[0, 15], ids[0, 30], ids[1, 40] = eos, eos, eos
        with torch.no_grad():
            o = K.clip_text_forward(P, cfg, ids)
            out = {"oracle": {"last_hidden_state": o["last_hidden_state"], "text_embeds": o["text_embeds"],
                              "penultimate": o["hidden_states"][-2]}, "reference": None, "inputs": dict(cfg=cfg, P=P, ids=ids)}
            if ref:
                rr = _rr()
                conf = _transformers_module(rr, "clip.configuration").CLIPTextConfig(**{k: v for k, v in cfg.items() if k != "with_projection"})
                net = _transformers_module(rr, "clip.modeling").CLIPTextModelWithProjection(conf)
                net.eval()
                rr.load_params(net, P)
                r = net(rr.to_shim(ids), output_hidden_states=True)
                out["reference"] = {"last_hidden_state": rr.from_shim(r.last_hidden_state), "text_embeds": rr.from_shim(r.text_embeds),
                                    "penultimate": rr.from_shim(r.hidden_states[-2])}
        return out
    return run


def _clip_vision_case(ref):
    from oracle import clip_ref as K
    cfg = C.MINI_CLIP_VISION
    g = torch.Generator().manual_seed(5)
    P = {}
    for k, shp in K.clip_vision_param_shapes(cfg).items():
        r = torch.randn(shp, generator=g)
        if k.endswith(".bias"):
            P[k] = 0.02 * r
        elif "embedding" in k and len(shp) <= 2:
            P[k] = 0.5 * r
        elif len(shp) == 1:
            P[k] = 1.0 + 0.02 * r
        else:
            P[k] = r / math.sqrt(shp[0] if len(shp) == 2 else shp[1] * shp[2] * shp[3])
    px = torch.randn(2, 3, 56, 56, generator=g)
    with torch.no_grad():
        o = K.clip_vision_forward(P, cfg, px)
        out = {"oracle": {"image_embeds": o["image_embeds"], "penultimate": o["hidden_states"][-2]}, "reference": None,
               "inputs": dict(cfg=cfg, P=P, px=px)}
        if ref:
            rr = _rr()
            conf = _transformers_module(rr, "clip.configuration").CLIPVisionConfig(**cfg)
            net = _transformers_module(rr, "clip.modeling").CLIPVisionModelWithProjection(conf)
            net.eval()
            rr.load_params(net, P)
            r = net(rr.to_shim(px), output_hidden_states=True)
            out["reference"] = {"image_embeds": rr.from_shim(r.image_embeds), "penultimate": rr.from_shim(r.hidden_states[-2])}
    return out


def _t5_case(ref):
    from oracle import t5_ref as T
    cfg = C.MINI_T5
    P = T.synth_t5_params(cfg, seed=6)
    ids = torch.randint(0, 500, (2, 33), generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        out = {"oracle": {"last_hidden_state": T.t5_encoder_forward(P, cfg, ids)}, "reference": None, "inputs": dict(cfg=cfg, P=P, ids=ids)}
        if ref:
            rr = _rr()
            conf = _transformers_module(rr, "t5.configuration").T5Config(**T.normalize_config(cfg))
            net = _transformers_module(rr, "t5.modeling").T5EncoderModel(conf)
            net.eval()
            rr.load_params(net, P)          # `encoder.embed_tokens.weight` is `shared.weight` (tied): one entry, like Paddle's state dict
            out["reference"] = {"last_hidden_state": rr.from_shim(net(rr.to_shim(ids))[0])}
    return out


# ------------------------------------------------------------------------------------------------------------------ schedulers
_SD = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear")


def _scheduler_case(module, cls, oracle_cls, kw, steps, scale=True, noisy=False):
    """full sampling loop around a deterministic stand-in for the model; the final latents and the timestep table are compared"""
    def run(ref):
        from oracle import schedulers_ref as S
        g = torch.Generator().manual_seed(0)
        x0, pat = torch.randn(1, 4, 8, 8, generator=g), torch.randn(1, 4, 8, 8, generator=g)
        draws = [torch.randn(1, 4, 8, 8, generator=g) for _ in range(steps)]

        def model(x, t):
            return 0.3 * x * math.cos(0.01 * float(t)) + 0.1 * pat

        sch = getattr(S, oracle_cls)(**kw)
        sch.set_timesteps(steps)
        x = (x0 * float(getattr(sch, "init_noise_sigma", 1.0))).numpy()
        for i, t in enumerate(sch.timesteps):
            xin = sch.scale_model_input(x, t) if scale else x
            eps = model(torch.from_numpy(np.asarray(xin, dtype=np.float32)), t).numpy()
            if noisy:
                x = sch.step(eps, t, x, noise=None if i == steps - 1 else draws[i].numpy())[0]
            else:
                x = sch.step(eps, t, x)
        out = {"oracle": {"latents": torch.from_numpy(np.asarray(x, dtype=np.float32)),
                          "timesteps": torch.from_numpy(np.asarray(sch.timesteps, dtype=np.float32))}, "reference": None}
        if ref:
            rr = _rr()
            rs = getattr(rr.ref_module(module, "schedulers"), cls)(**kw)
            rs.set_timesteps(steps)
            ins = getattr(rs, "init_noise_sigma", 1.0)
            xr = rr.to_shim(x0 * float(ins if isinstance(ins, (int, float)) else rr.from_shim(ins)))
            for i, t in enumerate(rs.timesteps):
                xin = rs.scale_model_input(xr, t) if scale else xr
                eps = rr.to_shim(model(rr.from_shim(xin), rr.from_shim(t)))
                if noisy:
                    xr = rs.step(eps, t, xr, generator=lambda shape, i=i: draws[i], return_dict=False)[0]
                else:
                    xr = rs.step(eps, t, xr, return_dict=False)[0]
            out["reference"] = {"latents": rr.from_shim(xr).float(),
                                "timesteps": torch.tensor([float(rr.from_shim(t)) for t in rs.timesteps])}
        return out
    return run


# ------------------------------------------------------------------------------------------------------------------ pipelines (the caller)
class _FakeVAE:     # with output_type="latent" the pipelines read only these two config entries
    def __init__(self, rr, **extra):
        self.config = rr.FrozenConfig(block_out_channels=(1, 1, 1, 1), **extra)


def _pipe_sd_case(ref):
    """StableDiffusionPipeline.__call__ (pipelines/stable_diffusion/pipeline_stable_diffusion.py:647-932) given prompt embeddings and
    start latents: CFG with [negative, positive] batching, guidance_rescale, DDIM. The oracle side is the loop of tests/test_pipeline.py."""
    from oracle import schedulers_ref as S, unet_ref as U
    cfg = C.TINY
    P = U.synth_unet_params(cfg, seed=1)
    g = torch.Generator().manual_seed(0)
    pe, ne, lat0 = torch.randn(1, 7, 64, generator=g), torch.randn(1, 7, 64, generator=g), torch.randn(1, 4, 8, 8, generator=g)
    steps, gs, gr = 6, 7.5, 0.7
    kw = dict(_SD, clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    sch = S.DDIMRef(**kw)
    sch.set_timesteps(steps)
    x = lat0.numpy() * sch.init_noise_sigma
    with torch.no_grad():
        for t in sch.timesteps:
            eps = U.unet_forward(P, cfg, torch.from_numpy(np.concatenate([x, x])), int(t), torch.cat([ne, pe]))
            eu, et = eps[:1], eps[1:]
            e = eu + gs * (et - eu)
            e = gr * (e * (et.std(dim=(1, 2, 3), keepdim=True) / e.std(dim=(1, 2, 3), keepdim=True))) + (1 - gr) * e   # rescale_noise_cfg (:69-80)
            x = sch.step(e.numpy(), t, x)
    out = {"oracle": {"latents": torch.from_numpy(x)}, "reference": None}
    if ref:
        rr = _rr()
        pm = rr.ref_pipeline("pipeline_stable_diffusion")
        pipe = pm.StableDiffusionPipeline(vae=_FakeVAE(rr, scaling_factor=0.18215), text_encoder=None, tokenizer=None, unet=rr.build_unet(cfg, P),
                                          scheduler=rr.ref_module("scheduling_ddim", "schedulers").DDIMScheduler(**kw), safety_checker=None,
                                          feature_extractor=None, requires_safety_checker=False)
        with torch.no_grad():
            r = pipe(prompt_embeds=rr.to_shim(pe), negative_prompt_embeds=rr.to_shim(ne), latents=rr.to_shim(lat0.clone()), num_inference_steps=steps,
                     guidance_scale=gs, guidance_rescale=gr, output_type="latent", height=64, width=64, return_dict=False)[0]
        out["reference"] = {"latents": rr.from_shim(r)}
    return out


def _pipe_sdxl_case(ref):
    """StableDiffusionXLPipeline.__call__ (pipelines/stable_diffusion_xl/pipeline_stable_diffusion_xl.py): pooled embeddings +
    micro-conditioning (_get_add_time_ids: original size, crop, target size), CFG, Euler with `leading` spacing."""
    from oracle import schedulers_ref as S, unet_ref as U
    cfg = C.MINI_XL
    P = U.synth_unet_params(cfg, seed=1)
    g = torch.Generator().manual_seed(0)
    cd = cfg["cross_attention_dim"]
    pe, ne = torch.randn(1, 9, cd, generator=g), torch.randn(1, 9, cd, generator=g)
    pp, npp, lat0 = torch.randn(1, 64, generator=g), torch.randn(1, 64, generator=g), torch.randn(1, 4, 8, 8, generator=g)
    steps, gs = 5, 5.0
    kw = dict(_SD, steps_offset=1, timestep_spacing="leading")
    sch = S.EulerRef(**kw)
    sch.set_timesteps(steps)
    x = lat0.numpy() * sch.init_noise_sigma
    added = dict(text_embeds=torch.cat([npp, pp]), time_ids=torch.tensor([[96.0, 80.0, 3.0, 5.0, 64.0, 64.0]]).repeat(2, 1))
    with torch.no_grad():
        for t in sch.timesteps:
            xin = sch.scale_model_input(np.concatenate([x, x]), t)
            eps = U.unet_forward(P, cfg, torch.from_numpy(np.asarray(xin, dtype=np.float32)), float(t), torch.cat([ne, pe]), added_cond_kwargs=added).numpy()
            x = sch.step(eps[:1] + gs * (eps[1:] - eps[:1]), t, x)
    out = {"oracle": {"latents": torch.from_numpy(np.asarray(x, dtype=np.float32))}, "reference": None}
    if ref:
        rr = _rr()
        pm = rr.ref_pipeline("pipeline_stable_diffusion_xl", "pipelines.stable_diffusion_xl")
        te2 = type("TextEncoder2", (), {"config": rr.FrozenConfig(projection_dim=64), "dtype": torch.float32})()
        pipe = pm.StableDiffusionXLPipeline(vae=_FakeVAE(rr, scaling_factor=0.13025, force_upcast=False), text_encoder=None, text_encoder_2=te2,
                                            tokenizer=None, tokenizer_2=None, unet=rr.build_unet(cfg, P),
                                            scheduler=rr.ref_module("scheduling_euler_discrete", "schedulers").EulerDiscreteScheduler(**kw))
        with torch.no_grad():
            r = pipe(prompt_embeds=rr.to_shim(pe), negative_prompt_embeds=rr.to_shim(ne), pooled_prompt_embeds=rr.to_shim(pp),
                     negative_pooled_prompt_embeds=rr.to_shim(npp), latents=rr.to_shim(lat0.clone()), num_inference_steps=steps, guidance_scale=gs,
                     output_type="latent", height=64, width=64, original_size=(96, 80), crops_coords_top_left=(3, 5), target_size=(64, 64),
                     return_dict=False)[0]
        out["reference"] = {"latents": rr.from_shim(r)}
    return out


def _pipe_sd3_case(ref):
    """StableDiffusion3Pipeline.__call__ (pipelines/stable_diffusion_3/pipeline_stable_diffusion_3.py:795-860): CFG on the MMDiT's
    velocity, per-sample timestep vector, flow-matching Euler (shift 3)."""
    from oracle import schedulers_ref as S, sd3_ref as R3
    cfg = C.MINI_SD3
    P = R3.synth_sd3_params(cfg, seed=3)
    g = torch.Generator().manual_seed(0)
    pe, ne = torch.randn(1, 9, 64, generator=g), torch.randn(1, 9, 64, generator=g)
    pp, npp, lat0 = torch.randn(1, 64, generator=g), torch.randn(1, 64, generator=g), torch.randn(1, 4, 16, 16, generator=g)
    steps, gs = 6, 7.0
    sch = S.FlowMatchEulerRef(shift=3.0)
    sch.set_timesteps(steps)
    x = lat0.numpy()
    with torch.no_grad():
        for t in sch.timesteps:
            v = R3.sd3_forward(P, cfg, torch.from_numpy(np.concatenate([x, x])), torch.cat([ne, pe]), torch.cat([npp, pp]),
                               torch.tensor([float(t)] * 2)).numpy()
            x = sch.step(v[:1] + gs * (v[1:] - v[:1]), t, x)
    out = {"oracle": {"latents": torch.from_numpy(x)}, "reference": None}
    if ref:
        rr = _rr()
        pm = rr.ref_pipeline("pipeline_stable_diffusion_3", "pipelines.stable_diffusion_3")
        net = rr.ref_module("transformer_sd3").SD3Transformer2DModel(**cfg)
        net.eval()
        rr.load_params(net, P, computed=SD3_COMPUTED + SD3_OPTIONAL)
        sched = rr.ref_module("scheduling_flow_match_euler_discrete", "schedulers").FlowMatchEulerDiscreteScheduler(shift=3.0)
        pipe = pm.StableDiffusion3Pipeline(transformer=net, scheduler=sched, vae=_FakeVAE(rr, scaling_factor=1.5305, shift_factor=0.0609), text_encoder=None,
                                           tokenizer=None, text_encoder_2=None, tokenizer_2=None, text_encoder_3=None, tokenizer_3=None)
        with torch.no_grad():
            r = pipe(prompt_embeds=rr.to_shim(pe), negative_prompt_embeds=rr.to_shim(ne), pooled_prompt_embeds=rr.to_shim(pp),
                     negative_pooled_prompt_embeds=rr.to_shim(npp), latents=rr.to_shim(lat0.clone()), num_inference_steps=steps, guidance_scale=gs,
                     output_type="latent", height=128, width=128, return_dict=False)[0]
        out["reference"] = {"latents": rr.from_shim(r)}
    return out


def _ref_vae(rr, Pv):
    from oracle import vae_ref as V
    full = V.normalize_config(C.MINI_VAE)
    n = len(full["block_out_channels"])
    full.update(down_block_types=("DownEncoderBlock2D",) * n, up_block_types=("UpDecoderBlock2D",) * n)
    full.pop("use_post_quant_conv", None)
    full.pop("use_quant_conv", None)
    vae = rr.ref_module("autoencoder_kl").AutoencoderKL(**full)
    vae.eval()
    return rr.load_params(vae, Pv)


def _sd_parts(rr, pipeline_module, cls, cfg, P, sched_module, sched_cls, sched_kw, vae=None, **extra):
    pm = rr.ref_pipeline(pipeline_module, extra.pop("package", "pipelines.stable_diffusion"))
    sched = getattr(rr.ref_module(sched_module, "schedulers"), sched_cls)(**sched_kw)
    return getattr(pm, cls)(vae=vae or _FakeVAE(rr, scaling_factor=0.18215), text_encoder=None, tokenizer=None, unet=rr.build_unet(cfg, P),
                            scheduler=sched, safety_checker=None, feature_extractor=None, requires_safety_checker=False, **extra)


def _vae_params(seed):
    from oracle import vae_ref as V
    P = V.synth_decoder_params(C.MINI_VAE, seed=seed)
    P.update(_synth(V.encoder_param_shapes(C.MINI_VAE), seed + 1))
    return P


def _pipe_img2img_case(kind):
    """StableDiffusionImg2ImgPipeline.__call__ (pipeline_stable_diffusion_img2img.py): encode -> posterior sample * scaling_factor
    -> add_noise at the first kept timestep -> the last int(steps * strength) steps. Also fixes the ORDER of the random draws
    (posterior noise [1,4,h,w], then forward-process noise [B,4,h,w]) that paddlemix_amd/pipeline.py assumes."""
    def run(ref):
        from oracle import schedulers_ref as S, unet_ref as U, vae_ref as V
        cfg = C.TINY
        P, Pv = U.synth_unet_params(cfg, seed=1), _vae_params(6)
        if kind == "ddim":
            kw, ocls, smod, scls = dict(_SD, clip_sample=False, set_alpha_to_one=False, steps_offset=1), S.DDIMRef, "scheduling_ddim", "DDIMScheduler"
        else:
            kw, ocls, smod, scls = dict(_SD, steps_offset=1, timestep_spacing="leading"), S.EulerRef, "scheduling_euler_discrete", "EulerDiscreteScheduler"
        g = torch.Generator().manual_seed(3)
        pe, ne = torch.randn(2, 7, 64, generator=g), torch.randn(2, 7, 64, generator=g)
        image = torch.rand(1, 3, 32, 32, generator=g) * 2 - 1
        steps, strength, gs, sf = 10, 0.6, 5.0, C.MINI_VAE["scaling_factor"]
        gg = torch.Generator().manual_seed(11)
        n1, n2 = torch.randn(1, 4, 8, 8, generator=gg), torch.randn(2, 4, 8, 8, generator=gg)
        with torch.no_grad():
            z = V.encode(Pv, C.MINI_VAE, image, n1)[2]
            sch = ocls(**kw)
            sch.set_timesteps(steps)
            kept = sch.timesteps[steps - int(steps * strength):]
            x = sch.add_noise(torch.cat([z * sf] * 2).numpy(), n2.numpy(), kept[0] if ocls is S.DDIMRef else np.repeat(kept[:1], 2))
            for t in kept:
                xin = sch.scale_model_input(np.concatenate([x, x]), t)
                eps = U.unet_forward(P, cfg, torch.from_numpy(np.asarray(xin, dtype=np.float32)), float(t), torch.cat([ne, pe])).numpy()
                x = sch.step(eps[:2] + gs * (eps[2:] - eps[:2]), t, x)
        out = {"oracle": {"latents": torch.from_numpy(np.asarray(x, dtype=np.float32))}, "reference": None}
        if ref:
            rr = _rr()
            pipe = _sd_parts(rr, "pipeline_stable_diffusion_img2img", "StableDiffusionImg2ImgPipeline", cfg, P, smod, scls, kw, vae=_ref_vae(rr, Pv))
            gg = torch.Generator().manual_seed(11)
            shapes = []
            with torch.no_grad():
                r = pipe(prompt_embeds=rr.to_shim(pe), negative_prompt_embeds=rr.to_shim(ne), image=rr.to_shim(image), strength=strength,
                         num_inference_steps=steps, guidance_scale=gs, output_type="latent", return_dict=False,
                         generator=lambda shape: (shapes.append(list(shape)), torch.randn(shape, generator=gg))[1])[0]
            assert shapes[:2] == [[1, 4, 8, 8], [2, 4, 8, 8]], shapes
            out["reference"] = {"latents": rr.from_shim(r)}
        return out
    return run


def _pipe_inpaint_case(nine):
    """StableDiffusionInpaintPipeline.__call__ (pipeline_stable_diffusion_inpaint.py): a 4-channel UNet has the kept region re-imposed
    after every step from the re-noised image latents; a 9-channel UNet is fed [latents | mask | masked-image latents]. Mask
    binarisation / resize by the reference's real VaeImageProcessor; draw order image posterior, noise, masked-image posterior."""
    def run(ref):
        import torch.nn.functional as F
        from oracle import schedulers_ref as S, unet_ref as U, vae_ref as V
        Pv, sf = _vae_params(6), C.MINI_VAE["scaling_factor"]
        g = torch.Generator().manual_seed(8)
        pe, ne = torch.randn(1, 7, 64, generator=g), torch.randn(1, 7, 64, generator=g)
        image = torch.rand(1, 3, 32, 32, generator=g) * 2 - 1
        mask_px = torch.zeros(1, 1, 32, 32)
        mask_px[:, :, 8:24, 12:32] = 0.9
        mk = F.interpolate((mask_px >= 0.5).float(), size=(8, 8)).numpy()
        enc = lambda img, n: V.encode(Pv, C.MINI_VAE, img, n)[2] * sf  # noqa: E731
        if not nine:
            cfg, seed, steps, gs, strength = C.TINY, 31, 5, 4.0, 1.0
            P = U.synth_unet_params(cfg, seed=1)
            kw, smod, scls = dict(_SD, clip_sample=False, set_alpha_to_one=False, steps_offset=1), "scheduling_ddim", "DDIMScheduler"
        else:
            cfg, seed, steps, gs, strength = dict(C.TINY, in_channels=9), 32, 10, 1.0, 0.6
            P = U.synth_unet_params(cfg, seed=77)
            kw, smod, scls = dict(_SD, steps_offset=1, timestep_spacing="leading"), "scheduling_euler_discrete", "EulerDiscreteScheduler"
        gg = torch.Generator().manual_seed(seed)
        n_img, noise, n_msk = (torch.randn(1, 4, 8, 8, generator=gg) for _ in range(3))
        with torch.no_grad():
            img_lat = enc(image, n_img).numpy()
            if not nine:
                sch = S.DDIMRef(**kw)
                sch.set_timesteps(steps)
                x = noise.numpy() * sch.init_noise_sigma
                for i, t in enumerate(sch.timesteps):
                    eps = U.unet_forward(P, cfg, torch.from_numpy(np.concatenate([x, x]).astype(np.float32)), int(t), torch.cat([ne, pe])).numpy()
                    x = sch.step(eps[:1] + gs * (eps[1:] - eps[:1]), t, x)
                    proper = img_lat if i == steps - 1 else sch.add_noise(img_lat, noise.numpy(), int(sch.timesteps[i + 1]))
                    x = (1 - mk) * proper + mk * x
            else:
                mil = enc(image * (mask_px < 0.5).float(), n_msk).numpy()
                sch = S.EulerRef(**kw)
                sch.set_timesteps(steps)
                kept = sch.timesteps[steps - int(steps * strength):]
                x = sch.add_noise(img_lat, noise.numpy(), kept[:1])
                for t in kept:
                    xin = np.concatenate([sch.scale_model_input(x, t), mk, mil], axis=1).astype(np.float32)
                    x = sch.step(U.unet_forward(P, cfg, torch.from_numpy(xin), float(t), pe).numpy(), t, x)
        out = {"oracle": {"latents": torch.from_numpy(np.asarray(x, dtype=np.float32))}, "reference": None}
        if ref:
            rr = _rr()
            pipe = _sd_parts(rr, "pipeline_stable_diffusion_inpaint", "StableDiffusionInpaintPipeline", cfg, P, smod, scls, kw, vae=_ref_vae(rr, Pv))
            gg = torch.Generator().manual_seed(seed)
            with torch.no_grad():
                r = pipe(prompt_embeds=rr.to_shim(pe), negative_prompt_embeds=rr.to_shim(ne) if gs > 1 else None, image=rr.to_shim(image),
                         mask_image=rr.to_shim(mask_px), strength=strength, num_inference_steps=steps, guidance_scale=gs, output_type="latent",
                         height=32, width=32, return_dict=False, generator=lambda shape: torch.randn(shape, generator=gg))[0]
            out["reference"] = {"latents": rr.from_shim(r)}
        return out
    return run


def _pipe_controlnet_case(guess, scale):
    """StableDiffusionControlNetPipeline.__call__ (pipelines/controlnet/pipeline_controlnet.py): every step the ControlNet sees the
    UNet's scaled input batch; guess mode under CFG runs it on the conditional half only, zeros for the unconditional half."""
    def run(ref):
        from oracle import schedulers_ref as S, unet_ref as U
        cfg = C.TINY
        P, Pc = U.synth_unet_params(cfg, seed=1), _synth(U.controlnet_param_shapes(cfg), 8)
        g = torch.Generator().manual_seed(0)
        pe, ne = torch.randn(1, 7, 64, generator=g), torch.randn(1, 7, 64, generator=g)
        lat0, hint = torch.randn(1, 4, 8, 8, generator=g), torch.rand(1, 3, 64, 64, generator=g)
        kw = dict(_SD, clip_sample=False, set_alpha_to_one=False, steps_offset=1)
        steps, gs = 3, 5.0
        sch = S.DDIMRef(**kw)
        sch.set_timesteps(steps)
        x, emb = lat0.numpy() * sch.init_noise_sigma, torch.cat([ne, pe])
        with torch.no_grad():
            for t in sch.timesteps:
                xin = torch.from_numpy(np.concatenate([x, x]).astype(np.float32))
                if guess:
                    d, m = U.controlnet_forward(Pc, cfg, xin[1:], int(t), pe, hint, scale, True)
                    d, m = tuple(torch.cat([torch.zeros_like(v), v]) for v in d), torch.cat([torch.zeros_like(m), m])
                else:
                    d, m = U.controlnet_forward(Pc, cfg, xin, int(t), emb, torch.cat([hint, hint]), scale, False)
                eps = U.unet_forward(P, cfg, xin, int(t), emb, down_block_additional_residuals=d, mid_block_additional_residual=m).numpy()
                x = sch.step(eps[:1] + gs * (eps[1:] - eps[:1]), t, x)
        out = {"oracle": {"latents": torch.from_numpy(np.asarray(x, dtype=np.float32))}, "reference": None}
        if ref:
            rr = _rr()
            rr.ref_pipeline("pipeline_stable_diffusion")
            cn = rr.ref_module("controlnet").ControlNetModel(**{k: v for k, v in cfg.items() if k not in ("up_block_types", "sample_size")})
            cn.eval()
            rr.load_params(cn, Pc)
            pipe = _sd_parts(rr, "pipeline_controlnet", "StableDiffusionControlNetPipeline", cfg, P, "scheduling_ddim", "DDIMScheduler", kw,
                             package="pipelines.controlnet", controlnet=cn)
            with torch.no_grad():
                r = pipe(prompt_embeds=rr.to_shim(pe), negative_prompt_embeds=rr.to_shim(ne), image=rr.to_shim(hint), latents=rr.to_shim(lat0.clone()),
                         num_inference_steps=steps, guidance_scale=gs, controlnet_conditioning_scale=scale, guess_mode=guess, output_type="latent",
                         height=64, width=64, return_dict=False)[0]
            out["reference"] = {"latents": rr.from_shim(r)}
        return out
    return run


def _pipe_lcm_case(ref):
    """Latent-consistency sampling through StableDiffusionPipeline.__call__: a guidance-distilled UNet (time_cond_proj_dim) takes
    get_guidance_scale_embedding(guidance_scale - 1) as timestep_cond instead of a doubled batch (pipeline_stable_diffusion.py:
    588-616, 846-852); LCMScheduler re-noises between its steps with the pipeline's generator."""
    from oracle import schedulers_ref as S, unet_ref as U
    cfg = dict(C.TINY, time_cond_proj_dim=32)
    P = U.synth_unet_params(cfg, seed=1)
    g = torch.Generator().manual_seed(0)
    pe, lat0 = torch.randn(2, 7, 64, generator=g), torch.randn(2, 4, 8, 8, generator=g)
    steps, gs = 4, 8.0
    f = np.exp(-np.log(10000.0) * np.arange(16) / 15)
    tc = torch.from_numpy(np.tile(np.concatenate([np.sin(7000.0 * f), np.cos(7000.0 * f)]), (2, 1)).astype(np.float32))
    sch = S.LCMRef(**_SD)
    sch.set_timesteps(steps)
    gg = torch.Generator().manual_seed(21)
    x = lat0.numpy().astype(np.float64)
    with torch.no_grad():
        for i, t in enumerate(sch.timesteps):
            eps = U.unet_forward(P, cfg, torch.from_numpy(x.astype(np.float32)), int(t), pe, timestep_cond=tc).numpy()
            x, _ = sch.step(eps, t, x, None if i == steps - 1 else torch.randn(lat0.shape, generator=gg).numpy())
    out = {"oracle": {"latents": torch.from_numpy(x.astype(np.float32))}, "reference": None}
    if ref:
        rr = _rr()
        pipe = _sd_parts(rr, "pipeline_stable_diffusion", "StableDiffusionPipeline", cfg, P, "scheduling_lcm", "LCMScheduler", dict(_SD))
        gg = torch.Generator().manual_seed(21)
        with torch.no_grad():
            r = pipe(prompt_embeds=rr.to_shim(pe), latents=rr.to_shim(lat0.clone()), num_inference_steps=steps, guidance_scale=gs, output_type="latent",
                     height=64, width=64, return_dict=False, generator=lambda shape: torch.randn(shape, generator=gg))[0]
        out["reference"] = {"latents": rr.from_shim(r)}
    return out


class _FakeTokenizer:
    """the tokenizers' vocabulary files are not part of the path: a table prompt -> ids stands in (padding="longest" answers a
    shorter row, like a real tokenizer on a short prompt, so the pipelines' truncation warning stays silent)"""
    model_max_length = 77

    def __init__(self, rr, table):
        self.rr, self.table = rr, table

    def __call__(self, prompt, padding=None, max_length=None, truncation=None, return_tensors=None, add_special_tokens=True):
        prompt = [prompt] if isinstance(prompt, str) else prompt
        ids = torch.stack([self.table[p] for p in prompt])
        if padding == "longest":
            ids = ids[:, :20]
        return type("Encoding", (), {"input_ids": self.rr.to_shim(ids)})()


def _text_encoders(rr, specs):
    """specs: (kind, config, params) -> the reference's CLIPTextModel / CLIPTextModelWithProjection / T5EncoderModel"""
    out = []
    for kind, cfg, P in specs:
        if kind == "t5":
            from oracle import t5_ref as T
            net = _transformers_module(rr, "t5.modeling").T5EncoderModel(_transformers_module(rr, "t5.configuration").T5Config(**T.normalize_config(cfg)))
        else:
            conf = _transformers_module(rr, "clip.configuration").CLIPTextConfig(**{k: v for k, v in cfg.items() if k != "with_projection"})
            net = getattr(_transformers_module(rr, "clip.modeling"), "CLIPTextModelWithProjection" if kind == "clip_proj" else "CLIPTextModel")(conf)
        net.eval()
        out.append(rr.load_params(net, P))
    return out


def encode_prompt_inputs():
    """token ids and encoder parameters shared by the encode_prompt cases and the product test that replays them"""
    from oracle import clip_ref as K, t5_ref as T
    g = torch.Generator().manual_seed(4)
    ids = {}
    for name in ("a", "b", "c"):
        t = torch.randint(3, 1000, (77,), generator=g)
        t[30], t[31:] = 2, 0
        ids[name] = t
    t5_ids = torch.randint(1, 500, (77,), generator=g)
    c1 = dict(C.MINI_CLIP, with_projection=False)
    c1p = dict(C.MINI_CLIP, with_projection=True)
    c2 = dict(C.MINI_CLIP, with_projection=True, hidden_act="gelu")
    t5 = dict(C.MINI_T5, d_model=160)
    return dict(ids=ids, t5_ids=t5_ids, c1=c1, c1p=c1p, c2=c2, t5=t5, P1=K.synth_clip_params(c1, seed=2), P1p=K.synth_clip_params(c1p, seed=2),
                P2=K.synth_clip_params(c2, seed=3), P3=T.synth_t5_params(t5, seed=6))


def _encode_prompt_sd_clip_skip_case(ref):
    """StableDiffusionPipeline.encode_prompt with clip_skip (pipeline_stable_diffusion.py:378-391): the hidden state clip_skip layers
    before the last one, passed through the text encoder's final LayerNorm."""
    import torch.nn.functional as F
    from oracle import clip_ref as K
    E = encode_prompt_inputs()
    a = E["ids"]["a"][None]
    with torch.no_grad():
        o = K.clip_text_forward(E["P1"], E["c1"], a)
        D = E["c1"]["hidden_size"]
        skip1 = F.layer_norm(o["hidden_states"][-2], (D,), E["P1"]["text_model.final_layer_norm.weight"], E["P1"]["text_model.final_layer_norm.bias"], 1e-5)
        out = {"oracle": {"prompt_embeds": o["last_hidden_state"], "prompt_embeds_clip_skip_1": skip1}, "reference": None}
        if ref:
            rr = _rr()
            pm = rr.ref_pipeline("pipeline_stable_diffusion")
            (te,) = _text_encoders(rr, [("clip", E["c1"], E["P1"])])
            unet_stub = type("U", (), {"config": rr.FrozenConfig(sample_size=8), "dtype": torch.float32})()
            pipe = pm.StableDiffusionPipeline(vae=_FakeVAE(rr), text_encoder=te, tokenizer=_FakeTokenizer(rr, E["ids"]), unet=unet_stub, scheduler=type("S", (), {"config": rr.FrozenConfig()})(),
                                              safety_checker=None, feature_extractor=None, requires_safety_checker=False)
            pe, _ = pipe.encode_prompt("a", 1, False)
            ps, _ = pipe.encode_prompt("a", 1, False, clip_skip=1)
            out["reference"] = {"prompt_embeds": rr.from_shim(pe), "prompt_embeds_clip_skip_1": rr.from_shim(ps)}
    return out


def _encode_prompt_sdxl_case(ref):
    """StableDiffusionXLPipeline.encode_prompt (pipeline_stable_diffusion_xl.py:262-460): hidden_states[-2] of both encoders side by
    side, pooled = the second encoder's projected EOS row, zeros for the empty negative prompt (force_zeros_for_empty_prompt)."""
    from oracle import clip_ref as K
    E = encode_prompt_inputs()
    a, b = E["ids"]["a"][None], E["ids"]["b"][None]
    with torch.no_grad():
        o1, o2 = K.clip_text_forward(E["P1"], E["c1"], a), K.clip_text_forward(E["P2"], E["c2"], b)
        out = {"oracle": {"prompt_embeds": torch.cat([o1["hidden_states"][-2], o2["hidden_states"][-2]], -1), "pooled": o2["text_embeds"]}, "reference": None}
        if ref:
            rr = _rr()
            pm = rr.ref_pipeline("pipeline_stable_diffusion_xl", "pipelines.stable_diffusion_xl")
            te1, te2 = _text_encoders(rr, [("clip", E["c1"], E["P1"]), ("clip_proj", E["c2"], E["P2"])])
            unet_stub = type("U", (), {"config": rr.FrozenConfig(sample_size=8), "dtype": torch.float32})()
            pipe = pm.StableDiffusionXLPipeline(vae=_FakeVAE(rr), text_encoder=te1, text_encoder_2=te2, tokenizer=_FakeTokenizer(rr, E["ids"]),
                                                tokenizer_2=_FakeTokenizer(rr, E["ids"]), unet=unet_stub, scheduler=None)
            pe, ne, pp, npp = pipe.encode_prompt(prompt="a", prompt_2="b", do_classifier_free_guidance=True)
            assert float(rr.from_shim(ne).abs().max()) == 0 and float(rr.from_shim(npp).abs().max()) == 0 and ne.shape == pe.shape
            out["reference"] = {"prompt_embeds": rr.from_shim(pe), "pooled": rr.from_shim(pp)}
    return out


def _encode_prompt_sd3_case(ref):
    """StableDiffusion3Pipeline.encode_prompt (pipeline_stable_diffusion_3.py:201-420): two projected CLIP encoders (hidden_states[-2]
    side by side, zero-padded to the T5 width) followed on the token axis by the T5 sequence; pooled = both projected EOS rows."""
    from oracle import clip_ref as K, t5_ref as T
    E = encode_prompt_inputs()
    a, b, c = E["ids"]["a"][None], E["ids"]["b"][None], E["t5_ids"][None]
    with torch.no_grad():
        o1, o2 = K.clip_text_forward(E["P1p"], E["c1p"], a), K.clip_text_forward(E["P2"], E["c2"], b)
        t5 = T.t5_encoder_forward(E["P3"], E["t5"], c)
        clip = torch.cat([o1["hidden_states"][-2], o2["hidden_states"][-2]], -1)
        clip = torch.nn.functional.pad(clip, (0, t5.shape[-1] - clip.shape[-1]))
        out = {"oracle": {"prompt_embeds": torch.cat([clip, t5], -2), "pooled": torch.cat([o1["text_embeds"], o2["text_embeds"]], -1)}, "reference": None}
        if ref:
            rr = _rr()
            pm = rr.ref_pipeline("pipeline_stable_diffusion_3", "pipelines.stable_diffusion_3")
            te1, te2, te3 = _text_encoders(rr, [("clip_proj", E["c1p"], E["P1p"]), ("clip_proj", E["c2"], E["P2"]), ("t5", E["t5"], E["P3"])])
            tr = type("T", (), {"config": rr.FrozenConfig(sample_size=8, joint_attention_dim=160)})()
            tok3 = _FakeTokenizer(rr, {"c": E["t5_ids"]})
            pipe = pm.StableDiffusion3Pipeline(transformer=tr, scheduler=None, vae=_FakeVAE(rr), text_encoder=te1, tokenizer=_FakeTokenizer(rr, E["ids"]),
                                               text_encoder_2=te2, tokenizer_2=_FakeTokenizer(rr, E["ids"]), text_encoder_3=te3, tokenizer_3=tok3)
            pe, _, pp, _ = pipe.encode_prompt(prompt="a", prompt_2="b", prompt_3="c", do_classifier_free_guidance=False)
            out["reference"] = {"prompt_embeds": rr.from_shim(pe), "pooled": rr.from_shim(pp)}
    return out


DIT_PIPE_CFG = dict(C.MINI_DIT, num_embeds_ada_norm=1000)     # the reference pipeline hard-codes the null class 1000 (pipeline_dit.py:184)


def _pipe_dit_case(ref):
    """DiTPipeline.__call__ (pipelines/dit/pipeline_dit.py:158-246): class-conditional CFG with the null class on the duplicated latent
    half, guidance on the epsilon channels only, learned-sigma channels dropped before scheduler.step. The reference returns decoded
    images only, so its vae slot holds a linear stand-in (decode(x) = 0.01 x, scaling_factor 1) that is inverted afterwards."""
    from oracle import dit_ref as D, schedulers_ref as S
    cfg = DIT_PIPE_CFG
    P = D.synth_dit_params(cfg, seed=2)
    labels, gs, steps = [3, 8], 4.0, 5
    lat0 = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(5))
    kw = dict(_SD, clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    sch = S.DDIMRef(**kw)
    sch.set_timesteps(steps)
    x = torch.cat([lat0, lat0]).numpy()
    lab = torch.tensor(labels + [1000, 1000])
    with torch.no_grad():
        for t in sch.timesteps:
            x[2:] = x[:2]
            n = D.dit_forward(P, cfg, torch.from_numpy(x), torch.full((4,), int(t)), lab).numpy()
            eps = n[:, :4]
            half = eps[2:] + gs * (eps[:2] - eps[2:])
            x = sch.step(np.concatenate([half, half]), t, x)
    out = {"oracle": {"latents": torch.from_numpy(np.asarray(x[:2], dtype=np.float32))}, "reference": None}
    if ref:
        rr = _rr()
        pm = rr.ref_pipeline("pipeline_dit", "pipelines.dit")
        full = D.normalize_config(cfg)
        full.pop("inner_dim")
        net = rr.ref_module("transformer_2d").Transformer2DModel(**full)
        net.eval()
        rr.load_params(net, P)

        class LinearVAE:
            config = rr.FrozenConfig(scaling_factor=1.0)

            def decode(self, z):
                return type("O", (), {"sample": z * 0.01})()

        pipe = pm.DiTPipeline(transformer=net, vae=LinearVAE(), scheduler=rr.ref_module("scheduling_ddim", "schedulers").DDIMScheduler(**kw))
        img = pipe(class_labels=labels, guidance_scale=gs, num_inference_steps=steps, output_type="np", return_dict=False,
                   generator=lambda shape: lat0.clone())[0]
        out["reference"] = {"latents": torch.from_numpy(((np.asarray(img, dtype=np.float64) - 0.5) * 200.0).astype(np.float32)).permute(0, 3, 1, 2).contiguous()}
    return out


def _scheduler_ddim_eta_case(ref):
    """DDIM with eta > 0 (scheduling_ddim.py:350-475): the stochastic term std_dev_t * variance_noise on top of the deterministic
    update, use_clipped_model_output, clip_sample; the noise is injected (variance_noise) so both sides see the same draws."""
    from oracle import schedulers_ref as S
    kw, steps, eta = dict(_SD, clip_sample=True, clip_sample_range=2.0, set_alpha_to_one=False, steps_offset=1), 10, 0.6
    g = torch.Generator().manual_seed(0)
    x0, pat = torch.randn(1, 4, 8, 8, generator=g), torch.randn(1, 4, 8, 8, generator=g)
    draws = [torch.randn(1, 4, 8, 8, generator=g) for _ in range(steps)]
    model = lambda x, t: 0.3 * x * math.cos(0.01 * float(t)) + 0.1 * pat  # noqa: E731
    sch = S.DDIMRef(**kw)
    sch.set_timesteps(steps)
    x = x0.numpy().copy()
    for i, t in enumerate(sch.timesteps):
        eps = model(torch.from_numpy(x), t).numpy()
        t_i = int(t)
        prev_t = t_i - sch.T // steps
        a_t = sch.alphas_cumprod[t_i]
        a_prev = sch.alphas_cumprod[prev_t] if prev_t >= 0 else sch.final_alpha_cumprod
        x_0 = np.clip((x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5, -2.0, 2.0)
        std = eta * sch._get_variance(t_i, prev_t) ** 0.5
        eps_c = (x - a_t ** 0.5 * x_0) / (1 - a_t) ** 0.5                      # use_clipped_model_output: epsilon re-derived from the clipped x0
        x = (a_prev ** 0.5 * x_0 + (1 - a_prev - std ** 2) ** 0.5 * eps_c + std * draws[i].numpy()).astype(np.float32)
    out = {"oracle": {"latents": torch.from_numpy(x)}, "reference": None}
    if ref:
        rr = _rr()
        rs = rr.ref_module("scheduling_ddim", "schedulers").DDIMScheduler(**kw)
        rs.set_timesteps(steps)
        xr = rr.to_shim(x0.clone())
        for i, t in enumerate(rs.timesteps):
            eps = rr.to_shim(model(rr.from_shim(xr), rr.from_shim(t)))
            xr = rs.step(eps, t, xr, eta=eta, use_clipped_model_output=True, variance_noise=rr.to_shim(draws[i]), return_dict=False)[0]
        out["reference"] = {"latents": rr.from_shim(xr).float()}
    return out


def _checkpoint_layout_case(ref):
    """torch-layout checkpoint -> Paddle layouts: the reference's convert_pytorch_state_dict_to_paddle (models/
    modeling_pytorch_paddle_utils.py:27-63: every nn.Linear weight transposed, embedding tables and conv kernels not) against the
    PRODUCT's paddlemix_amd.checkpoint.to_paddle_layout on a UNet that has all three kinds (linear projections, a class-embedding
    table, convs). "oracle" here is the product's host code; what is compared is a per-parameter digest of the converted tensors."""
    from oracle import unet_ref as U
    from paddlemix_amd.checkpoint import to_paddle_layout
    from paddlemix_amd.unet import unet_param_shapes
    cfg = dict(C.TINY, use_linear_projection=True, num_class_embeds=10)
    P = U.synth_unet_params(cfg, seed=5)
    shapes = unet_param_shapes(cfg)
    # a torch-layout checkpoint of this model: what torch.nn.Linear stores is [out, in]
    is_table = lambda k: k == "class_embedding.weight"  # noqa: E731
    pt = {k: (v.t().contiguous() if v.dim() == 2 and not is_table(k) else v.clone()) for k, v in P.items()}

    def digest(d):
        keys = sorted(d)
        return torch.tensor([float((torch.as_tensor(np.asarray(d[k])).double().flatten() * (torch.arange(1, torch.as_tensor(np.asarray(d[k])).numel() + 1) % 97).double()).sum())
                             for k in keys], dtype=torch.float64)

    ours = to_paddle_layout(pt, shapes, "pt")
    assert all(torch.equal(ours[k], P[k]) for k in P)                      # and the round trip is exact
    out = {"oracle": {"digest": digest(ours).float()}, "reference": None}
    if ref:
        rr = _rr()
        net = rr.ref_module("unet_2d_condition").UNet2DConditionModel(**cfg)
        conv = rr.ref_module("modeling_pytorch_paddle_utils").convert_pytorch_state_dict_to_paddle
        theirs = conv(net, {k: v.numpy().copy() for k, v in pt.items()})
        assert sorted(theirs) == sorted(P) and all(tuple(theirs[k].shape) == tuple(P[k].shape) for k in P)
        out["reference"] = {"digest": digest(theirs).float()}
    return out


def _pipe_lcm_img2img_case(ref):
    """LatentConsistencyModelImg2ImgPipeline.__call__ (pipelines/latent_consistency_models/pipeline_latent_consistency_img2img.py):
    `strength` goes INTO LCMScheduler.set_timesteps (the distillation schedule is shortened and ALL num_inference_steps of it run --
    not the SD img2img rule), the start latents are the image latents re-noised at the first timestep."""
    from oracle import schedulers_ref as S, unet_ref as U
    cfg = dict(C.TINY, time_cond_proj_dim=32)
    P = U.synth_unet_params(cfg, seed=1)
    g = torch.Generator().manual_seed(0)
    pe, img_lat = torch.randn(1, 7, 64, generator=g), torch.randn(1, 4, 8, 8, generator=g)
    steps, gs, strength = 4, 8.0, 0.5
    f = np.exp(-np.log(10000.0) * np.arange(16) / 15)
    tc = torch.from_numpy(np.concatenate([np.sin(7000.0 * f), np.cos(7000.0 * f)])[None].astype(np.float32))
    sch = S.LCMRef(**_SD)
    sch.set_timesteps(steps, strength=strength)
    assert list(sch.timesteps) == [499, 379, 259, 139]
    gg = torch.Generator().manual_seed(3)
    n0 = torch.randn(img_lat.shape, generator=gg).numpy()
    a0 = float(sch.alphas_cumprod[int(sch.timesteps[0])])
    x = (a0 ** 0.5 * img_lat.numpy() + (1 - a0) ** 0.5 * n0).astype(np.float64)
    with torch.no_grad():
        for i, t in enumerate(sch.timesteps):
            eps = U.unet_forward(P, cfg, torch.from_numpy(x.astype(np.float32)), int(t), pe, timestep_cond=tc).numpy()
            x, _ = sch.step(eps, t, x, None if i == steps - 1 else torch.randn(img_lat.shape, generator=gg).numpy())
    out = {"oracle": {"latents": torch.from_numpy(x.astype(np.float32))}, "reference": None}
    if ref:
        rr = _rr()
        rr.ref_pipeline("pipeline_stable_diffusion")
        import importlib
        pm = importlib.import_module("ppdiffusers.pipelines.latent_consistency_models.pipeline_latent_consistency_img2img")
        sched = rr.ref_module("scheduling_lcm", "schedulers").LCMScheduler(**_SD)
        pipe = pm.LatentConsistencyModelImg2ImgPipeline(vae=_FakeVAE(rr, scaling_factor=0.18215), text_encoder=None, tokenizer=None, unet=rr.build_unet(cfg, P),
                                                        scheduler=sched, safety_checker=None, feature_extractor=None, requires_safety_checker=False)
        gg = torch.Generator().manual_seed(3)
        with torch.no_grad():
            r = pipe(prompt_embeds=rr.to_shim(pe), image=rr.to_shim(img_lat), num_inference_steps=steps, strength=strength, guidance_scale=gs,
                     output_type="latent", return_dict=False, generator=lambda shape: torch.randn(shape, generator=gg))[0]
        assert [int(rr.from_shim(t)) for t in sched.timesteps] == [499, 379, 259, 139]
        out["reference"] = {"latents": rr.from_shim(r)}
    return out


def _pipe_sd_from_prompts_case(ref):
    """StableDiffusionPipeline.__call__ from prompt STRINGS (table tokenizer): encode_prompt of the prompt and of the negative prompt
    through the real CLIP text tower, [negative, positive] batching, CFG, DDIM. The cross-attention width of the UNet is the text
    tower's hidden size."""
    from oracle import clip_ref as K, schedulers_ref as S, unet_ref as U
    E = encode_prompt_inputs()
    cfg = C.TINY                                   # cross_attention_dim 64 = MINI_CLIP hidden size
    P = U.synth_unet_params(cfg, seed=1)
    lat0 = torch.randn(1, 4, 8, 8, generator=torch.Generator().manual_seed(9))
    steps, gs = 4, 6.0
    kw = dict(_SD, clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    with torch.no_grad():
        pe = K.clip_text_forward(E["P1"], E["c1"], E["ids"]["a"][None])["last_hidden_state"]
        ne = K.clip_text_forward(E["P1"], E["c1"], E["ids"]["b"][None])["last_hidden_state"]
        sch = S.DDIMRef(**kw)
        sch.set_timesteps(steps)
        x = lat0.numpy() * sch.init_noise_sigma
        for t in sch.timesteps:
            eps = U.unet_forward(P, cfg, torch.from_numpy(np.concatenate([x, x])), int(t), torch.cat([ne, pe])).numpy()
            x = sch.step(eps[:1] + gs * (eps[1:] - eps[:1]), t, x)
    out = {"oracle": {"latents": torch.from_numpy(x)}, "reference": None}
    if ref:
        rr = _rr()
        (te,) = _text_encoders(rr, [("clip", E["c1"], E["P1"])])
        pipe = _sd_parts(rr, "pipeline_stable_diffusion", "StableDiffusionPipeline", cfg, P, "scheduling_ddim", "DDIMScheduler", kw)
        pipe.text_encoder, pipe.tokenizer = te, _FakeTokenizer(rr, E["ids"])
        with torch.no_grad():
            r = pipe(prompt="a", negative_prompt="b", latents=rr.to_shim(lat0.clone()), num_inference_steps=steps, guidance_scale=gs,
                     output_type="latent", height=64, width=64, return_dict=False)[0]
        out["reference"] = {"latents": rr.from_shim(r)}
    return out


def _labels(kind):
    return {
        "index": lambda g: torch.tensor([3, 8]),
        "time": lambda g: torch.tensor([3.0, 8.0]),
        "vec256": lambda g: torch.randn(2, 256, generator=g),
        "vec24": lambda g: torch.randn(2, 24, generator=g),
    }[kind]


CASES = {
    # UNet2DConditionModel.forward (models/unet_2d_condition.py) on the tiny / SDXL-shaped configs and every configuration variant
    "unet_tiny": _unet_case(C.TINY),
    "unet_mini_xl": _unet_case(C.MINI_XL),
    **{"unet_" + k.replace("-", "_"): _unet_case(v) for k, v in C.UNET_VARIANTS.items()},
    "unet_mini_xl_refiner_5_time_ids": _unet_case(dict(C.MINI_XL, projection_class_embeddings_input_dim=32 * 5 + 64, _n_time_ids=5)),
    "unet_mini_xl_odd_size": _unet_case(C.MINI_XL, hw=18),          # 18 % 4 != 0: forward_upsample_size (unet_2d_condition.py:900-906)
    "unet_tiny_masks": _unet_case(C.TINY, masks=dict(self_len=64)),
    "unet_mini_xl_encoder_mask": _unet_case(C.MINI_XL, masks=dict()),
    "unet_class_embeds": _unet_case(dict(C.TINY, num_class_embeds=10), seed=4, class_labels=_labels("index")),
    "unet_class_timestep": _unet_case(dict(C.TINY, class_embed_type="timestep"), seed=4, class_labels=_labels("time")),
    "unet_class_identity": _unet_case(dict(C.TINY, class_embed_type="identity"), seed=4, class_labels=_labels("vec256")),
    "unet_class_projection": _unet_case(dict(C.TINY, class_embed_type="projection", projection_class_embeddings_input_dim=24), seed=4,
                                        class_labels=_labels("vec24")),
    "unet_class_simple_projection": _unet_case(dict(C.TINY, class_embed_type="simple_projection", projection_class_embeddings_input_dim=24),
                                               seed=4, class_labels=_labels("vec24")),
    "unet_class_concat": _unet_case(dict(C.TINY, class_embed_type="projection", projection_class_embeddings_input_dim=24,
                                         class_embeddings_concat=True), seed=4, class_labels=_labels("vec24")),
    "unet_controlnet_residuals": _unet_case(C.TINY, seed=4, controlnet_residuals=True),
    "unet_ip_adapter": _ip_adapter_case(1.0),
    "unet_ip_adapter_scale_0p6": _ip_adapter_case(0.6),
    "lora_fuse": _lora_case,
    "checkpoint_layout_pt_to_paddle": _checkpoint_layout_case,
    # ControlNetModel.forward (models/controlnet.py)
    "controlnet_rgb": _controlnet_case("rgb", False),
    "controlnet_bgr_guess_mode": _controlnet_case("bgr", True),
    # Transformer2DModel.forward, DiT branch (models/transformer_2d.py); SD3Transformer2DModel.forward (models/transformer_sd3.py)
    "dit_mini": _dit_case,
    "dit_mini_other_resolution": lambda ref: _dit_case(ref, side=24),     # latents larger than sample_size: the position table is rebuilt for the grid
    "sd3_mini": _sd3_case(False),
    "sd3_mini_trained_norm_bias": _sd3_case(True),
    "sd3_mini_nonsquare_8x24": _sd3_case(False, hw=(8, 24)),          # the centre crop of the position table: height / width order
    # AutoencoderKL.decode / encode (models/autoencoder_kl.py, models/vae.py)
    "vae_mini": _vae_case,
    "vae_mini_ragged": lambda ref: _vae_case(ref, zhw=(7, 8), ihw=(30, 32)),      # odd sizes through the bottom / right padded stride-2 convs
    # transformers/clip/modeling.py (text towers of SD / SDXL, the IP-Adapter image tower), transformers/t5/modeling.py (SD3's T5 encoder)
    "clip_text_quick_gelu": _clip_text_case("quick_gelu"),
    "clip_text_gelu": _clip_text_case("gelu"),
    "clip_text_eos_by_id": _clip_text_case("quick_gelu", eos=7),
    "clip_vision": _clip_vision_case,
    "t5_encoder": _t5_case,
    # the callers: pipelines/*/pipeline_*.py __call__ from prompt embeddings + start latents to final latents
    "pipe_sd_ddim_cfg_rescale": _pipe_sd_case,
    "pipe_sd_from_prompt_strings": _pipe_sd_from_prompts_case,
    "pipe_sdxl_euler_cfg_microcond": _pipe_sdxl_case,
    "pipe_sd3_flow_match_cfg": _pipe_sd3_case,
    "encode_prompt_sd_clip_skip": _encode_prompt_sd_clip_skip_case,
    "encode_prompt_sdxl": _encode_prompt_sdxl_case,
    "encode_prompt_sd3": _encode_prompt_sd3_case,
    "pipe_dit_class_cfg": _pipe_dit_case,
    "pipe_img2img_ddim": _pipe_img2img_case("ddim"),
    "pipe_img2img_euler": _pipe_img2img_case("euler"),
    "pipe_inpaint_4ch_ddim_cfg": _pipe_inpaint_case(False),
    "pipe_inpaint_9ch_euler": _pipe_inpaint_case(True),
    "pipe_controlnet": _pipe_controlnet_case(False, 0.8),
    "pipe_controlnet_guess_mode": _pipe_controlnet_case(True, 1.0),
    "pipe_lcm_timestep_cond": _pipe_lcm_case,
    "pipe_lcm_img2img_strength": _pipe_lcm_img2img_case,
    # schedulers/*.py: whole sampling loops
    "sched_ddim_sd15": _scheduler_case("scheduling_ddim", "DDIMScheduler", "DDIMRef", dict(_SD, clip_sample=False, set_alpha_to_one=False, steps_offset=1), 20),
    "sched_ddim_trailing_clip": _scheduler_case("scheduling_ddim", "DDIMScheduler", "DDIMRef", dict(_SD, clip_sample=True, timestep_spacing="trailing"), 10),
    "sched_euler_sdxl": _scheduler_case("scheduling_euler_discrete", "EulerDiscreteScheduler", "EulerRef", dict(_SD, timestep_spacing="leading", steps_offset=1), 30),
    "sched_euler_karras": _scheduler_case("scheduling_euler_discrete", "EulerDiscreteScheduler", "EulerRef", dict(_SD, use_karras_sigmas=True), 12),
    "sched_flow_match_sd3": _scheduler_case("scheduling_flow_match_euler_discrete", "FlowMatchEulerDiscreteScheduler", "FlowMatchEulerRef", dict(shift=3.0), 28, scale=False),
    "sched_pndm_sd15": _scheduler_case("scheduling_pndm", "PNDMScheduler", "PNDMRef", dict(_SD, skip_prk_steps=True, steps_offset=1), 20),
    "sched_pndm_prk": _scheduler_case("scheduling_pndm", "PNDMScheduler", "PNDMRef", dict(_SD), 10),
    "sched_dpmpp_2m": _scheduler_case("scheduling_dpmsolver_multistep", "DPMSolverMultistepScheduler", "DPMSolverMultistepRef", dict(_SD), 20),
    "sched_dpmpp_2m_karras_heun": _scheduler_case("scheduling_dpmsolver_multistep", "DPMSolverMultistepScheduler", "DPMSolverMultistepRef",
                                                  dict(_SD, use_karras_sigmas=True, solver_type="heun"), 12),
    "sched_dpm_order1_leading": _scheduler_case("scheduling_dpmsolver_multistep", "DPMSolverMultistepScheduler", "DPMSolverMultistepRef",
                                                dict(_SD, algorithm_type="dpmsolver", solver_order=1, timestep_spacing="leading", steps_offset=1), 10),
    "sched_lcm": _scheduler_case("scheduling_lcm", "LCMScheduler", "LCMRef", dict(_SD), 4, scale=False, noisy=True),
    "sched_ddim_eta_clipped": _scheduler_ddim_eta_case,
    # the other prediction types (SD-2.x checkpoints are v-prediction) and spacings
    "sched_ddim_v_prediction": _scheduler_case("scheduling_ddim", "DDIMScheduler", "DDIMRef", dict(_SD, clip_sample=False, set_alpha_to_one=False, steps_offset=1,
                                                                                                  prediction_type="v_prediction"), 12),
    "sched_ddim_sample_prediction": _scheduler_case("scheduling_ddim", "DDIMScheduler", "DDIMRef", dict(_SD, clip_sample=False, prediction_type="sample"), 8),
    "sched_euler_v_prediction_trailing": _scheduler_case("scheduling_euler_discrete", "EulerDiscreteScheduler", "EulerRef",
                                                         dict(_SD, prediction_type="v_prediction", timestep_spacing="trailing"), 10),
    "sched_pndm_v_prediction": _scheduler_case("scheduling_pndm", "PNDMScheduler", "PNDMRef", dict(_SD, skip_prk_steps=True, steps_offset=1, prediction_type="v_prediction"), 12),
    "sched_dpmpp_v_prediction": _scheduler_case("scheduling_dpmsolver_multistep", "DPMSolverMultistepScheduler", "DPMSolverMultistepRef",
                                                dict(_SD, prediction_type="v_prediction"), 12),
    "sched_dpmpp_sample_prediction_euler_final": _scheduler_case("scheduling_dpmsolver_multistep", "DPMSolverMultistepScheduler", "DPMSolverMultistepRef",
                                                                 dict(_SD, prediction_type="sample", euler_at_final=True, timestep_spacing="trailing"), 16),
}


def golden_path(name):
    return os.path.join(GOLDEN_DIR, name + ".npz")
