"""Model steps that tests/test_export_host_logic.py (CPU: the launch list through the host-memory emulator) and
tests/test_gpu_export.py (MI355X) export with paddlemix_amd.export and replay through mi355x_sd_program_*.

    build(name, backend_kw) -> (model, run, outputs)       run() executes one forward and returns nothing; the result of the
                                                           step is whatever the plan's output regions hold afterwards"""
import torch

from tests import configs as C


def _dev(t, cpu):
    if isinstance(t, dict):
        return {k: _dev(v, cpu) for k, v in t.items()}
    if isinstance(t, (list, tuple)):
        return type(t)(_dev(v, cpu) for v in t)
    return t if cpu or not torch.is_tensor(t) else t.cuda()


def build(name, cpu, **kw):
    g = torch.Generator().manual_seed(7)
    rn = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    if name in ("unet_tiny", "unet_mini_xl", "unet_tiny_masked_controlnet", "unet_class_labels", "unet_ip_adapter"):
        from paddlemix_amd.unet import UNet2DConditionModel, synth_unet_params
        cfg = {"unet_tiny": C.TINY, "unet_mini_xl": C.MINI_XL, "unet_tiny_masked_controlnet": C.TINY,
               "unet_class_labels": dict(C.TINY, num_class_embeds=10),
               "unet_ip_adapter": dict(C.TINY, encoder_hid_dim_type="ip_image_proj", encoder_hid_dim=48)}[name]
        m = UNet2DConditionModel(cfg, synth_unet_params(cfg, seed=1), **kw)
        B, L = 2, 7
        x, enc = rn(B, 4, 16, 16), rn(B, L, cfg["cross_attention_dim"])
        fkw = {}
        if cfg.get("addition_embed_type") == "text_time":
            td = cfg["projection_class_embeddings_input_dim"] - 6 * cfg["addition_time_embed_dim"]
            fkw["added_cond_kwargs"] = dict(text_embeds=rn(B, td), time_ids=torch.tensor([[1024., 1024., 0., 0., 1024., 1024.]]).repeat(B, 1))
        if name == "unet_tiny_masked_controlnet":
            em = torch.ones(B, L)
            em[:, 5:] = 0
            fkw["encoder_attention_mask"] = em
            c0, c1 = cfg["block_out_channels"]
            shapes = [(B, c0, 16, 16)] * 3 + [(B, c0, 8, 8)] + [(B, c1, 8, 8)] * 2
            fkw["down_block_additional_residuals"] = [0.3 * rn(*s) for s in shapes]
            fkw["mid_block_additional_residual"] = 0.3 * rn(B, c1, 8, 8)
        if name == "unet_class_labels":
            fkw["class_labels"] = torch.tensor([3, 8])
        if name == "unet_ip_adapter":
            fkw["added_cond_kwargs"] = dict(image_embeds=rn(B, 48))
        return m, (lambda: m(_dev(x, cpu), 501, _dev(enc, cpu), **_dev(fkw, cpu))), ("out",)
    if name == "controlnet_tiny":
        from paddlemix_amd.unet import ControlNetModel, synth_controlnet_params
        cfg = dict(C.TINY, controlnet_conditioning_channel_order="bgr")
        m = ControlNetModel(cfg, synth_controlnet_params(cfg, seed=3), **kw)
        x, enc, cond = rn(1, 4, 16, 16), rn(1, 7, 64), rn(1, 3, 128, 128)
        return m, (lambda: m(_dev(x, cpu), 20, _dev(enc, cpu), _dev(cond, cpu), conditioning_scale=0.7)), ("ctrl_out",)
    if name == "sd3_mini":
        from paddlemix_amd.sd3 import SD3Transformer2DModel, synth_sd3_params
        m = SD3Transformer2DModel(C.MINI_SD3, synth_sd3_params(C.MINI_SD3, seed=3), **kw)
        x, enc, pooled = rn(2, 4, 16, 16), rn(2, 9, 64), rn(2, 64)
        return m, (lambda: m(_dev(x, cpu), _dev(enc, cpu), _dev(pooled, cpu), 501.0)), ("out",)
    if name == "dit_mini":
        from paddlemix_amd.dit import DiTTransformer2DModel, synth_dit_params
        m = DiTTransformer2DModel(C.MINI_DIT, synth_dit_params(C.MINI_DIT, seed=2), **kw)
        x, t, y = rn(2, 4, 16, 16), torch.tensor([3, 900]), torch.tensor([1, 7])
        return m, (lambda: m(_dev(x, cpu), timestep=_dev(t, cpu), class_labels=_dev(y, cpu))), ("out",)
    if name in ("vae_decode", "vae_encode"):
        from paddlemix_amd.vae import AutoencoderKL, synth_vae_params
        m = AutoencoderKL(C.MINI_VAE, synth_vae_params(C.MINI_VAE, 9), **kw)
        if name == "vae_decode":
            z = rn(2, 4, 8, 8)
            return m, (lambda: m.decode(_dev(z, cpu))), ("out",)
        img = torch.rand(2, 3, 32, 32, generator=g) * 2 - 1
        return m, (lambda: m.encode(_dev(img, cpu))), ("mean", "logvar", "out")
    if name == "clip_text":
        from paddlemix_amd.clip import CLIPTextModelWithProjection, synth_clip_params
        cfg = C.MINI_CLIP
        m = CLIPTextModelWithProjection(cfg, synth_clip_params(dict(cfg, with_projection=True), seed=2), **kw)
        ids = torch.randint(3, 1000, (2, 77), generator=g)
        ids[:, 40] = 2
        return m, (lambda: m(_dev(ids, cpu), output_hidden_states=True)), ("out", "last", "hidden", "pooled", "embeds", "text_embeds")
    if name == "clip_vision":
        from paddlemix_amd.clip import CLIPVisionModelWithProjection, synth_clip_vision_params
        m = CLIPVisionModelWithProjection(C.MINI_CLIP_VISION, synth_clip_vision_params(C.MINI_CLIP_VISION, seed=5), **kw)
        px = rn(2, 3, 56, 56)
        return m, (lambda: m(_dev(px, cpu), output_hidden_states=True)), ("out", "embeds", "hidden", "pooled")
    if name == "t5_encoder":
        from paddlemix_amd.t5 import T5EncoderModel, synth_t5_params
        m = T5EncoderModel(C.MINI_T5, synth_t5_params(C.MINI_T5, seed=6), **kw)
        ids = torch.randint(0, 500, (2, 33), generator=g)
        return m, (lambda: m(_dev(ids, cpu))), ("out", "last")
    raise KeyError(name)


NAMES = ["unet_tiny", "unet_mini_xl", "unet_tiny_masked_controlnet", "unet_class_labels", "unet_ip_adapter", "controlnet_tiny", "sd3_mini",
         "dit_mini", "vae_decode", "vae_encode", "clip_text", "clip_vision", "t5_encoder"]


def last_plan(model):
    return list(model._plans.values())[-1]
