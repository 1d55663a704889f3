"""The denoising loop of ``StableDiffusionPipeline.__call__`` / ``StableDiffusionXLPipeline.__call__`` with the
MI355X UNet in the ``unet`` slot.

Mirrors ppdiffusers/ppdiffusers/pipelines/stable_diffusion/pipeline_stable_diffusion.py:813-908 (SDXL:
stable_diffusion_xl/pipeline_stable_diffusion_xl.py:1039-1093): classifier-free-guidance batch doubling,
``scheduler.scale_model_input``, the UNet call, the guidance combine (+ optional ``rescale_noise_cfg`` :69-80),
``scheduler.step`` and ``callback_on_step_end``, then -- when a ``vae`` is given and ``output_type`` is not "latent" --
``vae.decode(latents / scaling_factor)`` and the image processor's denormalise (:911-925,
image_processor.py VaeImageProcessor.postprocess).  Prompt encoding (CLIP / T5) is a "next row" of SURVEY.md 8f and
stays outside: the loop starts from ``prompt_embeds``.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import numpy as np
import torch


def rescale_noise_cfg(noise_cfg, noise_pred_text, guidance_rescale=0.0):
    dims = list(range(1, noise_pred_text.ndim))
    std_text = noise_pred_text.std(dim=dims, keepdim=True)
    std_cfg = noise_cfg.std(dim=dims, keepdim=True)
    rescaled = noise_cfg * (std_text / std_cfg)
    return guidance_rescale * rescaled + (1 - guidance_rescale) * noise_cfg


class StableDiffusionDenoiser:
    """``pipe = StableDiffusionDenoiser(unet, scheduler); latents = pipe(prompt_embeds=..., ...)``"""

    def __init__(self, unet, scheduler, vae=None, text_encoder=None, text_encoder_2=None, controlnet=None):
        self.unet, self.scheduler, self.vae = unet, scheduler, vae
        self.text_encoder, self.text_encoder_2 = text_encoder, text_encoder_2
        self.controlnet = controlnet   # ControlNetModel: StableDiffusionControlNetPipeline (controlnet/pipeline_controlnet.py)

    def encode_prompt(self, input_ids: torch.Tensor, input_ids_2: Optional[torch.Tensor] = None,
                      clip_skip: Optional[int] = None):
        """Token ids -> (prompt_embeds, pooled_prompt_embeds or None). Tokenisation stays with the caller (the CLIP
        vocabulary files are not part of a model's weights).
        SD (one encoder, pipeline_stable_diffusion.py:368-391): ``text_encoder(ids)[0]``.
        SDXL (two encoders, pipeline_stable_diffusion_xl.py:363-375): hidden_states[-2] of both encoders concatenated on
        the channel axis, pooled = ``text_encoder_2(ids)[0]`` (the projected EOS row)."""
        if self.text_encoder is None:
            raise ValueError("encode_prompt needs a `text_encoder`")
        if self.text_encoder_2 is None:
            if clip_skip is None:
                return self.text_encoder(input_ids)[0], None
            # :378-391: the hidden state clip_skip layers before the last, then the encoder's final LayerNorm
            hidden = self.text_encoder(input_ids, output_hidden_states=True).hidden_states[-(clip_skip + 1)]
            return self.text_encoder.text_model.final_layer_norm(hidden), None
        embeds, pooled = [], None
        for enc, ids in ((self.text_encoder, input_ids), (self.text_encoder_2, input_ids if input_ids_2 is None else input_ids_2)):
            out = enc(ids, output_hidden_states=True)
            pooled = out[0]   # only the final encoder's pooled output is kept
            embeds.append(out.hidden_states[-2 if clip_skip is None else -(clip_skip + 2)])
        return torch.cat(embeds, dim=-1), pooled

    def get_add_time_ids(self, original_size, crops_coords_top_left, target_size, text_encoder_projection_dim: int,
                         device=None) -> torch.Tensor:
        """pipeline_stable_diffusion_xl.py:603-619"""
        ids = list(original_size + crops_coords_top_left + target_size)
        cfg = self.unet.config
        passed = cfg.addition_time_embed_dim * len(ids) + text_encoder_projection_dim
        if passed != cfg.projection_class_embeddings_input_dim:
            raise ValueError(f"Model expects an added time embedding vector of length "
                             f"{cfg.projection_class_embeddings_input_dim}, but a vector of {passed} was created.")
        return torch.tensor([ids], dtype=torch.float32, device=device)

    def _fused_plan(self, guidance_rescale: float, device):
        """(per-step input scales, device table of (a, b), library, stream getter) when the scheduler's step is the
        linear epsilon update ``prev = a * x + b * eps`` (``step_coefficients``); None -> the generic torch path."""
        sch = self.scheduler
        if guidance_rescale > 0.0 or not hasattr(sch, "step_coefficients") or not hasattr(self.unet, "_lib"):
            return None
        try:
            scales, coefs = [], []
            for t in sch.timesteps:
                scales.append(float(sch.model_input_scale(t)) if hasattr(sch, "model_input_scale") else 1.0)
                coefs.append(tuple(float(v) for v in sch.step_coefficients(t)))   # Euler: advances the step index
        except NotImplementedError:
            return None
        finally:
            if hasattr(sch, "_step_index"):
                sch._step_index = None
        coef = torch.tensor(coefs, dtype=torch.float32, device=device).contiguous()
        if getattr(self.unet, "_emulated", False):
            stream = lambda: 0  # noqa: E731
        else:
            stream = lambda: torch.cuda.current_stream(device).cuda_stream  # noqa: E731
        return scales, coef, self.unet._lib, stream

    def decode_latents(self, latents: torch.Tensor, output_type: str = "pt"):
        """pipeline_stable_diffusion.py:911 + VaeImageProcessor.postprocess: decode, (x / 2 + 0.5).clamp(0, 1)."""
        if self.vae is None:
            raise ValueError("output_type != 'latent' needs a `vae`")
        if output_type not in ("pt", "np"):
            raise ValueError(f"output_type must be 'latent', 'pt' or 'np', got {output_type!r}")
        vc = self.vae.config
        mean, std = getattr(vc, "latents_mean", None), getattr(vc, "latents_std", None)
        if mean is not None and std is not None:
            # StableDiffusionXLPipeline (pipeline_stable_diffusion_xl.py:1105-1110): VAEs that publish per-channel latent statistics
            # are denormalised with them instead of the plain 1 / scaling_factor
            shape = (1, -1, 1, 1)
            latents = latents * torch.as_tensor(std, dtype=latents.dtype, device=latents.device).reshape(shape) / vc.scaling_factor \
                + torch.as_tensor(mean, dtype=latents.dtype, device=latents.device).reshape(shape)
            image = self.vae.decode(latents, return_dict=False)[0]
        else:
            image = self.vae.decode(latents, return_dict=False, in_scale=1.0 / vc.scaling_factor)[0]
        image = (image / 2 + 0.5).clamp(0, 1)
        return image if output_type == "pt" else image.cpu().permute(0, 2, 3, 1).float().numpy()

    def prepare_latents(self, batch_size, num_channels_latents, height, width, dtype, generator=None, latents=None,
                        device=None):
        shape = (batch_size, num_channels_latents, height, width)
        if latents is None:
            latents = torch.randn(shape, generator=generator, dtype=dtype, device=device)
        elif tuple(latents.shape) != shape:
            raise ValueError(f"Unexpected latents shape, got {tuple(latents.shape)}, expected {shape}")
        return latents * self.scheduler.init_noise_sigma  # pipeline_stable_diffusion.py:581-586

    @staticmethod
    def get_guidance_scale_embedding(w: torch.Tensor, embedding_dim: int = 512) -> torch.Tensor:
        """sinusoidal embedding of (guidance_scale - 1) for guidance-distilled UNets (LCM; pipeline_stable_diffusion.py:588-616)
        -> fp32 [len(w), embedding_dim], the UNet's ``timestep_cond``"""
        if w.dim() != 1:
            raise ValueError("w: expected a 1-D tensor of guidance scales")
        half = embedding_dim // 2
        freq = torch.exp(torch.arange(half, dtype=torch.float32, device=w.device) * -(np.log(10000.0) / (half - 1)))
        emb = (w.to(torch.float32) * 1000.0)[:, None] * freq[None, :]
        emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)
        if embedding_dim % 2 == 1:
            emb = torch.nn.functional.pad(emb, (0, 1))
        return emb

    def _encode_image(self, image: torch.Tensor, generator) -> torch.Tensor:
        """``_encode_vae_image`` (pipeline_stable_diffusion_inpaint.py:746-758): posterior sample * scaling_factor"""
        if self.vae is None:
            raise ValueError("an image input needs a `vae` with encoder parameters")
        return self.vae.encode(image.to(torch.float32)).latent_dist.sample(generator, out_scale=self.vae.config.scaling_factor)

    def prepare_inpaint(self, image, mask_image, masked_image_latents, timesteps, batch_size: int, generator,
                        is_strength_max: bool, do_cfg: bool):
        """``prepare_latents`` + ``prepare_mask_latents`` of the inpaint pipeline (pipeline_stable_diffusion_inpaint.py:689-803),
        in the reference's order of random draws: image posterior, initial noise, masked-image posterior."""
        lc = self.vae.config.latent_channels if self.vae is not None else 4
        four = self.unet.config.in_channels == lc
        if not torch.is_tensor(image) or image.dim() != 4 or not torch.is_tensor(mask_image) or mask_image.dim() != 4:
            raise ValueError("`image` and `mask_image` have to be [B, C, H, W] tensors")
        image = image.to(torch.float32)
        f = 2 ** (len(self.vae.config.block_out_channels) - 1) if self.vae is not None else 8     # vae_scale_factor
        hl, wl = (image.shape[-2], image.shape[-1]) if image.shape[1] == lc else (image.shape[-2] // f, image.shape[-1] // f)
        image_latents = None
        if four or not is_strength_max:
            image_latents = image if image.shape[1] == lc else self._encode_image(image, generator)
            image_latents = image_latents.repeat(batch_size // image_latents.shape[0], 1, 1, 1)
        noise = torch.randn((batch_size, lc, hl, wl), generator=generator, dtype=torch.float32, device=image.device)
        if is_strength_max:
            latents = noise * self.scheduler.init_noise_sigma
        else:
            ts = torch.as_tensor(np.asarray(timesteps[:1]).reshape(-1)).repeat(batch_size)
            latents = self.scheduler.add_noise(image_latents, noise, ts)
        mask_px = (mask_image.to(torch.float32) >= 0.5).to(torch.float32)          # mask_processor: do_binarize
        mask = torch.nn.functional.interpolate(mask_px, size=(hl, wl))              # nearest, like the reference (:766-768)
        if masked_image_latents is None:
            if image.shape[1] == lc:
                raise ValueError("latents passed as `image` need `masked_image_latents`")
            masked_image_latents = self._encode_image(image * (mask_px < 0.5).to(image.dtype), generator)   # (:1134)
        for nm, t in (("masks", mask), ("images", masked_image_latents)):
            if batch_size % t.shape[0]:
                raise ValueError(f"The passed {nm} and the required batch size don't match: {t.shape[0]} vs {batch_size}.")
        mask = mask.repeat(batch_size // mask.shape[0], 1, 1, 1)
        mil = masked_image_latents.to(torch.float32).repeat(batch_size // masked_image_latents.shape[0], 1, 1, 1)
        if not four and lc + mask.shape[1] + mil.shape[1] != self.unet.config.in_channels:
            raise ValueError(f"Incorrect configuration settings! The config of `pipeline.unet` expects "
                             f"{self.unet.config.in_channels} but received `num_channels_latents`: {lc} + `num_channels_mask`: "
                             f"{mask.shape[1]} + `num_channels_masked_image`: {mil.shape[1]}")
        rep2 = (lambda t: torch.cat([t] * 2)) if do_cfg else (lambda t: t)
        return latents, dict(mask=mask, noise=noise, image_latents=image_latents, mask_in=rep2(mask), masked_in=rep2(mil))

    def get_timesteps(self, num_inference_steps: int, strength: float):
        """img2img: keep the last ``int(steps * strength)`` steps (pipeline_stable_diffusion_img2img.py:616-623)."""
        init_timestep = min(int(num_inference_steps * strength), num_inference_steps)
        t_start = max(num_inference_steps - init_timestep, 0)
        order = getattr(self.scheduler, "order", 1)
        return self.scheduler.timesteps[t_start * order:], num_inference_steps - t_start, t_start * order

    def prepare_image_latents(self, image: torch.Tensor, timestep, batch_size: int, generator=None) -> torch.Tensor:
        """img2img ``prepare_latents`` (pipeline_stable_diffusion_img2img.py:625-681): encode (unless `image` already has the
        latent channel count), ``* scaling_factor``, duplicate to the prompt batch, ``scheduler.add_noise`` at the first kept
        timestep. The posterior sample and its scaling are one launch (``latent_dist.sample(out_scale=...)``)."""
        if not torch.is_tensor(image) or image.dim() != 4:
            raise ValueError(f"`image` has to be a [B, C, H, W] tensor but is {type(image)}")
        if image.shape[1] == self.unet.config.in_channels:
            init = image.to(torch.float32)
        else:
            if self.vae is None:
                raise ValueError("an image input needs a `vae` with encoder parameters")
            dist = self.vae.encode(image.to(torch.float32)).latent_dist
            init = dist.sample(generator, out_scale=self.vae.config.scaling_factor)
        if batch_size > init.shape[0]:
            if batch_size % init.shape[0]:
                raise ValueError(f"Cannot duplicate `image` of batch size {init.shape[0]} to {batch_size} text prompts.")
            init = torch.cat([init] * (batch_size // init.shape[0]))
        noise = torch.randn(init.shape, generator=generator, dtype=torch.float32, device=init.device)
        ts = torch.as_tensor(np.asarray(timestep).reshape(-1)[:1]).repeat(init.shape[0])
        return self.scheduler.add_noise(init, noise, ts)

    @torch.no_grad()
    def __call__(self, prompt_embeds: Optional[torch.Tensor] = None, negative_prompt_embeds: Optional[torch.Tensor] = None,
                 height: Optional[int] = None, width: Optional[int] = None, num_inference_steps: int = 50,
                 guidance_scale: float = 7.5, guidance_rescale: float = 0.0, latents: Optional[torch.Tensor] = None,
                 generator=None, added_cond_kwargs: Optional[Dict[str, torch.Tensor]] = None,
                 negative_added_cond_kwargs: Optional[Dict[str, torch.Tensor]] = None,
                 callback_on_step_end: Optional[Callable] = None, vae_scale_factor: int = 8,
                 output_type: str = "latent", prompt_ids: Optional[torch.Tensor] = None,
                 negative_prompt_ids: Optional[torch.Tensor] = None, prompt_ids_2: Optional[torch.Tensor] = None,
                 negative_prompt_ids_2: Optional[torch.Tensor] = None, original_size=None,
                 crops_coords_top_left=(0, 0), target_size=None, fused_update: bool = True,
                 image: Optional[torch.Tensor] = None, strength: Optional[float] = None, eta: float = 0.0,
                 mask_image: Optional[torch.Tensor] = None, masked_image_latents: Optional[torch.Tensor] = None,
                 control_image: Optional[torch.Tensor] = None, controlnet_conditioning_scale: float = 1.0,
                 guess_mode: bool = False):
        """``image`` (extension of the text2img call = StableDiffusionImg2ImgPipeline.__call__,
        pipeline_stable_diffusion_img2img.py:735-1010): start from the encoded, re-noised image and run the last
        ``int(num_inference_steps * strength)`` steps (strength defaults to 0.8).

        ``image`` + ``mask_image`` [B|1, 1, H, W] in [0, 1], 1 = repaint (= StableDiffusionInpaintPipeline.__call__,
        pipeline_stable_diffusion_inpaint.py:1094-1236; strength defaults to 1.0): a 9-channel UNet gets
        ``[latents | mask | masked-image latents]`` every step, a 4-channel UNet has the kept region re-imposed after every
        step from the re-noised image latents."""
        inpaint = mask_image is not None
        if inpaint and image is None:
            raise ValueError("`mask_image` needs `image`")
        if strength is None:
            strength = 1.0 if inpaint else 0.8
        # guidance-distilled UNets (LCM) take the scale as an embedding instead of a doubled batch (:634-635, :846-852)
        tc_dim = getattr(self.unet.config, "time_cond_proj_dim", None)
        do_cfg = guidance_scale > 1.0 and tc_dim is None
        if image is not None and (strength < 0 or strength > 1):
            raise ValueError(f"The value of strength should in [0.0, 1.0] but is {strength}")
        if prompt_embeds is None:
            if prompt_ids is None:
                raise ValueError("Provide either `prompt_embeds` or `prompt_ids`")
            prompt_embeds, pooled = self.encode_prompt(prompt_ids, prompt_ids_2)
            neg_pooled = None
            if do_cfg and negative_prompt_embeds is None:
                if negative_prompt_ids is None:
                    # SDXL only: force_zeros_for_empty_prompt (pipeline_stable_diffusion_xl.py:378-381). The single-encoder
                    # pipelines encode the empty prompt "" through the text encoder instead (pipeline_stable_diffusion.py:409-441),
                    # and the tokenizer is not part of this path -> the caller must pass its token ids.
                    if self.text_encoder_2 is None:
                        raise ValueError("classifier-free guidance on a single-encoder pipeline needs `negative_prompt_ids` "
                                         "(the tokenised \"\" of the reference) or `negative_prompt_embeds`")
                    negative_prompt_embeds = torch.zeros_like(prompt_embeds)
                    neg_pooled = None if pooled is None else torch.zeros_like(pooled)
                else:
                    negative_prompt_embeds, neg_pooled = self.encode_prompt(negative_prompt_ids, negative_prompt_ids_2)
            if pooled is not None and added_cond_kwargs is None:   # SDXL micro-conditioning (:1007-1036)
                hw = (height or self.unet.config.sample_size * vae_scale_factor,
                      width or self.unet.config.sample_size * vae_scale_factor)
                tids = self.get_add_time_ids(tuple(original_size or hw), tuple(crops_coords_top_left),
                                             tuple(target_size or hw), pooled.shape[-1], pooled.device)
                tids = tids.repeat(pooled.shape[0], 1)
                added_cond_kwargs = {"text_embeds": pooled, "time_ids": tids}
                if do_cfg:
                    negative_added_cond_kwargs = {"text_embeds": neg_pooled, "time_ids": tids}
        if do_cfg and negative_prompt_embeds is None:
            raise ValueError("classifier-free guidance needs `negative_prompt_embeds`")
        B = prompt_embeds.shape[0]
        cfg = self.unet.config
        h = (height // vae_scale_factor) if height else cfg.sample_size
        w = (width // vae_scale_factor) if width else cfg.sample_size
        if latents is not None and height is None and width is None:
            h, w = latents.shape[-2:]
        if do_cfg:  # :813-814: [negative, positive]
            prompt_embeds = torch.cat([negative_prompt_embeds, prompt_embeds])
            if added_cond_kwargs is not None:
                neg = negative_added_cond_kwargs or added_cond_kwargs
                added_cond_kwargs = {k: torch.cat([neg[k], v]) for k, v in added_cond_kwargs.items()}
        import inspect
        # LatentConsistencyModelImg2ImgPipeline (pipeline_latent_consistency_img2img.py:760-764): a scheduler whose
        # set_timesteps takes `strength` (LCMScheduler) shortens its own distillation schedule and ALL num_inference_steps of it
        # run; every other scheduler keeps the SD img2img rule (the last int(steps * strength) entries, get_timesteps)
        lcm_strength = image is not None and "strength" in inspect.signature(self.scheduler.set_timesteps).parameters
        if lcm_strength:
            self.scheduler.set_timesteps(num_inference_steps, strength=strength)
        else:
            self.scheduler.set_timesteps(num_inference_steps)
        timesteps, first, inp = self.scheduler.timesteps, 0, None
        if image is not None:
            if not lcm_strength:
                timesteps, _, first = self.get_timesteps(num_inference_steps, strength)
            if len(timesteps) < 1:
                raise ValueError(f"After adjusting the num_inference_steps by strength parameter: {strength}, the number of "
                                 "pipeline steps is 0 which is < 1 and not appropriate for this pipeline.")
            if inpaint:
                latents, inp = self.prepare_inpaint(image, mask_image, masked_image_latents, timesteps, B, generator,
                                                    strength == 1.0, do_cfg)
            else:
                latents = self.prepare_image_latents(image, timesteps[:1], B, generator)
        else:
            latents = self.prepare_latents(B, cfg.in_channels, h, w, torch.float32, generator, latents, prompt_embeds.device)
        step_params = inspect.signature(self.scheduler.step).parameters   # prepare_extra_step_kwargs (:520-535)
        extra = {}
        if "eta" in step_params:
            extra["eta"] = eta
        if "generator" in step_params:
            extra["generator"] = generator
        unet_kw = {}
        if tc_dim is not None:
            w = torch.full((B,), float(guidance_scale) - 1.0, device=latents.device)
            unet_kw["timestep_cond"] = self.get_guidance_scale_embedding(w, embedding_dim=tc_dim)
        fused = self._fused_plan(guidance_rescale, latents.device) if fused_update and not eta else None
        nine = inp is not None and cfg.in_channels != latents.shape[1]

        def control(x_in, t):
            """ControlNet residuals of this step (pipeline_controlnet.py:1148-1192): the ControlNet sees the UNet's input batch
            -- or, in guess mode under CFG, only the conditional half, the unconditional half getting zero residuals"""
            if control_image is None:
                return {}
            if self.controlnet is None:
                raise ValueError("`control_image` needs a `controlnet`")
            half = guess_mode and do_cfg
            pick = (lambda v: v.chunk(2)[1]) if half else (lambda v: v)
            img = control_image if (half or not do_cfg) else torch.cat([control_image] * 2)
            added = None if added_cond_kwargs is None else {k: pick(v) for k, v in added_cond_kwargs.items()}
            d, m = self.controlnet(pick(x_in)[:, :latents.shape[1]], t, encoder_hidden_states=pick(prompt_embeds),
                                   controlnet_cond=img, conditioning_scale=controlnet_conditioning_scale,
                                   guess_mode=guess_mode, added_cond_kwargs=added, return_dict=False)
            if half:
                d, m = tuple(torch.cat([torch.zeros_like(x), x]) for x in d), torch.cat([torch.zeros_like(m), m])
            return dict(down_block_additional_residuals=d, mid_block_additional_residual=m)

        def extend(x):   # 9-channel inpainting UNet: [scaled latents | mask | masked-image latents] (:1194-1198)
            return torch.cat([x, inp["mask_in"], inp["masked_in"]], dim=1) if nine else x

        def reimpose(lat, step_no):   # 4-channel UNet: keep the unmasked region on the re-noised image latents (:1218-1231)
            if inp is None or nine:
                return lat
            proper = inp["image_latents"]
            if step_no < len(timesteps) - 1:
                nt = torch.as_tensor(np.asarray(timesteps[step_no + 1]).reshape(1)).repeat(proper.shape[0])
                proper = self.scheduler.add_noise(proper, inp["noise"], nt)
            return (1 - inp["mask"]) * proper + inp["mask"] * lat

        for i, t in enumerate(timesteps, start=first):
            latent_model_input = torch.cat([latents] * 2) if do_cfg else latents
            if fused is not None:
                # guidance combine + scheduler update as ONE device pass over the latents (mi355x_sd_cfg_axpby): the
                # epsilon-prediction step of Euler / DDIM(eta=0) is prev = a*x + b*eps with per-step (a, b) kept in HBM
                scales, coef, lib, stream = fused
                scaled = latent_model_input * scales[i]
                noise_pred = self.unet(extend(scaled), t, encoder_hidden_states=prompt_embeds,
                                       added_cond_kwargs=added_cond_kwargs, return_dict=False, **unet_kw, **control(scaled, t))[0]
                lat = latents.contiguous()
                out = torch.empty_like(lat)
                n, cp = lat.numel(), coef.data_ptr() + 8 * i
                if do_cfg:
                    rc = lib.mi355x_sd_cfg_axpby(lat.data_ptr(), noise_pred.data_ptr(), noise_pred.data_ptr() + 4 * n,
                                                 out.data_ptr(), cp, float(guidance_scale), n, stream())
                else:
                    rc = lib.mi355x_sd_axpby(lat.data_ptr(), noise_pred.data_ptr(), out.data_ptr(), cp, n, stream())
                if rc:
                    from . import _lib
                    _lib.check(rc)
                latents = reimpose(out, i - first)
                if callback_on_step_end is not None:
                    cb = callback_on_step_end(self, i, t, {"latents": latents})
                    latents = cb.pop("latents", latents)
                continue
            latent_model_input = self.scheduler.scale_model_input(latent_model_input, t)
            noise_pred = self.unet(extend(latent_model_input), t, encoder_hidden_states=prompt_embeds,
                                   added_cond_kwargs=added_cond_kwargs, return_dict=False, **unet_kw,
                                   **control(latent_model_input, t))[0]
            if do_cfg:
                noise_uncond, noise_text = noise_pred.chunk(2)
                noise_pred = noise_uncond + guidance_scale * (noise_text - noise_uncond)
                if guidance_rescale > 0.0:
                    noise_pred = rescale_noise_cfg(noise_pred, noise_text, guidance_rescale)
            latents = reimpose(self.scheduler.step(noise_pred, t, latents, return_dict=False, **extra)[0], i - first)
            if callback_on_step_end is not None:
                out = callback_on_step_end(self, i, t, {"latents": latents})
                latents = out.pop("latents", latents)
        if output_type == "latent":
            return latents
        return self.decode_latents(latents, output_type)


class StableDiffusion3Denoiser:
    """The denoising loop of ``StableDiffusion3Pipeline.__call__`` (pipelines/stable_diffusion_3/
    pipeline_stable_diffusion_3.py:772-870) with the MI355X MMDiT in the ``transformer`` slot: CFG batch doubling
    ([negative, positive]), the transformer call, the guidance combine, ``FlowMatchEulerDiscreteScheduler.step`` and
    ``callback_on_step_end``; then (:880-886) ``vae.decode(latents / scaling_factor + shift_factor)``.
    Prompt encoding (2 x CLIP + T5) stays outside: the loop takes ``prompt_embeds`` [B, L, joint_attention_dim] and
    ``pooled_prompt_embeds`` [B, pooled_projection_dim]."""

    def __init__(self, transformer, scheduler, vae=None, text_encoder=None, text_encoder_2=None, text_encoder_3=None):
        self.transformer, self.scheduler, self.vae = transformer, scheduler, vae
        self.text_encoder, self.text_encoder_2, self.text_encoder_3 = text_encoder, text_encoder_2, text_encoder_3

    def encode_prompt(self, input_ids: torch.Tensor, input_ids_2: torch.Tensor, input_ids_3: torch.Tensor,
                      clip_skip: Optional[int] = None):
        """Token ids of the three tokenizers -> (prompt_embeds [B, S_clip + S_t5, joint_dim], pooled [B, P1 + P2]);
        pipeline_stable_diffusion_3.py:375-398: both CLIP encoders (with projection) contribute hidden_states[-2],
        concatenated on the channel axis and zero-padded to the T5 width, then the T5 sequence is appended on the token
        axis; pooled = the two projected EOS rows side by side."""
        if self.text_encoder is None or self.text_encoder_2 is None or self.text_encoder_3 is None:
            raise ValueError("encode_prompt needs `text_encoder`, `text_encoder_2` (CLIP with projection) and "
                             "`text_encoder_3` (T5)")
        embeds, pooled = [], []
        for enc, ids in ((self.text_encoder, input_ids), (self.text_encoder_2, input_ids_2)):
            out = enc(ids, output_hidden_states=True)
            pooled.append(out[0])
            embeds.append(out.hidden_states[-2 if clip_skip is None else -(clip_skip + 2)])
        clip = torch.cat(embeds, dim=-1)
        t5 = self.text_encoder_3(input_ids_3)[0]
        clip = torch.nn.functional.pad(clip, (0, t5.shape[-1] - clip.shape[-1]))
        return torch.cat([clip, t5], dim=-2), torch.cat(pooled, dim=-1)

    @torch.no_grad()
    def __call__(self, prompt_embeds: torch.Tensor, pooled_prompt_embeds: torch.Tensor,
                 negative_prompt_embeds: Optional[torch.Tensor] = None,
                 negative_pooled_prompt_embeds: Optional[torch.Tensor] = None, height: Optional[int] = None,
                 width: Optional[int] = None, num_inference_steps: int = 28, guidance_scale: float = 7.0,
                 latents: Optional[torch.Tensor] = None, generator=None, callback_on_step_end: Optional[Callable] = None,
                 vae_scale_factor: int = 8, output_type: str = "latent"):
        do_cfg = guidance_scale > 1.0
        if do_cfg and (negative_prompt_embeds is None or negative_pooled_prompt_embeds is None):
            raise ValueError("classifier-free guidance needs `negative_prompt_embeds` and `negative_pooled_prompt_embeds`")
        cfg = self.transformer.config
        B = prompt_embeds.shape[0]
        h = (height // vae_scale_factor) if height else cfg.sample_size
        w = (width // vae_scale_factor) if width else cfg.sample_size
        shape = (B, cfg.in_channels, h, w)
        if latents is None:
            latents = torch.randn(shape, generator=generator, dtype=torch.float32, device=prompt_embeds.device)
        elif height is None and width is None:
            shape = tuple(latents.shape)
        if tuple(latents.shape) != shape:
            raise ValueError(f"Unexpected latents shape, got {tuple(latents.shape)}, expected {shape}")
        if do_cfg:
            prompt_embeds = torch.cat([negative_prompt_embeds, prompt_embeds], dim=0)
            pooled_prompt_embeds = torch.cat([negative_pooled_prompt_embeds, pooled_prompt_embeds], dim=0)
        self.scheduler.set_timesteps(num_inference_steps)
        for i, t in enumerate(self.scheduler.timesteps):
            x = torch.cat([latents] * 2) if do_cfg else latents
            v = self.transformer(hidden_states=x, timestep=t, encoder_hidden_states=prompt_embeds,
                                 pooled_projections=pooled_prompt_embeds, return_dict=False)[0]
            if do_cfg:
                v_uncond, v_text = v.chunk(2)
                v = v_uncond + guidance_scale * (v_text - v_uncond)
            latents = self.scheduler.step(v, t, latents, return_dict=False)[0]
            if callback_on_step_end is not None:
                out = callback_on_step_end(self, i, t, {"latents": latents})
                latents = out.pop("latents", latents)
        if output_type == "latent":
            return latents
        if self.vae is None:
            raise ValueError("output_type != 'latent' needs a `vae`")
        if output_type not in ("pt", "np"):
            raise ValueError(f"output_type must be 'latent', 'pt' or 'np', got {output_type!r}")
        vc = self.vae.config
        z = latents / vc.scaling_factor + (getattr(vc, "shift_factor", None) or 0.0)
        image = (self.vae.decode(z, return_dict=False)[0] / 2 + 0.5).clamp(0, 1)
        return image if output_type == "pt" else image.cpu().permute(0, 2, 3, 1).float().numpy()


class DiTDenoiser:
    """The loop of ``DiTPipeline.__call__`` (pipelines/dit/pipeline_dit.py:158-246) with the MI355X DiT in the ``transformer``
    slot: class-conditional classifier-free guidance with the null class (index ``num_embeds_ada_norm``) on the DUPLICATED
    latent half (:182-184), guidance applied to the epsilon channels only (:209-216), the learned-sigma channels dropped
    before ``scheduler.step`` (:219-225), then ``vae.decode(latents / scaling_factor)`` and the denormalise (:235-246)."""

    def __init__(self, transformer, scheduler, vae=None):
        self.transformer, self.scheduler, self.vae = transformer, scheduler, vae

    @torch.no_grad()
    def __call__(self, class_labels, guidance_scale: float = 4.0, num_inference_steps: int = 50, generator=None,
                 latents: Optional[torch.Tensor] = None, output_type: str = "latent",
                 callback_on_step_end: Optional[Callable] = None, device=None):
        cfg = self.transformer.config
        labels = torch.as_tensor(class_labels).reshape(-1)
        device = device or (latents.device if latents is not None else labels.device)
        labels = labels.to(device)
        B, C, side = labels.numel(), cfg.in_channels, cfg.sample_size
        if latents is None:
            latents = torch.randn((B, C, side, side), generator=generator, dtype=torch.float32, device=device)
        elif tuple(latents.shape[:2]) != (B, C):
            raise ValueError(f"Unexpected latents shape, got {tuple(latents.shape)}, expected ({B}, {C}, h, w)")
        do_cfg = guidance_scale > 1
        x = torch.cat([latents] * 2) if do_cfg else latents
        null = torch.full((B,), cfg.num_embeds_ada_norm, dtype=labels.dtype, device=device)
        labels_in = torch.cat([labels, null]) if do_cfg else labels
        self.scheduler.set_timesteps(num_inference_steps)
        for i, t in enumerate(self.scheduler.timesteps):
            if do_cfg:
                half = x[: len(x) // 2]
                x = torch.cat([half, half], dim=0)
            x = self.scheduler.scale_model_input(x, t)
            ts = torch.as_tensor(t, device=device).reshape(-1).expand(x.shape[0])
            noise = self.transformer(x, timestep=ts, class_labels=labels_in, return_dict=False)[0]
            if do_cfg:
                eps, rest = noise[:, :C], noise[:, C:]
                cond, uncond = eps.chunk(2, dim=0)
                half_eps = uncond + guidance_scale * (cond - uncond)
                noise = torch.cat([torch.cat([half_eps, half_eps], dim=0), rest], dim=1)
            model_output = noise[:, :C] if cfg.out_channels // 2 == C else noise   # learned sigma is not used by the step
            x = self.scheduler.step(model_output, t, x, return_dict=False)[0]
            if callback_on_step_end is not None:
                x = callback_on_step_end(self, i, t, {"latents": x}).pop("latents", x)
        latents = x.chunk(2, dim=0)[0] if do_cfg else x
        if output_type == "latent":
            return latents
        if self.vae is None:
            raise ValueError("output_type != 'latent' needs a `vae`")
        if output_type not in ("pt", "np"):
            raise ValueError(f"output_type must be 'latent', 'pt' or 'np', got {output_type!r}")
        image = self.vae.decode(latents, return_dict=False, in_scale=1.0 / self.vae.config.scaling_factor)[0]
        image = (image / 2 + 0.5).clamp(0, 1)
        return image if output_type == "pt" else image.cpu().permute(0, 2, 3, 1).float().numpy()
