"""MI355X-native CLIP encoders. Text encoder: ``CLIPTextModel`` / ``CLIPTextModelWithProjection`` as the
``text_encoder`` / ``text_encoder_2`` of the Stable-Diffusion pipelines (``encode_prompt``,
pipeline_stable_diffusion.py:287-463; SDXL takes ``hidden_states[-2]`` and the projected pooled output,
pipeline_stable_diffusion_xl.py:373-395).

Mirrors PPD/transformers/clip/modeling.py (CLIPTextTransformer.forward :745-833): token + position embedding,
pre-LayerNorm encoder layers with causal self-attention, final LayerNorm, EOS pooling, optional text_projection. The
program reuses the UNet's kernels -- LayerNorm, fused-QKV GEMM (bias epilogue), the flash attention kernel with the
causal mask as its additive bias, output / MLP GEMMs with residual epilogues -- plus an embedding gather and an
elementwise activation kernel (quick_gelu / gelu). Runs once per prompt; no CPU fallback.

``CLIPVisionModelWithProjection`` (modeling.py:162-196, 896-953, 1300-1373) is the image encoder of the IP-Adapter pipelines
(``encode_image``: ``image_encoder(image).image_embeds`` -> the UNet's ``added_cond_kwargs["image_embeds"]``): the bias-free patch
convolution is the patchify kernel + one GEMM that writes the patch rows of the [B, 1 + N, D] token buffer in place with the
position embedding as its residual input; the class-token row is a constant written at plan time; the encoder layers are the
text tower's without a mask; post_layernorm reads only the class-token rows.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, List, Mapping, Optional

import torch

from . import _lib
from .checkpoint import PretrainedMixin, Table
from .program import DeviceProgram, _Plan, _Ref, _V

Tensor = torch.Tensor

CLIP_DEFAULTS = dict(vocab_size=49408, hidden_size=512, intermediate_size=2048, projection_dim=512,
                     num_hidden_layers=12, num_attention_heads=8, max_position_embeddings=77, hidden_act="quick_gelu",
                     layer_norm_eps=1e-5, eos_token_id=2, with_projection=False)
_ACT_KIND = {"quick_gelu": 0, "gelu": 1}


def normalize_config(config: Mapping) -> dict:
    cfg = dict(CLIP_DEFAULTS)
    cfg.update({k: v for k, v in config.items() if not k.startswith("_")})
    if cfg["hidden_act"] not in _ACT_KIND:
        raise NotImplementedError(f"hidden_act={cfg['hidden_act']!r} (quick_gelu and gelu are implemented)")
    D, H = cfg["hidden_size"], cfg["num_attention_heads"]
    if D % H or (D // H) % 8 or (D // H) > 160 or cfg["intermediate_size"] % 8:
        raise ValueError("unsupported geometry: head_dim must be a multiple of 8 and <= 160")
    return cfg


def clip_param_shapes(config: Mapping) -> Dict[str, tuple]:
    """name -> shape in Paddle layouts (Linear [in, out]); names as in the reference checkpoints."""
    cfg = normalize_config(config)
    D, I = cfg["hidden_size"], cfg["intermediate_size"]
    S: Dict[str, tuple] = {"text_model.embeddings.token_embedding.weight": Table((cfg["vocab_size"], D)),
                           "text_model.embeddings.position_embedding.weight": Table((cfg["max_position_embeddings"], D))}
    _encoder_layer_shapes(S, "text_model", cfg["num_hidden_layers"], D, I)
    S["text_model.final_layer_norm.weight"], S["text_model.final_layer_norm.bias"] = (D,), (D,)
    if cfg["with_projection"]:
        S["text_projection.weight"] = (D, cfg["projection_dim"])
    return S


def synth_clip_params(config: Mapping, seed: int = 1234, device="cpu") -> Dict[str, Tensor]:
    g = torch.Generator(device=device).manual_seed(seed)
    P: Dict[str, Tensor] = {}
    for name, shape in clip_param_shapes(config).items():
        r = torch.randn(shape, generator=g, device=device)
        if name.endswith(".bias"):
            t = r * 0.02
        elif "embedding" in name:
            t = r * 0.5
        elif len(shape) == 1:
            t = 1.0 + r * 0.02
        else:
            t = r / shape[0] ** 0.5
        P[name] = t
    return P


def _load_encoder_layers(W, get, bf, tower: str, n_layers: int) -> None:
    """CLIPEncoderLayer weights (modeling.py CLIPAttention / CLIPMLP) of `tower` ("text_model" | "vision_model"): fused
    [q; k; v] projection, [out, in] bf16 matrices, fp32 biases and LayerNorm affines"""
    for i in range(n_layers):
        b = f"{tower}.encoder.layers.{i}"
        a = b + ".self_attn."
        W[f"l{i}.qkv.w"] = bf(torch.cat([get(a + n + ".weight").t() for n in ("q_proj", "k_proj", "v_proj")], 0))
        W[f"l{i}.qkv.b"] = torch.cat([get(a + n + ".bias") for n in ("q_proj", "k_proj", "v_proj")], 0).contiguous()
        for key, name in ((f"l{i}.out", a + "out_proj"), (f"l{i}.fc1", b + ".mlp.fc1"), (f"l{i}.fc2", b + ".mlp.fc2")):
            W[key + ".w"] = bf(get(name + ".weight").t())
            W[key + ".b"] = get(name + ".bias").contiguous()
        for key, name in ((f"l{i}.ln1", b + ".layer_norm1"), (f"l{i}.ln2", b + ".layer_norm2")):
            W[key + ".g"], W[key + ".b"] = get(name + ".weight").contiguous(), get(name + ".bias").contiguous()


def _encoder_layer_shapes(S: Dict[str, tuple], tower: str, n_layers: int, D: int, I: int) -> None:
    for i in range(n_layers):
        b = f"{tower}.encoder.layers.{i}"
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            S[f"{b}.self_attn.{nm}.weight"], S[f"{b}.self_attn.{nm}.bias"] = (D, D), (D,)
        S[b + ".layer_norm1.weight"], S[b + ".layer_norm1.bias"] = (D,), (D,)
        S[b + ".mlp.fc1.weight"], S[b + ".mlp.fc1.bias"] = (D, I), (I,)
        S[b + ".mlp.fc2.weight"], S[b + ".mlp.fc2.bias"] = (I, D), (D,)
        S[b + ".layer_norm2.weight"], S[b + ".layer_norm2.bias"] = (D,), (D,)


def _emit_encoder_layers(model, emit, persist, hidden: List[Tensor], B: int, S: int, mask_ptr: Optional[int]) -> None:
    """CLIPEncoder.forward: pre-LayerNorm attention + MLP blocks, hidden[i] -> hidden[i + 1] (modeling.py CLIPEncoderLayer).
    mask_ptr: fp32 [S, S] additive mask shared by batch items and heads (the text tower's causal mask) or None."""
    cfg, lib, W, stream, gemm_ws = model.cfg, model._lib, model.w, model._stream_ptr, model._gemm_ws
    D, H, I, n = cfg["hidden_size"], cfg["num_attention_heads"], cfg["intermediate_size"], cfg["num_hidden_layers"]
    d, eps, rows = D // H, float(cfg["layer_norm_eps"]), B * S

    def linear(a: Tensor, lda, wkey, out: Tensor, ldc, R: Optional[Tensor] = None):
        w = W[wkey + ".w"]
        N, K = w.shape
        emit(lib.mi355x_sd_linear, (a.data_ptr(), lda, w.data_ptr(), out.data_ptr(), ldc, rows, N, K, W[wkey + ".b"].data_ptr(),
                                    None, 0, 0, R.data_ptr() if R is not None else None, N if R is not None else 0, 1.0, 0,
                                    *gemm_ws, stream), "gemm", 2.0 * rows * N * K, f"{rows}x{N}x{K}")

    def lnorm(x: Tensor, key, out: Tensor):
        emit(lib.mi355x_sd_layernorm, (x.data_ptr(), rows, D, D, W[key + ".g"].data_ptr(), W[key + ".b"].data_ptr(), eps,
                                       out.data_ptr(), D, stream), "ln")

    ln, ao = persist((rows, D), _lib.elem_dtype()), persist((rows, D), _lib.elem_dtype())
    qkv = persist((rows, 3 * D), _lib.elem_dtype())
    f1, f2 = persist((rows, I), _lib.elem_dtype()), persist((rows, I), _lib.elem_dtype())
    mid = persist((rows, D), _lib.elem_dtype())
    for i in range(n):
        x, y = hidden[i], hidden[i + 1]
        lnorm(x, f"l{i}.ln1", ln)
        linear(ln, D, f"l{i}.qkv", qkv, 3 * D)
        qp = qkv.data_ptr()
        emit(lib.mi355x_sd_sdpa, (qp, qp + 2 * D, qp + 4 * D, mask_ptr, ao.data_ptr(), B, H, S, S, d,
                                  S * 3 * D, 3 * D, S * 3 * D, 3 * D, S * 3 * D, 3 * D, S * D, D, 0, 0, S if mask_ptr else 0,
                                  d ** -0.5, stream), "attn", 4.0 * B * H * S * S * d, f"{B}x{H}x{S}x{S}x{d}")
        linear(ao, D, f"l{i}.out", mid, D, R=x)
        lnorm(mid, f"l{i}.ln2", ln)
        linear(ln, D, f"l{i}.fc1", f1, I)
        emit(lib.mi355x_sd_activation, (f1.data_ptr(), f2.data_ptr(), rows * I, _ACT_KIND[cfg["hidden_act"]], stream), "misc")
        linear(f2, I, f"l{i}.fc2", y, D, R=mid)


class CLIPTextModelOutput(SimpleNamespace):
    """last_hidden_state, pooler_output, hidden_states (tuple or None), text_embeds (with projection)."""

    def __getitem__(self, i):   # tuple-style access used by the pipelines: out[0] (SDXL: the pooled text_embeds), out[-1]
        first = (self.text_embeds, self.last_hidden_state) if self.text_embeds is not None else \
            (self.last_hidden_state, self.pooler_output)
        return tuple(v for v in first + (self.hidden_states,) if v is not None)[i]


class CLIPTextModel(DeviceProgram, PretrainedMixin):
    _param_shapes = staticmethod(clip_param_shapes)

    def __init__(self, config: Mapping, params: Mapping[str, Tensor], device="cuda", use_graph: bool = True,
                 profile: bool = False):
        self._init_backend(device, use_graph, profile)
        self.cfg = normalize_config(dict(config, with_projection=self._WITH_PROJECTION or config.get("with_projection", False)))
        self.config = SimpleNamespace(**self.cfg)
        self._load_weights(params)

    _WITH_PROJECTION = False

    # ------------------------------------------------------------------ weights
    def _load_weights(self, params: Mapping[str, Tensor]) -> None:
        cfg, dev, W = self.cfg, self.device, self.w
        shapes = clip_param_shapes(cfg)
        missing = [k for k in shapes if k not in params]
        if missing:
            raise KeyError(f"missing parameters: {missing[:5]}{'...' if len(missing) > 5 else ''}")

        def get(name):
            t = params[name]
            if tuple(t.shape) != shapes[name]:
                raise ValueError(f"{name}: expected shape {shapes[name]} (Paddle layout), got {tuple(t.shape)}")
            return t.to(device=dev, dtype=torch.float32)

        bf = lambda t: t.to(_lib.elem_dtype()).contiguous()  # noqa: E731
        W["tok"] = bf(get("text_model.embeddings.token_embedding.weight"))
        W["pos"] = bf(get("text_model.embeddings.position_embedding.weight"))
        _load_encoder_layers(W, get, bf, "text_model", cfg["num_hidden_layers"])
        W["lnf.g"] = get("text_model.final_layer_norm.weight").contiguous()
        W["lnf.b"] = get("text_model.final_layer_norm.bias").contiguous()
        if cfg["with_projection"]:
            W["proj.w"] = bf(get("text_projection.weight").t())

    # ------------------------------------------------------------------ plan
    def _build_plan(self, B: int, S: int) -> _Plan:
        cfg, lib, dev, W = self.cfg, self._lib, self.device, self.w
        stream = self._stream_ptr
        D, H, I, n = cfg["hidden_size"], cfg["num_attention_heads"], cfg["intermediate_size"], cfg["num_hidden_layers"]
        d, eps = D // H, float(cfg["layer_norm_eps"])
        rows = B * S
        plan = _Plan()
        prog: List[tuple] = []
        keep: List[Tensor] = []

        def persist(shape, dtype) -> Tensor:
            t = torch.empty(shape, device=dev, dtype=dtype)
            keep.append(t)
            return t

        def emit(fn, args, kind, flops=0.0, desc=""):
            prog.append((fn, tuple(args), kind if not desc else f"{kind}:{desc}", flops))

        def lnorm(x: Tensor, key, out: Tensor):
            emit(lib.mi355x_sd_layernorm, (x.data_ptr(), rows, D, D, W[key + ".g"].data_ptr(), W[key + ".b"].data_ptr(), eps,
                                           out.data_ptr(), D, stream), "ln")

        plan.ids = persist((rows,), torch.int32)
        # causal mask as the attention kernel's additive bias: [S, S] shared by every batch item and head
        mask = persist((S, S), torch.float32)
        mask.copy_(torch.triu(torch.full((S, S), -1e30), diagonal=1))
        plan.consts = [mask]       # filled here, read by every run (paddlemix_amd/export.py ships its contents)
        plan.hidden = [persist((rows, D), _lib.elem_dtype()) for _ in range(n + 1)]   # encoder hidden_states tuple
        plan.last = persist((rows, D), _lib.elem_dtype())
        emit(lib.mi355x_sd_embed_tokens, (plan.ids.data_ptr(), rows, S, W["tok"].data_ptr(), W["pos"].data_ptr(), D,
                                          plan.hidden[0].data_ptr(), D, stream), "misc")
        _emit_encoder_layers(self, emit, persist, plan.hidden, B, S, mask.data_ptr())
        lnorm(plan.hidden[n], "lnf", plan.last)
        plan.prog, plan.keep, plan.graph = prog, keep, None
        plan.out = plan.last
        plan.B, plan.S = B, S
        return plan

    def forward(self, input_ids: Tensor, attention_mask=None, position_ids=None, output_attentions=None,
                output_hidden_states: Optional[bool] = None, return_dict: Optional[bool] = True):
        if input_ids is None:
            raise ValueError("You have to specify input_ids")
        if attention_mask is not None or position_ids is not None or output_attentions:
            raise NotImplementedError("attention_mask / position_ids / output_attentions are not implemented")
        cfg = self.cfg
        ids = input_ids.reshape(-1, input_ids.shape[-1])
        B, S = ids.shape
        if S > cfg["max_position_embeddings"]:
            raise ValueError(f"sequence length {S} exceeds max_position_embeddings {cfg['max_position_embeddings']}")
        if not self._emulated and not ids.is_cuda:
            raise _lib.MI355XError("inputs must be GPU tensors (no CPU fallback)")
        if int(ids.min()) < 0 or int(ids.max()) >= cfg["vocab_size"]:
            raise ValueError("input_ids out of range of the token embedding")
        key = (B, S)
        if key not in self._plans:
            self._plans[key] = self._build_plan(B, S)
        plan = self._plans[key]
        if self._emulated:
            plan.ids.copy_(ids.reshape(-1).to(torch.int32))
            self._run_eager(plan)
        else:
            cur = torch.cuda.current_stream(self.device)
            self._stream.wait_stream(cur)
            with torch.cuda.stream(self._stream):
                plan.ids.copy_(ids.reshape(-1).to(torch.int32), non_blocking=True)
                self.run(plan)
            cur.wait_stream(self._stream)
        D = cfg["hidden_size"]
        last = plan.last.reshape(B, S, D).float()
        pos = ids.argmax(-1) if cfg["eos_token_id"] == 2 else (ids == cfg["eos_token_id"]).int().argmax(-1)
        pooled = last[torch.arange(B, device=last.device), pos.to(last.device)]
        out = CLIPTextModelOutput(last_hidden_state=last, pooler_output=pooled, hidden_states=None, text_embeds=None)
        if output_hidden_states:
            out.hidden_states = tuple(h.reshape(B, S, D).float() for h in plan.hidden)
        if cfg["with_projection"]:
            out.text_embeds = self._project(pooled)
        if not return_dict:
            return tuple(v for v in (out.text_embeds if cfg["with_projection"] else out.last_hidden_state,
                                     out.last_hidden_state if cfg["with_projection"] else out.pooler_output,
                                     out.hidden_states) if v is not None)
        return out

    __call__ = forward

    def final_layer_norm(self, hidden: Tensor) -> Tensor:
        """``text_model.final_layer_norm`` on arbitrary hidden states [B, S, D] -- what the single-encoder pipelines apply to the
        clip_skip layer (pipeline_stable_diffusion.py:378-391); also reachable as ``model.text_model.final_layer_norm``."""
        D = self.cfg["hidden_size"]
        x = hidden.reshape(-1, D).to(_lib.elem_dtype()).contiguous()
        out = torch.empty_like(x)
        s = 0 if self._emulated else torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._lib.mi355x_sd_layernorm(x.data_ptr(), x.shape[0], D, D, self.w["lnf.g"].data_ptr(), self.w["lnf.b"].data_ptr(),
                                                 float(self.cfg["layer_norm_eps"]), out.data_ptr(), D, s))
        return out.reshape(hidden.shape).float()

    @property
    def text_model(self):
        return SimpleNamespace(final_layer_norm=self.final_layer_norm)

    def _project(self, pooled: Tensor) -> Tensor:
        """text_projection (bias-free Linear) of the pooled rows through the C ABI."""
        w = self.w["proj.w"]
        N, K = w.shape
        B = pooled.shape[0]
        a = pooled.to(_lib.elem_dtype()).contiguous()
        out = torch.empty((B, N), device=a.device, dtype=torch.float32)
        s = 0 if self._emulated else torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._lib.mi355x_sd_linear(a.data_ptr(), K, w.data_ptr(), out.data_ptr(), N, B, N, K, None, None, 0, 0,
                                              None, 0, 1.0, _lib.OUT_F32, *self._gemm_ws, s))
        return out


class CLIPTextModelWithProjection(CLIPTextModel):
    """CLIPTextModelWithProjection (modeling.py): forward returns text_embeds = text_projection(pooled)."""
    _WITH_PROJECTION = True


# ---------------------------------------------------------------------------------------------------------------- vision tower
CLIP_VISION_DEFAULTS = dict(hidden_size=768, intermediate_size=3072, projection_dim=512, num_hidden_layers=12,
                            num_attention_heads=12, num_channels=3, image_size=224, patch_size=32, hidden_act="quick_gelu",
                            layer_norm_eps=1e-5)


def normalize_vision_config(config: Mapping) -> dict:
    cfg = dict(CLIP_VISION_DEFAULTS)
    cfg.update({k: v for k, v in config.items() if not k.startswith("_")})
    if cfg["hidden_act"] not in _ACT_KIND:
        raise NotImplementedError(f"hidden_act={cfg['hidden_act']!r} (quick_gelu and gelu are implemented)")
    D, H = cfg["hidden_size"], cfg["num_attention_heads"]
    if D % H or (D // H) % 8 or (D // H) > 160 or cfg["intermediate_size"] % 8 or D % 8:
        raise ValueError("unsupported geometry: head_dim must be a multiple of 8 and <= 160")
    if cfg["image_size"] % cfg["patch_size"]:
        raise ValueError("image_size must be a multiple of patch_size")
    return cfg


def clip_vision_param_shapes(config: Mapping) -> Dict[str, tuple]:
    """name -> shape in Paddle layouts; names as in the reference checkpoints (incl. the upstream spelling `pre_layrnorm`)."""
    cfg = normalize_vision_config(config)
    D, I, p = cfg["hidden_size"], cfg["intermediate_size"], cfg["patch_size"]
    S: Dict[str, tuple] = {"vision_model.embeddings.class_embedding": (D,),
                           "vision_model.embeddings.patch_embedding.weight": (D, cfg["num_channels"], p, p),
                           "vision_model.embeddings.position_embedding.weight": Table(((cfg["image_size"] // p) ** 2 + 1, D)),
                           "vision_model.pre_layrnorm.weight": (D,), "vision_model.pre_layrnorm.bias": (D,)}
    _encoder_layer_shapes(S, "vision_model", cfg["num_hidden_layers"], D, I)
    S["vision_model.post_layernorm.weight"], S["vision_model.post_layernorm.bias"] = (D,), (D,)
    S["visual_projection.weight"] = (D, cfg["projection_dim"])
    return S


def synth_clip_vision_params(config: Mapping, seed: int = 1234, device="cpu") -> Dict[str, Tensor]:
    g = torch.Generator(device=device).manual_seed(seed)
    P: Dict[str, Tensor] = {}
    for name, shape in clip_vision_param_shapes(config).items():
        r = torch.randn(shape, generator=g, device=device)
        if name.endswith(".bias"):
            t = r * 0.02
        elif "embedding" in name and len(shape) <= 2:
            t = r * 0.5
        elif len(shape) == 1:
            t = 1.0 + r * 0.02
        elif len(shape) == 4:
            t = r / (shape[1] * shape[2] * shape[3]) ** 0.5
        else:
            t = r / shape[0] ** 0.5
        P[name] = t
    return P


class CLIPVisionModelOutput(SimpleNamespace):
    def __getitem__(self, i):
        return (self.image_embeds, self.last_hidden_state)[i]


class CLIPVisionModelWithProjection(DeviceProgram, PretrainedMixin):
    _param_shapes = staticmethod(clip_vision_param_shapes)

    def __init__(self, config: Mapping, params: Mapping[str, Tensor], device="cuda", use_graph: bool = True,
                 profile: bool = False):
        self._init_backend(device, use_graph, profile)
        self.cfg = normalize_vision_config(config)
        self.config = SimpleNamespace(**self.cfg)
        cfg, dev, W = self.cfg, self.device, self.w
        shapes = clip_vision_param_shapes(cfg)
        missing = [k for k in shapes if k not in params]
        if missing:
            raise KeyError(f"missing parameters: {missing[:5]}{'...' if len(missing) > 5 else ''}")

        def get(name):
            t = params[name]
            if tuple(t.shape) != shapes[name]:
                raise ValueError(f"{name}: expected shape {shapes[name]} (Paddle layout), got {tuple(t.shape)}")
            return t.to(device=dev, dtype=torch.float32)

        bf = lambda t: t.to(_lib.elem_dtype()).contiguous()  # noqa: E731
        w = get("vision_model.embeddings.patch_embedding.weight")
        K = w.shape[1] * w.shape[2] * w.shape[3]
        self._kpad = (K + 7) // 8 * 8                      # 3 * 14 * 14 = 588 -> 592: the GEMM wants K % 8 == 0
        wp = torch.zeros((w.shape[0], self._kpad), device=dev, dtype=torch.float32)
        wp[:, :K] = w.reshape(w.shape[0], K)
        W["patch.w"] = bf(wp)
        pos = get("vision_model.embeddings.position_embedding.weight")
        W["pos_patches"] = bf(pos[1:])                                           # residual input of the patch GEMM
        W["cls_row"] = bf(get("vision_model.embeddings.class_embedding") + pos[0])   # token 0 of every image, constant
        _load_encoder_layers(W, get, bf, "vision_model", cfg["num_hidden_layers"])
        for key, name in (("pre", "vision_model.pre_layrnorm"), ("post", "vision_model.post_layernorm")):
            W[key + ".g"], W[key + ".b"] = get(name + ".weight").contiguous(), get(name + ".bias").contiguous()
        W["proj.w"] = bf(get("visual_projection.weight").t())

    def _build_plan(self, B: int) -> _Plan:
        cfg, lib, dev, W = self.cfg, self._lib, self.device, self.w
        stream = self._stream_ptr
        D, n, p, C = cfg["hidden_size"], cfg["num_hidden_layers"], cfg["patch_size"], cfg["num_channels"]
        side = cfg["image_size"]
        N = (side // p) ** 2
        S, eps, Kp = N + 1, float(cfg["layer_norm_eps"]), self._kpad
        plan = _Plan()
        prog: List[tuple] = []
        keep: List[Tensor] = []

        def persist(shape, dtype) -> Tensor:
            t = torch.empty(shape, device=dev, dtype=dtype)
            keep.append(t)
            return t

        def emit(fn, args, kind, flops=0.0, desc=""):
            prog.append((fn, tuple(args), kind if not desc else f"{kind}:{desc}", flops))

        plan.pixels = persist((B, C, side, side), torch.float32)
        cols = persist((B * N, Kp), _lib.elem_dtype())
        cols.zero_()                                       # the pad columns stay zero; patchify rewrites the first C*p*p
        emb = persist((B * S, D), _lib.elem_dtype())
        emb.reshape(B, S, D)[:, 0].copy_(W["cls_row"])      # class token + position 0: never overwritten
        pos_t = persist((B * N, D), _lib.elem_dtype())
        pos_t.reshape(B, N, D).copy_(W["pos_patches"])
        plan.consts = [cols, emb, pos_t]   # initialised here (paddlemix_amd/export.py ships their contents)
        emit(lib.mi355x_sd_patchify, (plan.pixels.data_ptr(), B, C, side, side, p, cols.data_ptr(), Kp, stream), "misc")
        # patch rows of image b land at token rows b * S + 1 ..: C row remap (rows_per_batch N, batch stride S * D)
        emit(lib.mi355x_sd_linear_ex, (cols.data_ptr(), Kp, 0, 0, W["patch.w"].data_ptr(), None, emb.data_ptr() + 2 * D, D, N,
                                       S * D, B * N, D, Kp, None, None, 0, None, 0, 0, pos_t.data_ptr(), D, 1.0, 0, *self._gemm_ws, stream),
             "gemm", 2.0 * B * N * D * Kp, f"{B * N}x{D}x{Kp}")
        plan.hidden = [persist((B * S, D), _lib.elem_dtype()) for _ in range(n + 1)]
        emit(lib.mi355x_sd_layernorm, (emb.data_ptr(), B * S, D, D, W["pre.g"].data_ptr(), W["pre.b"].data_ptr(), eps,
                                       plan.hidden[0].data_ptr(), D, stream), "ln")
        _emit_encoder_layers(self, emit, persist, plan.hidden, B, S, None)
        plan.pooled = persist((B, D), _lib.elem_dtype())
        # post_layernorm of the class-token rows only: B rows at stride S * D
        emit(lib.mi355x_sd_layernorm, (plan.hidden[n].data_ptr(), B, D, S * D, W["post.g"].data_ptr(), W["post.b"].data_ptr(), eps,
                                       plan.pooled.data_ptr(), D, stream), "ln")
        plan.embeds = persist((B, W["proj.w"].shape[0]), torch.float32)
        emit(lib.mi355x_sd_linear, (plan.pooled.data_ptr(), D, W["proj.w"].data_ptr(), plan.embeds.data_ptr(),
                                    W["proj.w"].shape[0], B, W["proj.w"].shape[0], D, None, None, 0, 0, None, 0, 1.0,
                                    _lib.OUT_F32, *self._gemm_ws, stream), "gemm", 2.0 * B * D * W["proj.w"].shape[0])
        plan.prog, plan.keep, plan.graph = prog, keep, None
        plan.out = plan.embeds
        plan.B, plan.S = B, S
        return plan

    def forward(self, pixel_values: Tensor, output_attentions=None, output_hidden_states: Optional[bool] = None,
                return_dict: Optional[bool] = True):
        """pixel_values [B, num_channels, image_size, image_size] (already CLIP-normalised) -> image_embeds [B, projection_dim]"""
        if pixel_values is None:
            raise ValueError("You have to specify pixel_values")
        if output_attentions:
            raise NotImplementedError("output_attentions is not implemented")
        cfg = self.cfg
        want = (cfg["num_channels"], cfg["image_size"], cfg["image_size"])
        if pixel_values.dim() != 4 or tuple(pixel_values.shape[1:]) != want:
            raise ValueError(f"pixel_values: expected [B, {want[0]}, {want[1]}, {want[2]}], got {tuple(pixel_values.shape)}")
        if not self._emulated and not pixel_values.is_cuda:
            raise _lib.MI355XError("inputs must be GPU tensors (no CPU fallback)")
        B = pixel_values.shape[0]
        if B not in self._plans:
            self._plans[B] = self._build_plan(B)
        plan = self._plans[B]
        if self._emulated:
            plan.pixels.copy_(pixel_values)
            self._run_eager(plan)
        else:
            cur = torch.cuda.current_stream(self.device)
            self._stream.wait_stream(cur)
            with torch.cuda.stream(self._stream):
                plan.pixels.copy_(pixel_values, non_blocking=True)
                self.run(plan)
            cur.wait_stream(self._stream)
        D, S = cfg["hidden_size"], plan.S
        out = CLIPVisionModelOutput(image_embeds=plan.embeds.clone(), last_hidden_state=plan.hidden[-1].reshape(B, S, D).float(),
                                    hidden_states=None)
        if output_hidden_states:
            out.hidden_states = tuple(h.reshape(B, S, D).float() for h in plan.hidden)
        if not return_dict:
            return tuple(v for v in (out.image_embeds, out.last_hidden_state, out.hidden_states) if v is not None)
        return out

    __call__ = forward
