"""TEST INFRASTRUCTURE ONLY -- CPU oracle of ppdiffusers' CLIP text encoder.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this module.

torch-CPU fp32 restatement of PPD/transformers/clip/modeling.py:
  CLIPTextEmbeddings :199-231, CLIPAttention :234-335 (q scaled by head_dim^-0.5, causal mask added to the scores),
  CLIPMLP :338-350, CLIPEncoderLayer :353-400, CLIPEncoder :629-723 (hidden_states tuple = input of every layer + the
  last output), CLIPTextTransformer :726-843 (final_layer_norm, pooled = row of the EOS token: argmax(input_ids) when
  eos_token_id == 2, else first position equal to eos_token_id), CLIPTextModelWithProjection (text_projection, no bias).

Pinned against the reference's own CLIPTextModelWithProjection / CLIPVisionModelWithProjection code (transformers/clip/modeling.py)
executed over oracle/paddle_shim.py (tests/test_reference_modules.py, cases clip_text_*, clip_vision: bit-identical). The reference's
CLIP tests need Paddle and real checkpoints; neither exists here.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

CLIP_DEFAULTS = dict(vocab_size=49408, hidden_size=512, intermediate_size=2048, projection_dim=512,
                     num_hidden_layers=12, num_attention_heads=8, max_position_embeddings=77, hidden_act="quick_gelu",
                     layer_norm_eps=1e-5, eos_token_id=2, with_projection=False)


def normalize_config(config: dict) -> dict:
    cfg = dict(CLIP_DEFAULTS)
    cfg.update({k: v for k, v in config.items() if not k.startswith("_")})
    return cfg


def clip_param_shapes(config: dict) -> Dict[str, tuple]:
    cfg = normalize_config(config)
    D, I = cfg["hidden_size"], cfg["intermediate_size"]
    S: Dict[str, tuple] = {"text_model.embeddings.token_embedding.weight": (cfg["vocab_size"], D),
                           "text_model.embeddings.position_embedding.weight": (cfg["max_position_embeddings"], D)}
    for i in range(cfg["num_hidden_layers"]):
        b = f"text_model.encoder.layers.{i}"
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            S[f"{b}.self_attn.{nm}.weight"], S[f"{b}.self_attn.{nm}.bias"] = (D, D), (D,)
        S[b + ".layer_norm1.weight"], S[b + ".layer_norm1.bias"] = (D,), (D,)
        S[b + ".mlp.fc1.weight"], S[b + ".mlp.fc1.bias"] = (D, I), (I,)
        S[b + ".mlp.fc2.weight"], S[b + ".mlp.fc2.bias"] = (I, D), (D,)
        S[b + ".layer_norm2.weight"], S[b + ".layer_norm2.bias"] = (D,), (D,)
    S["text_model.final_layer_norm.weight"], S["text_model.final_layer_norm.bias"] = (D,), (D,)
    if cfg["with_projection"]:
        S["text_projection.weight"] = (D, cfg["projection_dim"])
    return S


def synth_clip_params(config: dict, seed: int = 1234) -> Params:
    g = torch.Generator().manual_seed(seed)
    P: Params = {}
    for name, shape in clip_param_shapes(config).items():
        r = torch.randn(shape, generator=g)
        if name.endswith(".bias"):
            t = r * 0.02
        elif "embedding" in name:
            t = r * 0.5
        elif len(shape) == 1:
            t = 1.0 + r * 0.02
        else:
            t = r / math.sqrt(shape[0])
        P[name] = t
    return P


def _act(name: str, x: Tensor) -> Tensor:
    if name == "quick_gelu":
        return x * torch.sigmoid(1.702 * x)
    if name == "gelu":
        return F.gelu(x)
    raise ValueError(name)


def clip_text_forward(P: Params, config: dict, input_ids: Tensor) -> dict:
    cfg = normalize_config(config)
    D, H = cfg["hidden_size"], cfg["num_attention_heads"]
    d = D // H
    B, S = input_ids.shape
    lin = lambda n, x: x @ P[n + ".weight"] + P[n + ".bias"]  # noqa: E731  (Paddle Linear: [in, out])
    x = P["text_model.embeddings.token_embedding.weight"][input_ids] + \
        P["text_model.embeddings.position_embedding.weight"][:S][None]
    mask = torch.triu(torch.full((S, S), torch.finfo(torch.float32).min), diagonal=1)
    hidden = [x]
    for i in range(cfg["num_hidden_layers"]):
        b = f"text_model.encoder.layers.{i}"
        h = F.layer_norm(x, (D,), P[b + ".layer_norm1.weight"], P[b + ".layer_norm1.bias"], cfg["layer_norm_eps"])
        q = (lin(b + ".self_attn.q_proj", h) * d ** -0.5).reshape(B, S, H, d).transpose(1, 2)
        k = lin(b + ".self_attn.k_proj", h).reshape(B, S, H, d).transpose(1, 2)
        v = lin(b + ".self_attn.v_proj", h).reshape(B, S, H, d).transpose(1, 2)
        w = torch.softmax(q @ k.transpose(-1, -2) + mask, -1)
        o = (w @ v).transpose(1, 2).reshape(B, S, D)
        x = x + lin(b + ".self_attn.out_proj", o)
        h = F.layer_norm(x, (D,), P[b + ".layer_norm2.weight"], P[b + ".layer_norm2.bias"], cfg["layer_norm_eps"])
        x = x + lin(b + ".mlp.fc2", _act(cfg["hidden_act"], lin(b + ".mlp.fc1", h)))
        hidden.append(x)
    last = F.layer_norm(x, (D,), P["text_model.final_layer_norm.weight"], P["text_model.final_layer_norm.bias"],
                        cfg["layer_norm_eps"])
    if cfg["eos_token_id"] == 2:
        pos = input_ids.argmax(-1)
    else:
        pos = (input_ids == cfg["eos_token_id"]).int().argmax(-1)
    pooled = last[torch.arange(B), pos]
    out = dict(last_hidden_state=last, pooler_output=pooled, hidden_states=tuple(hidden))
    if cfg["with_projection"]:
        out["text_embeds"] = pooled @ P["text_projection.weight"]
    return out


# --------------------------------------------------------------------------
# vision tower: CLIPVisionModelWithProjection (the IP-Adapter image encoder)
#   CLIPVisionEmbeddings modeling.py:162-196 (bias-free patch conv, class token, learned positions),
#   CLIPVisionTransformer :896-953 (pre_layrnorm [sic] -> encoder (no mask) -> post_layernorm of the class-token row),
#   CLIPVisionModelWithProjection :1300-1373 (visual_projection, no bias -> image_embeds)
# --------------------------------------------------------------------------
CLIP_VISION_DEFAULTS = dict(hidden_size=768, intermediate_size=3072, projection_dim=512, num_hidden_layers=12,
                            num_attention_heads=12, num_channels=3, image_size=224, patch_size=32, hidden_act="quick_gelu",
                            layer_norm_eps=1e-5)


def clip_vision_param_shapes(config: dict) -> Dict[str, tuple]:
    cfg = dict(CLIP_VISION_DEFAULTS, **{k: v for k, v in config.items() if not k.startswith("_")})
    D, I, p = cfg["hidden_size"], cfg["intermediate_size"], cfg["patch_size"]
    n_pos = (cfg["image_size"] // p) ** 2 + 1
    S: Dict[str, tuple] = {"vision_model.embeddings.class_embedding": (D,),
                           "vision_model.embeddings.patch_embedding.weight": (D, cfg["num_channels"], p, p),
                           "vision_model.embeddings.position_embedding.weight": (n_pos, D),
                           "vision_model.pre_layrnorm.weight": (D,), "vision_model.pre_layrnorm.bias": (D,)}
    for i in range(cfg["num_hidden_layers"]):
        b = f"vision_model.encoder.layers.{i}"
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            S[f"{b}.self_attn.{nm}.weight"], S[f"{b}.self_attn.{nm}.bias"] = (D, D), (D,)
        S[b + ".layer_norm1.weight"], S[b + ".layer_norm1.bias"] = (D,), (D,)
        S[b + ".mlp.fc1.weight"], S[b + ".mlp.fc1.bias"] = (D, I), (I,)
        S[b + ".mlp.fc2.weight"], S[b + ".mlp.fc2.bias"] = (I, D), (D,)
        S[b + ".layer_norm2.weight"], S[b + ".layer_norm2.bias"] = (D,), (D,)
    S["vision_model.post_layernorm.weight"], S["vision_model.post_layernorm.bias"] = (D,), (D,)
    S["visual_projection.weight"] = (D, cfg["projection_dim"])
    return S


def clip_vision_forward(P: Params, config: dict, pixel_values: Tensor) -> dict:
    cfg = dict(CLIP_VISION_DEFAULTS, **{k: v for k, v in config.items() if not k.startswith("_")})
    D, H, eps = cfg["hidden_size"], cfg["num_attention_heads"], cfg["layer_norm_eps"]
    d = D // H
    B = pixel_values.shape[0]
    lin = lambda n, x: x @ P[n + ".weight"] + P[n + ".bias"]  # noqa: E731
    ln = lambda n, x: F.layer_norm(x, (D,), P[n + ".weight"], P[n + ".bias"], eps)  # noqa: E731
    patches = F.conv2d(pixel_values.float(), P["vision_model.embeddings.patch_embedding.weight"], None,
                       stride=cfg["patch_size"]).flatten(2).transpose(1, 2)
    x = torch.cat([P["vision_model.embeddings.class_embedding"].expand(B, 1, D), patches], dim=1)
    x = x + P["vision_model.embeddings.position_embedding.weight"][None]
    x = ln("vision_model.pre_layrnorm", x)
    S = x.shape[1]
    hidden = [x]
    for i in range(cfg["num_hidden_layers"]):
        b = f"vision_model.encoder.layers.{i}"
        h = ln(b + ".layer_norm1", x)
        q = (lin(b + ".self_attn.q_proj", h) * d ** -0.5).reshape(B, S, H, d).transpose(1, 2)
        k = lin(b + ".self_attn.k_proj", h).reshape(B, S, H, d).transpose(1, 2)
        v = lin(b + ".self_attn.v_proj", h).reshape(B, S, H, d).transpose(1, 2)
        o = (torch.softmax(q @ k.transpose(-1, -2), -1) @ v).transpose(1, 2).reshape(B, S, D)
        x = x + lin(b + ".self_attn.out_proj", o)
        x = x + lin(b + ".mlp.fc2", _act(cfg["hidden_act"], lin(b + ".mlp.fc1", ln(b + ".layer_norm2", x))))
        hidden.append(x)
    pooled = ln("vision_model.post_layernorm", x[:, 0])
    return dict(last_hidden_state=x, pooler_output=pooled, hidden_states=tuple(hidden),
                image_embeds=pooled @ P["visual_projection.weight"])
