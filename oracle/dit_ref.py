"""CPU oracle of the class-conditional DiT forward -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

A torch-CPU fp32 restatement of ``Transformer2DModel.forward`` on its patched-input / ``norm_type="ada_norm_zero"``
branch -- the model ``DiTPipeline`` calls once per step (ppdiffusers/ppdiffusers/pipelines/dit/pipeline_dit.py; DiT-XL/2:
28 layers, 16 heads x 72, patch 2, 1000 classes, learned sigma -> out_channels 8). Reference lines:
  * ctor / forward           ppdiffusers/ppdiffusers/models/transformer_2d.py:80-270, 272-509 (patched branch :182-203, :383-385,
                             output :478-503)
  * PatchEmbed + sincos      ppdiffusers/ppdiffusers/models/embeddings.py:67-120, 122-247
  * BasicTransformerBlock    ppdiffusers/ppdiffusers/models/attention.py:230-373 (ctor), :376-490 (forward, ada_norm_zero path)
  * AdaLayerNormZero         ppdiffusers/ppdiffusers/models/normalization.py:50-86
  * CombinedTimestepLabelEmbeddings / LabelEmbedding / TimestepEmbedding   embeddings.py:549-565, 439-477, 250-295
  * FeedForward gelu-approximate   attention.py:600-677, activations.py:62-98
Pinned against the reference's own Transformer2DModel code executed over oracle/paddle_shim.py (tests/test_reference_modules.py,
case dit_mini: 2.6e-7 relative); the reference's DiT tests themselves need Paddle RNG or hosted weights (tests/pipelines/dit/test_dit.py).
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from .unet_ref import get_timestep_embedding, linear

Tensor = torch.Tensor
Params = Dict[str, Tensor]

DIT_DEFAULTS = dict(num_attention_heads=16, attention_head_dim=72, in_channels=4, out_channels=8, num_layers=28,
                    sample_size=32, patch_size=2, num_embeds_ada_norm=1000, activation_fn="gelu-approximate",
                    attention_bias=True, norm_type="ada_norm_zero", norm_elementwise_affine=False, norm_eps=1e-5,
                    cross_attention_dim=None, dropout=0.0)


def normalize_config(config: dict) -> dict:
    cfg = dict(DIT_DEFAULTS)
    cfg.update({k: v for k, v in config.items() if not k.startswith("_")})
    if cfg["norm_type"] != "ada_norm_zero" or cfg["patch_size"] is None:
        raise NotImplementedError("oracle: only the patched ada_norm_zero (DiT) branch of Transformer2DModel is restated")
    if cfg["norm_elementwise_affine"] or cfg["cross_attention_dim"] is not None or not cfg["attention_bias"]:
        raise NotImplementedError("oracle: DiT configuration only (no affine norms, no cross-attention, attention bias)")
    if cfg["activation_fn"] != "gelu-approximate":
        raise NotImplementedError(cfg["activation_fn"])
    cfg["inner_dim"] = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    if cfg["out_channels"] is None:
        cfg["out_channels"] = cfg["in_channels"]
    return cfg


def dit_param_shapes(config: dict) -> Dict[str, tuple]:
    """name -> shape (Paddle layouts: Linear [in, out], Conv OIHW, Embedding [num, dim]) in construction order"""
    cfg = normalize_config(config)
    D, p, n = cfg["inner_dim"], cfg["patch_size"], cfg["num_layers"]
    S: Dict[str, tuple] = {}

    def lin(name, i, o):
        S[name + ".weight"] = (i, o)
        S[name + ".bias"] = (o,)

    S["pos_embed.proj.weight"] = (D, cfg["in_channels"], p, p)
    S["pos_embed.proj.bias"] = (D,)
    for i in range(n):
        b = f"transformer_blocks.{i}"
        lin(b + ".norm1.emb.timestep_embedder.linear_1", 256, D)
        lin(b + ".norm1.emb.timestep_embedder.linear_2", D, D)
        S[b + ".norm1.emb.class_embedder.embedding_table.weight"] = (cfg["num_embeds_ada_norm"] + 1, D)   # + the CFG null class
        lin(b + ".norm1.linear", D, 6 * D)
        for nm in ("to_q", "to_k", "to_v", "to_out.0"):
            lin(b + ".attn1." + nm, D, D)
        lin(b + ".ff.net.0.proj", D, 4 * D)
        lin(b + ".ff.net.2", 4 * D, D)
    lin("proj_out_1", D, 2 * D)
    lin("proj_out_2", D, p * p * cfg["out_channels"])
    return S


def synth_dit_params(config: dict, seed: int = 1234) -> Params:
    g = torch.Generator().manual_seed(seed)
    P: Params = {}
    for name, shape in dit_param_shapes(config).items():
        r = torch.randn(shape, generator=g)
        if name.endswith(".bias"):
            t = r * 0.02
        elif "embedding_table" in name:
            t = r * 0.5
        elif len(shape) == 2:
            t = r / math.sqrt(shape[0])
            if ".norm1.linear" in name or name.startswith("proj_out_1"):
                t = t * 0.3
        else:
            t = r / math.sqrt(shape[1] * shape[2] * shape[3])
        P[name] = t
    return P


def sincos_pos_embed(embed_dim: int, grid: int, base_size: int, interpolation_scale: float = 1.0) -> Tensor:
    """get_2d_sincos_pos_embed (embeddings.py:67-120), [grid*grid, embed_dim]"""
    def one(dim, pos):
        omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)
    g = np.arange(grid, dtype=np.float32) / (grid / base_size) / interpolation_scale
    mesh = np.stack(np.meshgrid(g, g), axis=0).reshape([2, 1, grid, grid])   # w first
    return torch.from_numpy(np.concatenate([one(embed_dim // 2, mesh[0]), one(embed_dim // 2, mesh[1])], axis=1)).float()


def layer_norm(x: Tensor, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), None, None, eps)


def conditioning(P: Params, prefix: str, timestep: Tensor, class_labels: Tensor) -> Tensor:
    """CombinedTimestepLabelEmbeddings.forward (embeddings.py:557-565), eval mode (no label dropout)"""
    t = get_timestep_embedding(timestep, 256, flip_sin_to_cos=True, downscale_freq_shift=1)
    te = linear(P, prefix + ".timestep_embedder.linear_2", F.silu(linear(P, prefix + ".timestep_embedder.linear_1", t)))
    return te + P[prefix + ".class_embedder.embedding_table.weight"][class_labels.to(torch.int64)]


def dit_block(P: Params, name: str, x: Tensor, timestep: Tensor, class_labels: Tensor, heads: int, norm_eps: float) -> Tensor:
    """BasicTransformerBlock.forward, ada_norm_zero path without cross-attention (attention.py:376-490)"""
    emb = linear(P, name + ".norm1.linear", F.silu(conditioning(P, name + ".norm1.emb", timestep, class_labels)))
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = emb.chunk(6, dim=1)
    h = layer_norm(x, 1e-6) * (1 + scale_msa[:, None]) + shift_msa[:, None]          # AdaLayerNormZero (normalization.py:85)
    B, S, D = h.shape
    q, k, v = (linear(P, f"{name}.attn1.{nm}", h).reshape(B, S, heads, D // heads).permute(0, 2, 1, 3) for nm in ("to_q", "to_k", "to_v"))
    a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(D // heads), dim=-1) @ v
    a = linear(P, name + ".attn1.to_out.0", a.permute(0, 2, 1, 3).reshape(B, S, D))
    x = gate_msa[:, None] * a + x
    h = layer_norm(x, norm_eps) * (1 + scale_mlp[:, None]) + shift_mlp[:, None]     # norm3, then the adaLN-Zero modulation (:463-466)
    ff = linear(P, name + ".ff.net.2", F.gelu(linear(P, name + ".ff.net.0.proj", h), approximate="tanh"))
    return gate_mlp[:, None] * ff + x


def dit_forward(P: Params, config: dict, hidden_states: Tensor, timestep, class_labels: Tensor) -> Tensor:
    """Transformer2DModel.forward(hidden_states [B, C, H, W], timestep, class_labels) -> [B, out_channels, H, W]"""
    cfg = normalize_config(config)
    D, p, heads = cfg["inner_dim"], cfg["patch_size"], cfg["num_attention_heads"]
    B, _, H, W = hidden_states.shape
    hp, wp = H // p, W // p
    ts = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1).expand(B)
    # PatchEmbed (embeddings.py:209-247): conv p x p stride p, flatten, + sincos position embedding
    x = F.conv2d(hidden_states.float(), P["pos_embed.proj.weight"], P["pos_embed.proj.bias"], stride=p).flatten(2).transpose(1, 2)
    base = cfg["sample_size"] // p
    if hp != wp:
        raise NotImplementedError("oracle: square latents only (the reference reshapes with int(sqrt(tokens)), :495-496)")
    pos = sincos_pos_embed(D, hp, base, max(cfg["sample_size"] // 64, 1))
    x = x + pos[None]
    for i in range(cfg["num_layers"]):
        x = dit_block(P, f"transformer_blocks.{i}", x, ts, class_labels, heads, cfg["norm_eps"])
    cond = conditioning(P, "transformer_blocks.0.norm1.emb", ts, class_labels)      # :480-482
    shift, scale = linear(P, "proj_out_1", F.silu(cond)).chunk(2, dim=1)
    x = layer_norm(x, 1e-6) * (1 + scale[:, None]) + shift[:, None]
    x = linear(P, "proj_out_2", x)
    C = cfg["out_channels"]
    x = x.reshape(B, hp, wp, p, p, C).permute(0, 5, 1, 3, 2, 4)                       # nhwpqc -> nchpwq (:497-503)
    return x.reshape(B, C, hp * p, wp * p)
