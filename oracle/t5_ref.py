"""TEST INFRASTRUCTURE ONLY -- CPU oracle of ppdiffusers' T5 encoder (third text encoder of SD3).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this module.

torch-CPU fp32 restatement of PPD/transformers/t5/modeling.py for ``T5EncoderModel`` (:1535-1608) with
``feed_forward_proj="gated-gelu"`` (T5 v1.1): T5LayerNorm :86-108 (RMS norm, no bias), T5DenseGatedActDense :144-185
(gelu_new(wi_0 x) * wi_1 x -> wo), T5Attention :205-424 (no 1/sqrt(d) scaling; relative position bias
``_relative_position_bucket`` :246-291 / ``compute_bias`` :293-306 computed in block 0 and shared by every block),
T5LayerSelfAttention :426-453, T5LayerFF :187-203, T5Stack :922-1113 (final_layer_norm).

Pinned against the reference's own T5EncoderModel code (transformers/t5/modeling.py) executed over oracle/paddle_shim.py
(tests/test_reference_modules.py, case t5_encoder: bit-identical). The reference's T5 tests need Paddle and real checkpoints.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

T5_DEFAULTS = dict(vocab_size=32128, d_model=512, d_kv=64, d_ff=1024, num_layers=8, num_heads=6,
                   relative_attention_num_buckets=32, relative_attention_max_distance=128, layer_norm_epsilon=1e-6,
                   feed_forward_proj="gated-gelu")


def normalize_config(config: dict) -> dict:
    cfg = dict(T5_DEFAULTS)
    cfg.update({k: v for k, v in config.items() if not k.startswith("_")})
    if cfg["feed_forward_proj"] != "gated-gelu":
        raise NotImplementedError("only the gated-gelu (T5 v1.1) feed-forward is restated")
    return cfg


def t5_param_shapes(config: dict) -> Dict[str, tuple]:
    cfg = normalize_config(config)
    D, inner, Fd = cfg["d_model"], cfg["num_heads"] * cfg["d_kv"], cfg["d_ff"]
    S: Dict[str, tuple] = {"shared.weight": (cfg["vocab_size"], D)}
    for i in range(cfg["num_layers"]):
        b = f"encoder.block.{i}"
        for n in ("q", "k", "v"):
            S[f"{b}.layer.0.SelfAttention.{n}.weight"] = (D, inner)
        S[f"{b}.layer.0.SelfAttention.o.weight"] = (inner, D)
        if i == 0:
            S[f"{b}.layer.0.SelfAttention.relative_attention_bias.weight"] = (cfg["relative_attention_num_buckets"],
                                                                               cfg["num_heads"])
        S[f"{b}.layer.0.layer_norm.weight"] = (D,)
        S[f"{b}.layer.1.DenseReluDense.wi_0.weight"] = (D, Fd)
        S[f"{b}.layer.1.DenseReluDense.wi_1.weight"] = (D, Fd)
        S[f"{b}.layer.1.DenseReluDense.wo.weight"] = (Fd, D)
        S[f"{b}.layer.1.layer_norm.weight"] = (D,)
    S["encoder.final_layer_norm.weight"] = (D,)
    return S


def synth_t5_params(config: dict, seed: int = 1234) -> Params:
    g = torch.Generator().manual_seed(seed)
    P: Params = {}
    for name, shape in t5_param_shapes(config).items():
        r = torch.randn(shape, generator=g)
        if name == "shared.weight":
            t = r
        elif name.endswith("relative_attention_bias.weight"):
            t = r * 0.5
        elif len(shape) == 1:
            t = 1.0 + r * 0.05
        else:
            t = r / math.sqrt(shape[0])
        P[name] = t
    return P


def relative_position_bucket(relative_position: Tensor, num_buckets: int = 32, max_distance: int = 128) -> Tensor:
    """bidirectional branch of T5Attention._relative_position_bucket (:246-291)"""
    num_buckets //= 2
    buckets = (relative_position > 0).long() * num_buckets
    rp = relative_position.abs()
    max_exact = num_buckets // 2
    is_small = rp < max_exact
    large = max_exact + (torch.log(rp.float() / max_exact) / math.log(max_distance / max_exact)
                         * (num_buckets - max_exact)).long()
    large = torch.minimum(large, torch.full_like(large, num_buckets - 1))
    return buckets + torch.where(is_small, rp, large)


def compute_bias(P: Params, cfg: dict, S: int) -> Tensor:
    ctx = torch.arange(S)[:, None]
    mem = torch.arange(S)[None, :]
    bucket = relative_position_bucket(mem - ctx, cfg["relative_attention_num_buckets"],
                                      cfg["relative_attention_max_distance"])
    w = P["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"]
    return w[bucket].permute(2, 0, 1)[None]   # [1, heads, S, S]


def _rms(x: Tensor, w: Tensor, eps: float) -> Tensor:
    return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))


def t5_encoder_forward(P: Params, config: dict, input_ids: Tensor) -> Tensor:
    """T5EncoderModel(input_ids).last_hidden_state (no attention mask, as SD3's encode_prompt calls it)."""
    cfg = normalize_config(config)
    H, dk, eps = cfg["num_heads"], cfg["d_kv"], cfg["layer_norm_epsilon"]
    B, S = input_ids.shape
    x = P["shared.weight"][input_ids]
    bias = compute_bias(P, cfg, S)
    for i in range(cfg["num_layers"]):
        b = f"encoder.block.{i}"
        h = _rms(x, P[b + ".layer.0.layer_norm.weight"], eps)
        q, k, v = (((h @ P[f"{b}.layer.0.SelfAttention.{n}.weight"]).reshape(B, S, H, dk).transpose(1, 2)) for n in "qkv")
        w = torch.softmax(q @ k.transpose(-1, -2) + bias, -1)   # no scaling (:385-388)
        o = (w @ v).transpose(1, 2).reshape(B, S, H * dk)
        x = x + o @ P[b + ".layer.0.SelfAttention.o.weight"]
        h = _rms(x, P[b + ".layer.1.layer_norm.weight"], eps)
        ff = F.gelu(h @ P[b + ".layer.1.DenseReluDense.wi_0.weight"], approximate="tanh") * \
            (h @ P[b + ".layer.1.DenseReluDense.wi_1.weight"])
        x = x + ff @ P[b + ".layer.1.DenseReluDense.wo.weight"]
    return _rms(x, P["encoder.final_layer_norm.weight"], eps)
