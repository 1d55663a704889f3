"""TEST INFRASTRUCTURE ONLY -- CPU oracle of ppdiffusers' AutoencoderKL decode and encode paths.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this module; the product
(``paddlemix_amd/``) never does.

torch-CPU fp32 restatement of
  * ``AutoencoderKL._decode`` / ``decode``       PPD/models/autoencoder_kl.py:288-333 (post_quant_conv :121, :292-293)
  * ``Decoder.__init__`` / ``forward``           PPD/models/vae.py:205-343
  * ``UNetMidBlock2D``                           PPD/models/unet_2d_blocks.py:558-648 (one Attention, heads = C // C = 1,
                                                 residual_connection, group_norm, bias=True, upcast_softmax)
  * ``UpDecoderBlock2D``                         PPD/models/unet_2d_blocks.py:2530-2584 (layers_per_block + 1 resnets,
                                                 Upsample2D with conv)
  * ``ResnetBlock2D`` with ``temb_channels=None``  PPD/models/resnet.py:728-808 (no time_emb_proj)
  * ``AttnProcessor.__call__`` on a 4-D input    PPD/models/attention_processor.py:673-735
and the pipelines' ``latents / vae.config.scaling_factor`` (pipeline_stable_diffusion.py:911); for ``encode``
  * ``AutoencoderKL.encode``                     PPD/models/autoencoder_kl.py:250-283 (quant_conv :120, :274-277)
  * ``Encoder.__init__`` / ``forward``           PPD/models/vae.py:76-180
  * ``DownEncoderBlock2D``                       PPD/models/unet_2d_blocks.py:1301-1367 (layers_per_block resnets, Downsample2D
                                                 with padding=0 -> F.pad (0, 1, 0, 1) + unpadded stride-2 conv, resnet.py:277-279)
  * ``DiagonalGaussianDistribution``             PPD/models/vae.py:744-795

Pinned against the reference's own AutoencoderKL code (decode, encode mean / logvar) executed over oracle/paddle_shim.py
(tests/test_reference_modules.py, case vae_mini: bit-identical). The reference's VAE tests
(ppdiffusers/tests/models/test_models_vae.py) compare against slices produced with Paddle's RNG / real checkpoints, neither of
which exists here.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

from . import unet_ref as U

Tensor = torch.Tensor
Params = Dict[str, Tensor]

VAE_DEFAULTS = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                    layers_per_block=2, norm_num_groups=32, act_fn="silu", scaling_factor=0.18215,
                    use_post_quant_conv=True, use_quant_conv=True, sample_size=512)


def normalize_config(config: dict) -> dict:
    cfg = dict(VAE_DEFAULTS)
    cfg.update({k: v for k, v in config.items() if not k.startswith("_")})
    cfg["block_out_channels"] = tuple(cfg["block_out_channels"])
    if cfg["act_fn"] not in ("silu", "swish"):
        raise ValueError("only act_fn='silu' is restated")
    return cfg


def decoder_param_shapes(config: dict) -> Dict[str, tuple]:
    """Parameter names/shapes of the decode path in construction order (Paddle layouts: Linear [in, out], Conv OIHW)."""
    cfg = normalize_config(config)
    boc, lc = cfg["block_out_channels"], cfg["latent_channels"]
    S: Dict[str, tuple] = {}

    def conv(name, i, o, k):
        S[name + ".weight"] = (o, i, k, k)
        S[name + ".bias"] = (o,)

    def norm(name, c):
        S[name + ".weight"] = (c,)
        S[name + ".bias"] = (c,)

    def lin(name, i, o):
        S[name + ".weight"] = (i, o)
        S[name + ".bias"] = (o,)

    def resnet(name, cin, cout):
        norm(name + ".norm1", cin)
        conv(name + ".conv1", cin, cout, 3)
        norm(name + ".norm2", cout)
        conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(name + ".conv_shortcut", cin, cout, 1)

    if cfg["use_post_quant_conv"]:
        conv("post_quant_conv", lc, lc, 1)
    top = boc[-1]
    conv("decoder.conv_in", lc, top, 3)
    resnet("decoder.mid_block.resnets.0", top, top)
    a = "decoder.mid_block.attentions.0"
    norm(a + ".group_norm", top)
    for nm in ("to_q", "to_k", "to_v", "to_out.0"):
        lin(f"{a}.{nm}", top, top)
    resnet("decoder.mid_block.resnets.1", top, top)
    rev = list(reversed(boc))
    out_c = rev[0]
    for i, c in enumerate(rev):
        prev, out_c = out_c, c
        for j in range(cfg["layers_per_block"] + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else out_c, out_c)
        if i != len(boc) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", out_c, out_c, 3)
    norm("decoder.conv_norm_out", boc[0])
    conv("decoder.conv_out", boc[0], cfg["out_channels"], 3)
    return S


def synth_decoder_params(config: dict, seed: int = 1234, dtype=torch.float32) -> Params:
    """N(0, 1/fan_in) matrices, N(0, 0.02^2) biases, gamma = 1 + N(0, 0.02^2) (SURVEY 8d recipe)."""
    g = torch.Generator().manual_seed(seed)
    P: Params = {}
    for name, shape in decoder_param_shapes(config).items():
        r = torch.randn(shape, generator=g)
        if name.endswith(".bias"):
            t = r * 0.02
        elif len(shape) == 1:
            t = 1.0 + r * 0.02
        elif len(shape) == 2:
            t = r / math.sqrt(shape[0])
        else:
            t = r / math.sqrt(shape[1] * shape[2] * shape[3])
        P[name] = t.to(dtype)
    return P


def resnet_block(P: Params, name: str, x: Tensor, groups: int, eps: float = 1e-6) -> Tensor:
    """ResnetBlock2D.forward with temb=None, output_scale_factor=1 (resnet.py:728-808)."""
    h = F.silu(U.group_norm(P, name + ".norm1", x, groups, eps))
    h = U.conv2d(P, name + ".conv1", h)
    h = F.silu(U.group_norm(P, name + ".norm2", h, groups, eps))
    h = U.conv2d(P, name + ".conv2", h)
    if (name + ".conv_shortcut.weight") in P:
        x = U.conv2d(P, name + ".conv_shortcut", x, padding=0)
    return x + h


def mid_attention(P: Params, name: str, x: Tensor, groups: int, eps: float = 1e-6) -> Tensor:
    """AttnProcessor on [B, C, H, W]: GroupNorm, one head of width C, + residual, / rescale_output_factor (= 1)."""
    B, C, H, W = x.shape
    h = U.group_norm(P, name + ".group_norm", x, groups, eps).reshape(B, C, H * W).transpose(1, 2)
    q, k, v = (U.linear(P, f"{name}.{n}", h) for n in ("to_q", "to_k", "to_v"))
    o = U.sdpa_math(q[:, :, None, :], k[:, :, None, :], v[:, :, None, :])[:, :, 0, :]
    o = U.linear(P, name + ".to_out.0", o)
    return o.transpose(1, 2).reshape(B, C, H, W) + x


def decode(P: Params, config: dict, z: Tensor, scaled: bool = False) -> Tensor:
    """AutoencoderKL.decode(z).sample; ``scaled=True`` first divides by scaling_factor like the pipelines do."""
    cfg = normalize_config(config)
    groups = cfg["norm_num_groups"]
    if scaled:
        z = z / cfg["scaling_factor"]
    if cfg["use_post_quant_conv"]:
        z = U.conv2d(P, "post_quant_conv", z, padding=0)
    x = U.conv2d(P, "decoder.conv_in", z)
    x = resnet_block(P, "decoder.mid_block.resnets.0", x, groups)
    x = mid_attention(P, "decoder.mid_block.attentions.0", x, groups)
    x = resnet_block(P, "decoder.mid_block.resnets.1", x, groups)
    n = len(cfg["block_out_channels"])
    for i in range(n):
        for j in range(cfg["layers_per_block"] + 1):
            x = resnet_block(P, f"decoder.up_blocks.{i}.resnets.{j}", x, groups)
        if i != n - 1:
            x = U.upsample(P, f"decoder.up_blocks.{i}.upsamplers.0", x)
    x = F.silu(U.group_norm(P, "decoder.conv_norm_out", x, groups, 1e-6))
    return U.conv2d(P, "decoder.conv_out", x)


def encoder_param_shapes(config: dict) -> Dict[str, tuple]:
    """Parameter names/shapes of the encode path in construction order (Encoder.__init__, vae.py:76-144)."""
    cfg = normalize_config(config)
    boc, lc = cfg["block_out_channels"], cfg["latent_channels"]
    S: Dict[str, tuple] = {}

    def put(name, wshape):
        S[name + ".weight"] = wshape
        S[name + ".bias"] = (wshape[0] if len(wshape) != 2 else wshape[1],)

    def resnet(name, cin, cout):
        put(name + ".norm1", (cin,))
        put(name + ".conv1", (cout, cin, 3, 3))
        put(name + ".norm2", (cout,))
        put(name + ".conv2", (cout, cout, 3, 3))
        if cin != cout:
            put(name + ".conv_shortcut", (cout, cin, 1, 1))

    put("encoder.conv_in", (boc[0], cfg["in_channels"], 3, 3))
    ch = boc[0]
    for i, c in enumerate(boc):
        for j in range(cfg["layers_per_block"]):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", ch, c)
            ch = c
        if i != len(boc) - 1:
            put(f"encoder.down_blocks.{i}.downsamplers.0.conv", (c, c, 3, 3))
    resnet("encoder.mid_block.resnets.0", ch, ch)
    put("encoder.mid_block.attentions.0.group_norm", (ch,))
    for nm in ("to_q", "to_k", "to_v", "to_out.0"):
        put(f"encoder.mid_block.attentions.0.{nm}", (ch, ch))
    resnet("encoder.mid_block.resnets.1", ch, ch)
    put("encoder.conv_norm_out", (ch,))
    put("encoder.conv_out", (2 * lc, ch, 3, 3))
    if cfg["use_quant_conv"]:
        put("quant_conv", (2 * lc, 2 * lc, 1, 1))
    return S


def encode_moments(P: Params, config: dict, x: Tensor) -> Tensor:
    """quant_conv(Encoder.forward(x)): [B, 2 * latent_channels, H / 2^(n-1), W / 2^(n-1)] (autoencoder_kl.py:266-277)."""
    cfg = normalize_config(config)
    groups, n = cfg["norm_num_groups"], len(cfg["block_out_channels"])
    h = U.conv2d(P, "encoder.conv_in", x)
    for i in range(n):
        for j in range(cfg["layers_per_block"]):
            h = resnet_block(P, f"encoder.down_blocks.{i}.resnets.{j}", h, groups)
        if i != n - 1:
            name = f"encoder.down_blocks.{i}.downsamplers.0.conv"
            h = F.conv2d(F.pad(h, (0, 1, 0, 1), mode="constant", value=0.0), P[name + ".weight"], P[name + ".bias"], stride=2)
    h = resnet_block(P, "encoder.mid_block.resnets.0", h, groups)
    h = mid_attention(P, "encoder.mid_block.attentions.0", h, groups)
    h = resnet_block(P, "encoder.mid_block.resnets.1", h, groups)
    h = U.conv2d(P, "encoder.conv_out", F.silu(U.group_norm(P, "encoder.conv_norm_out", h, groups, 1e-6)))
    if cfg["use_quant_conv"]:
        h = U.conv2d(P, "quant_conv", h, padding=0)
    return h


def posterior(moments: Tensor):
    """DiagonalGaussianDistribution(moments) -> (mean, logvar clipped to [-30, 20], std) (vae.py:745-750)."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    return mean, logvar, torch.exp(0.5 * logvar)


def encode(P: Params, config: dict, x: Tensor, noise: Tensor = None):
    """-> (mean, logvar, sample): ``.latent_dist.mode()``, ``.logvar`` and ``.sample()`` with the given noise (mean if None)."""
    mean, logvar, std = posterior(encode_moments(P, config, x))
    return mean, logvar, (mean if noise is None else mean + std * noise)
