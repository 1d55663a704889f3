"""torch-CPU restatement of ppdiffusers' UNet2DConditionModel forward (oracle).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Pinned against the reference's own
UNet2DConditionModel / ControlNetModel code executed over oracle/paddle_shim.py
(tests/test_reference_modules.py: 23 UNet / ControlNet / IP-Adapter cases, bit-identical
outputs); the sub-pieces with RNG-free known answers are pinned in tests/test_oracle_pins.py.
Unpinned: Paddle's own kernels and real weights (neither exists here).

All paths relative to /root/reference/ppdiffusers/ppdiffusers/ (``PPD/``).
Parameters live in a flat dict keyed by the reference's parameter names
(``down_blocks.0.resnets.0.conv1.weight`` ...) in the reference's *Paddle*
layouts: ``nn.Linear.weight`` is ``[in, out]`` (y = x @ W + b,
PPD/models/modeling_pytorch_paddle_utils.py:27-63), conv weights are OIHW.

Only the branches the SD-1.5 / SDXL / tiny test configs exercise are restated
(SURVEY.md section 8a); every other config value raises NotImplementedError so
that an unsupported model can never silently produce numbers.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# --------------------------------------------------------------------------
# config handling  (PPD/models/unet_2d_condition.py:172-231 ctor defaults)
# --------------------------------------------------------------------------
UNET_DEFAULTS = dict(
    sample_size=None,
    in_channels=4,
    out_channels=4,
    center_input_sample=False,
    flip_sin_to_cos=True,
    freq_shift=0,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    mid_block_type="UNetMidBlock2DCrossAttn",
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    only_cross_attention=False,
    block_out_channels=(320, 640, 1280, 1280),
    layers_per_block=2,
    downsample_padding=1,
    mid_block_scale_factor=1,
    act_fn="silu",
    norm_num_groups=32,
    norm_eps=1e-5,
    cross_attention_dim=1280,
    transformer_layers_per_block=1,
    attention_head_dim=8,
    use_linear_projection=False,
    addition_embed_type=None,
    addition_time_embed_dim=None,
    upcast_attention=False,
    resnet_time_scale_shift="default",
    resnet_out_scale_factor=1.0,
    time_embedding_type="positional",
    projection_class_embeddings_input_dim=None,
    class_embed_type=None,
    time_cond_proj_dim=None,
    num_class_embeds=None,
    class_embeddings_concat=False,
    encoder_hid_dim=None,
    encoder_hid_dim_type=None,
    ip_adapter_num_tokens=4,       # not a reference config field: ImageProjection's num_image_text_embeds, which the reference
                                   # reads off the IP-Adapter checkpoint (loaders/unet.py _load_ip_adapter_weights)
)

_UNSUPPORTED_IF_SET = (
    "time_embedding_dim", "time_embedding_act_fn", "timestep_post_act",
    "cross_attention_norm", "dual_cross_attention", "resnet_skip_time_act",
)


def normalize_config(config: dict) -> dict:
    cfg = dict(UNET_DEFAULTS)
    for k, v in config.items():
        if k.startswith("_"):
            continue
        cfg[k] = v
    for k in _UNSUPPORTED_IF_SET:
        if cfg.get(k) not in (None, False):
            raise NotImplementedError(f"oracle: config field {k}={cfg[k]!r} is outside the restated hot path")
    if cfg["encoder_hid_dim_type"] not in (None, "ip_image_proj"):
        raise NotImplementedError(f"oracle: encoder_hid_dim_type={cfg['encoder_hid_dim_type']!r}")
    if cfg["encoder_hid_dim_type"] is not None and cfg["encoder_hid_dim"] is None:
        raise ValueError(f"`encoder_hid_dim` has to be defined when `encoder_hid_dim_type` is set to {cfg['encoder_hid_dim_type']}.")
    if cfg["time_embedding_type"] != "positional" or cfg["resnet_time_scale_shift"] != "default":
        raise NotImplementedError("oracle: only positional time embedding / default resnet time shift restated")
    if cfg["act_fn"] not in ("silu", "swish"):
        raise NotImplementedError("oracle: only SiLU resnets restated")
    if cfg.get("conv_in_kernel", 3) != 3 or cfg.get("conv_out_kernel", 3) != 3:
        raise NotImplementedError("oracle: 3x3 conv_in / conv_out only")
    if cfg["class_embed_type"] not in (None, "timestep", "identity", "projection", "simple_projection"):
        raise ValueError(f"class_embed_type {cfg['class_embed_type']!r}")
    if cfg["class_embed_type"] in ("projection", "simple_projection") and cfg["projection_class_embeddings_input_dim"] is None:
        raise ValueError(f"`class_embed_type`: '{cfg['class_embed_type']}' requires `projection_class_embeddings_input_dim` be set")
    n = len(cfg["down_block_types"])

    def tup(x):
        return tuple(x) if isinstance(x, (list, tuple)) else (x,) * n

    cfg["block_out_channels"] = tuple(cfg["block_out_channels"])
    cfg["layers_per_block"] = tup(cfg["layers_per_block"])
    cfg["transformer_layers_per_block"] = tup(cfg["transformer_layers_per_block"])
    cfg["cross_attention_dim"] = tup(cfg["cross_attention_dim"])
    # num_attention_heads = attention_head_dim (the naming quirk at unet_2d_condition.py:245)
    cfg["num_attention_heads"] = tup(cfg["attention_head_dim"])
    cfg["only_cross_attention"] = tup(cfg["only_cross_attention"])
    if any(cfg["only_cross_attention"]):
        raise NotImplementedError("oracle: only_cross_attention not restated")
    return cfg


# --------------------------------------------------------------------------
# primitive layers with Paddle semantics
# --------------------------------------------------------------------------
def linear(P: Params, name: str, x: Tensor) -> Tensor:
    """LoRACompatibleLinear.forward without LoRA: F.linear(x, W[in,out], b) (PPD/models/lora.py:453-459)."""
    w = P[name + ".weight"]
    y = x @ w
    b = P.get(name + ".bias")
    return y if b is None else y + b


def conv2d(P: Params, name: str, x: Tensor, stride: int = 1, padding: int = 1) -> Tensor:
    """LoRACompatibleConv.forward without LoRA: F.conv2d OIHW (PPD/models/lora.py:364-377)."""
    return F.conv2d(x, P[name + ".weight"], P.get(name + ".bias"), stride=stride, padding=padding)


def group_norm(P: Params, name: str, x: Tensor, groups: int, eps: float) -> Tensor:
    """paddle.nn.GroupNorm(epsilon): biased variance over (C/G, H, W), affine per channel."""
    return F.group_norm(x, groups, P[name + ".weight"], P[name + ".bias"], eps)


def layer_norm(P: Params, name: str, x: Tensor, eps: float = 1e-5) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), P.get(name + ".weight"), P.get(name + ".bias"), eps)


def get_timestep_embedding(timesteps: Tensor, embedding_dim: int, flip_sin_to_cos: bool = False,
                           downscale_freq_shift: float = 1, scale: float = 1, max_period: int = 10000) -> Tensor:
    """PPD/models/embeddings.py:26-64 -- computed in float32 regardless of model dtype."""
    assert timesteps.ndim == 1
    half_dim = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half_dim, dtype=torch.float32)
    exponent = exponent / (half_dim - downscale_freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].to(torch.float32) * emb[None, :]
    emb = scale * emb
    if flip_sin_to_cos:
        emb = torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)
    else:
        emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if embedding_dim % 2 == 1:
        emb = F.pad(emb, (0, 1))
    return emb


def class_embedding_shapes(cfg: dict, S: dict) -> None:
    """parameters of `class_embedding` (unet_2d_condition.py:354-382), in construction order"""
    boc = cfg["block_out_channels"]
    ted = boc[0] * 4
    ct, pdim = cfg["class_embed_type"], cfg["projection_class_embeddings_input_dim"]
    if ct is None and cfg["num_class_embeds"] is not None:
        S["class_embedding.weight"] = (cfg["num_class_embeds"], ted)          # nn.Embedding
    elif ct in ("timestep", "projection"):                                     # TimestepEmbedding
        i = boc[0] if ct == "timestep" else pdim
        S["class_embedding.linear_1.weight"], S["class_embedding.linear_1.bias"] = (i, ted), (ted,)
        S["class_embedding.linear_2.weight"], S["class_embedding.linear_2.bias"] = (ted, ted), (ted,)
    elif ct == "simple_projection":                                            # nn.Linear
        S["class_embedding.weight"], S["class_embedding.bias"] = (pdim, ted), (ted,)


def class_embedding(P: Params, cfg: dict, emb: Tensor, class_labels) -> Tensor:
    """emb (+|concat) class_embedding(class_labels) -- unet_2d_condition.py:953-975"""
    ct = cfg["class_embed_type"]
    if ct is None and cfg["num_class_embeds"] is None:
        return emb
    if class_labels is None:
        raise ValueError("class_labels should be provided when num_class_embeds > 0")
    if ct is None:
        ce = P["class_embedding.weight"][class_labels.to(torch.int64)]
    elif ct == "timestep":
        t = get_timestep_embedding(class_labels.reshape(-1).to(torch.float32), cfg["block_out_channels"][0],
                                   cfg["flip_sin_to_cos"], cfg["freq_shift"]).to(emb.dtype)
        ce = timestep_embedding_mlp(P, "class_embedding", t)
    elif ct == "identity":
        ce = class_labels.to(emb.dtype)
    elif ct == "projection":
        ce = timestep_embedding_mlp(P, "class_embedding", class_labels.to(emb.dtype))
    else:
        ce = linear(P, "class_embedding", class_labels.to(emb.dtype))
    return torch.cat([emb, ce], dim=-1) if cfg["class_embeddings_concat"] else emb + ce


def timestep_embedding_mlp(P: Params, name: str, x: Tensor) -> Tensor:
    """TimestepEmbedding.forward: linear_1 -> SiLU -> linear_2 (PPD/models/embeddings.py:283-295)."""
    x = linear(P, name + ".linear_1", x)
    x = F.silu(x)
    return linear(P, name + ".linear_2", x)


# --------------------------------------------------------------------------
# attention  (PPD/models/attention_processor.py)
# --------------------------------------------------------------------------
def sdpa_math(q: Tensor, k: Tensor, v: Tensor, attn_mask: Optional[Tensor] = None,
              scale: Optional[float] = None) -> Tensor:
    """The "math" branch of scaled_dot_product_attention_ (PPD/patches/paddle_patch.py:445-461).

    q [B,Sq,h,d], k/v [B,Skv,h,d], attn_mask additive [B,h,Sq,Skv] -> [B,Sq,h,d].
    """
    if scale is None:
        scale = 1.0 / math.sqrt(q.shape[-1])
    qt, kt, vt = q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3)
    s = (qt * scale) @ kt.transpose(-1, -2)
    if attn_mask is not None:
        s = s + attn_mask.to(s.dtype)
    p = torch.softmax(s, dim=-1)
    return (p @ vt).permute(0, 2, 1, 3)


def prepare_attention_mask(mask: Optional[Tensor], target_length: int, batch_size: int, heads: int):
    """Attention.prepare_attention_mask, out_dim=4 (attention_processor.py:587-630) incl. its padding quirk."""
    if mask is None:
        return None
    if mask.shape[-1] != target_length:
        mask = F.pad(mask, (0, target_length), value=0.0)
    if mask.shape[0] < batch_size * heads:
        mask = mask.repeat_interleave(heads, dim=0)
    return mask.reshape(batch_size, heads, -1, mask.shape[-1])


class EncWithIP:
    """encoder_hidden_states with IP-Adapter image tokens appended (unet_2d_condition.py:1054-1061): what the
    IPAdapterAttnProcessor of every cross-attention splits again (attention_processor.py:1857-1862)."""

    def __init__(self, hidden: Tensor, num_tokens: int, scale: float):
        self.hidden, self.num_tokens, self.scale = hidden, num_tokens, scale


def attention(P: Params, name: str, hidden: Tensor, heads: int, encoder_hidden=None,
              attention_mask: Optional[Tensor] = None, processor: str = "math") -> Tensor:
    """Attention.forward through AttnProcessor.__call__ (attention_processor.py:673-735) for the
    3-D input / no group_norm / no norm_cross / residual_connection=False / rescale=1 case.

    processor="math": head_to_batch_dim + get_attention_scores (:532-586);
    processor="sdpa": XFormersAttnProcessor layout [B,S,h,d] + sdpa_ math path (:1167-1249).
    Both are the same arithmetic; the reference pins their equivalence at 1e-3
    (tests/models/test_modeling_common.py:197-256).
    """
    B, Sq, _ = hidden.shape
    ip = None
    if isinstance(encoder_hidden, EncWithIP):           # IPAdapterAttnProcessor.__call__ (attention_processor.py:1819-1901)
        ip = encoder_hidden
        end = ip.hidden.shape[1] - ip.num_tokens
        encoder_hidden, ip_tokens = ip.hidden[:, :end], ip.hidden[:, end:]
    ctx = hidden if encoder_hidden is None else encoder_hidden
    Skv = ctx.shape[1]
    q = linear(P, name + ".to_q", hidden)
    k = linear(P, name + ".to_k", ctx)
    v = linear(P, name + ".to_v", ctx)
    inner = q.shape[-1]
    d = inner // heads
    mask4 = prepare_attention_mask(attention_mask, Skv, B, heads)
    scale = d ** -0.5
    if processor == "math":
        qh = q.reshape(B, Sq, heads, d).permute(0, 2, 1, 3)
        kh = k.reshape(B, Skv, heads, d).permute(0, 2, 1, 3)
        vh = v.reshape(B, Skv, heads, d).permute(0, 2, 1, 3)
        scores = (qh @ kh.transpose(-1, -2)) * scale
        if mask4 is not None:
            scores = scores + mask4
        probs = torch.softmax(scores, dim=-1)
        o = (probs @ vh).permute(0, 2, 1, 3).reshape(B, Sq, inner)
    elif processor == "sdpa":
        o = sdpa_math(q.reshape(B, Sq, heads, d), k.reshape(B, Skv, heads, d), v.reshape(B, Skv, heads, d),
                      attn_mask=mask4, scale=scale).reshape(B, Sq, inner)
    else:
        raise ValueError(processor)
    if ip is not None:                                   # + scale * softmax(q k_ip^T) v_ip, no mask (:1876-1886)
        ki = (ip_tokens @ P[name + ".processor.to_k_ip.weight"]).reshape(B, -1, heads, d).permute(0, 2, 1, 3)
        vi = (ip_tokens @ P[name + ".processor.to_v_ip.weight"]).reshape(B, -1, heads, d).permute(0, 2, 1, 3)
        qh = q.reshape(B, Sq, heads, d).permute(0, 2, 1, 3)
        pi = torch.softmax((qh @ ki.transpose(-1, -2)) * scale, dim=-1)
        o = o + ip.scale * (pi @ vi).permute(0, 2, 1, 3).reshape(B, Sq, inner)
    return linear(P, name + ".to_out.0", o)


def geglu_ff(P: Params, name: str, x: Tensor) -> Tensor:
    """FeedForward.forward with GEGLU (attention.py:670-677, activations.py:101-104): erf-GELU gate."""
    hg = linear(P, name + ".net.0.proj", x)
    h, g = hg.chunk(2, dim=-1)
    x = h * F.gelu(g)
    return linear(P, name + ".net.2", x)


def basic_transformer_block(P: Params, name: str, x: Tensor, heads: int, enc: Tensor,
                            attention_mask=None, encoder_attention_mask=None, processor="math") -> Tensor:
    """BasicTransformerBlock.forward, layer_norm variant (attention.py:376-489); LN eps 1e-5."""
    n = layer_norm(P, name + ".norm1", x)
    x = attention(P, name + ".attn1", n, heads, None, attention_mask, processor) + x
    n = layer_norm(P, name + ".norm2", x)
    x = attention(P, name + ".attn2", n, heads, enc, encoder_attention_mask, processor) + x
    n = layer_norm(P, name + ".norm3", x)
    x = geglu_ff(P, name + ".ff", n) + x
    return x


def transformer_2d(P: Params, name: str, x: Tensor, heads: int, num_layers: int, groups: int,
                   use_linear_projection: bool, enc: Tensor, attention_mask=None, encoder_attention_mask=None,
                   processor="math") -> Tensor:
    """Transformer2DModel.forward, continuous-input branch (transformer_2d.py:351-379, 406-439, 442-468)."""
    B, C, H, W = x.shape
    residual = x
    h = group_norm(P, name + ".norm", x, groups, 1e-6)
    if not use_linear_projection:
        h = conv2d(P, name + ".proj_in", h, padding=0)
        inner = h.shape[1]
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, inner)
    else:
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
        h = linear(P, name + ".proj_in", h)
        inner = h.shape[-1]
    for i in range(num_layers):
        h = basic_transformer_block(P, f"{name}.transformer_blocks.{i}", h, heads, enc,
                                    attention_mask, encoder_attention_mask, processor)
    if not use_linear_projection:
        h = h.reshape(B, H, W, inner).permute(0, 3, 1, 2)
        h = conv2d(P, name + ".proj_out", h, padding=0)
    else:
        h = linear(P, name + ".proj_out", h)
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return h + residual


# --------------------------------------------------------------------------
# resnet / sampling  (PPD/models/resnet.py)
# --------------------------------------------------------------------------
def resnet_block(P: Params, name: str, x: Tensor, temb: Tensor, groups: int, eps: float,
                 output_scale_factor: float = 1.0) -> Tensor:
    """ResnetBlock2D.forward, time_embedding_norm="default" (resnet.py:728-808)."""
    h = group_norm(P, name + ".norm1", x, groups, eps)
    h = F.silu(h)
    h = conv2d(P, name + ".conv1", h)
    t = linear(P, name + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = h + t
    h = group_norm(P, name + ".norm2", h, groups, eps)
    h = F.silu(h)
    h = conv2d(P, name + ".conv2", h)
    if (name + ".conv_shortcut.weight") in P:
        x = conv2d(P, name + ".conv_shortcut", x, padding=0)
    return (x + h) / output_scale_factor


def downsample(P: Params, name: str, x: Tensor, padding: int = 1) -> Tensor:
    """Downsample2D.forward with use_conv=True, name="op" -> param ``.conv`` (resnet.py:271-294)."""
    if padding == 0:
        x = F.pad(x, (0, 1, 0, 1))
    return conv2d(P, name + ".conv", x, stride=2, padding=padding)


def upsample(P: Params, name: str, x: Tensor, output_size=None) -> Tensor:
    """Upsample2D.forward: nearest x2 -- or to `output_size` when the UNet forwards the skip's size -- then conv3x3 (resnet.py:169-218)."""
    if output_size is None:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    else:
        x = F.interpolate(x, size=tuple(int(s) for s in output_size), mode="nearest")
    return conv2d(P, name + ".conv", x)


# --------------------------------------------------------------------------
# UNet2DConditionModel.forward  (PPD/models/unet_2d_condition.py:809-1207)
# --------------------------------------------------------------------------
def unet_forward(P: Params, config: dict, sample: Tensor, timestep, encoder_hidden_states: Tensor,
                 added_cond_kwargs: Optional[dict] = None, attention_mask: Optional[Tensor] = None,
                 encoder_attention_mask: Optional[Tensor] = None, processor: str = "math",
                 taps: Optional[dict] = None, down_block_additional_residuals=None,
                 mid_block_additional_residual: Optional[Tensor] = None, class_labels=None,
                 timestep_cond: Optional[Tensor] = None, ip_adapter_scale: float = 1.0,
                 _controlnet_cond: Optional[Tensor] = None) -> Tensor:
    """Returns the noise prediction [B, out_channels, H, W] (the ``(sample,)`` tuple's first element).

    ``taps``: optional dict that receives named intermediate activations (for layer-wise parity tests).
    """
    cfg = normalize_config(config)
    dtype = P["conv_in.weight"].dtype
    sample = sample.to(dtype)
    B = sample.shape[0]
    groups, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    boc = cfg["block_out_channels"]

    # mask -> additive bias (:921-927)
    if attention_mask is not None:
        attention_mask = ((1 - attention_mask.to(dtype)) * -10000.0).unsqueeze(1)
    if encoder_attention_mask is not None:
        encoder_attention_mask = ((1 - encoder_attention_mask.to(dtype)) * -10000.0).unsqueeze(1)
    if cfg["center_input_sample"]:
        sample = 2 * sample - 1.0

    # time embedding (:933-953): sinusoid in fp32, then cast to model dtype
    if not torch.is_tensor(timestep):
        timesteps = torch.tensor([timestep], dtype=torch.float64 if isinstance(timestep, float) else torch.int64)
    else:
        timesteps = timestep.reshape(-1) if timestep.ndim == 0 else timestep
    timesteps = timesteps.expand(B)
    t_emb = get_timestep_embedding(timesteps, boc[0], cfg["flip_sin_to_cos"], cfg["freq_shift"]).to(dtype)
    if cfg["time_cond_proj_dim"] is not None and timestep_cond is not None:   # embeddings.py:284-285 (LCM guidance embedding)
        t_emb = t_emb + linear(P, "time_embedding.cond_proj", timestep_cond.to(dtype))
    emb = timestep_embedding_mlp(P, "time_embedding", t_emb)
    emb = class_embedding(P, cfg, emb, class_labels)

    if cfg["addition_embed_type"] == "text_time":  # SDXL (:991-1010)
        if added_cond_kwargs is None or "text_embeds" not in added_cond_kwargs:
            raise ValueError("addition_embed_type 'text_time' requires `text_embeds` in `added_cond_kwargs`")
        if "time_ids" not in added_cond_kwargs:
            raise ValueError("addition_embed_type 'text_time' requires `time_ids` in `added_cond_kwargs`")
        text_embeds = added_cond_kwargs["text_embeds"]
        time_ids = added_cond_kwargs["time_ids"]
        time_embeds = get_timestep_embedding(time_ids.flatten(), cfg["addition_time_embed_dim"],
                                             cfg["flip_sin_to_cos"], cfg["freq_shift"])
        time_embeds = time_embeds.reshape(text_embeds.shape[0], -1).to(text_embeds.dtype)
        add_embeds = torch.cat([text_embeds, time_embeds], dim=-1).to(emb.dtype)
        emb = emb + timestep_embedding_mlp(P, "add_embedding", add_embeds)
    elif cfg["addition_embed_type"] is not None:
        raise NotImplementedError(cfg["addition_embed_type"])
    if taps is not None:
        taps["emb"] = emb

    enc = encoder_hidden_states.to(dtype)
    if cfg["encoder_hid_dim_type"] == "ip_image_proj":   # ImageProjection (embeddings.py:507-527) + concat (:1054-1061)
        if not added_cond_kwargs or "image_embeds" not in added_cond_kwargs:
            raise ValueError("UNet2DConditionModel has the config param `encoder_hid_dim_type` set to 'ip_image_proj' which "
                             "requires the keyword argument `image_embeds` to be passed in  `added_conditions`")
        if encoder_attention_mask is not None:
            raise NotImplementedError("oracle: encoder_attention_mask together with IP-Adapter tokens")
        T = cfg["ip_adapter_num_tokens"]
        img = linear(P, "encoder_hid_proj.image_embeds", added_cond_kwargs["image_embeds"].to(dtype))
        img = layer_norm(P, "encoder_hid_proj.norm", img.reshape(B, T, enc.shape[-1]))
        enc = EncWithIP(torch.cat([enc, img], dim=1), T, ip_adapter_scale)
    x = conv2d(P, "conv_in", sample)
    if _controlnet_cond is not None:   # ControlNetModel.forward (controlnet.py:806-811): + ControlNetConditioningEmbedding (:103-113)
        e = F.silu(conv2d(P, "controlnet_cond_embedding.conv_in", _controlnet_cond.to(dtype)))
        nb = sum(1 for k in P if k.startswith("controlnet_cond_embedding.blocks.") and k.endswith(".weight"))
        for j in range(nb):
            e = F.silu(conv2d(P, f"controlnet_cond_embedding.blocks.{j}", e, stride=2 if j % 2 else 1))
        x = x + conv2d(P, "controlnet_cond_embedding.conv_out", e)
    if taps is not None:
        taps["conv_in"] = x

    # down (:1097-1119)
    skips = [x]
    n_blocks = len(cfg["down_block_types"])
    for i, btype in enumerate(cfg["down_block_types"]):
        heads = cfg["num_attention_heads"][i]
        for j in range(cfg["layers_per_block"][i]):
            x = resnet_block(P, f"down_blocks.{i}.resnets.{j}", x, emb, groups, eps)
            if btype == "CrossAttnDownBlock2D":
                x = transformer_2d(P, f"down_blocks.{i}.attentions.{j}", x, heads,
                                   cfg["transformer_layers_per_block"][i], groups, cfg["use_linear_projection"],
                                   enc, attention_mask, encoder_attention_mask, processor)
            elif btype != "DownBlock2D":
                raise NotImplementedError(btype)
            skips.append(x)
        if i != n_blocks - 1:
            x = downsample(P, f"down_blocks.{i}.downsamplers.0", x, cfg["downsample_padding"])
            skips.append(x)
        if taps is not None:
            taps[f"down_{i}"] = x

    # ControlNet residuals on the skip tuple (:1121-1132): added after the down path, the mid block still sees x
    if down_block_additional_residuals is not None:
        if mid_block_additional_residual is None:
            raise NotImplementedError("T2I-adapter form (down residuals without a mid residual)")
        if len(down_block_additional_residuals) != len(skips):
            raise ValueError("one residual per skip tensor is required")
        skips = [s_ + r.to(dtype) for s_, r in zip(skips, down_block_additional_residuals)]

    # mid (:1134-1155)
    if cfg["mid_block_type"] == "UNetMidBlock2DCrossAttn":
        heads = cfg["num_attention_heads"][-1]
        x = resnet_block(P, "mid_block.resnets.0", x, emb, groups, eps, cfg["mid_block_scale_factor"])
        x = transformer_2d(P, "mid_block.attentions.0", x, heads, cfg["transformer_layers_per_block"][-1], groups,
                           cfg["use_linear_projection"], enc, attention_mask, encoder_attention_mask, processor)
        x = resnet_block(P, "mid_block.resnets.1", x, emb, groups, eps, cfg["mid_block_scale_factor"])
    elif cfg["mid_block_type"] is not None:
        raise NotImplementedError(cfg["mid_block_type"])
    if mid_block_additional_residual is not None and down_block_additional_residuals is not None:
        x = x + mid_block_additional_residual.to(dtype)   # (:1151-1155)
    if taps is not None:
        taps["mid"] = x
    if _controlnet_cond is not None:   # zero convolutions on every skip and on the mid output (controlnet.py:842-852)
        downs = tuple(conv2d(P, f"controlnet_down_blocks.{k}", s_, padding=0) for k, s_ in enumerate(skips))
        return downs, conv2d(P, "controlnet_mid_block", x, padding=0)

    # up (:1158-1191)
    forward_upsample_size = any(int(s) % (2 ** (n_blocks - 1)) != 0 for s in sample.shape[-2:])
    rev_heads = tuple(reversed(cfg["num_attention_heads"]))
    rev_layers = tuple(reversed(cfg["layers_per_block"]))
    rev_tlayers = tuple(reversed(cfg["transformer_layers_per_block"]))
    for i, btype in enumerate(cfg["up_block_types"]):
        for j in range(rev_layers[i] + 1):
            skip = skips.pop()
            x = torch.cat([x, skip], dim=1)
            x = resnet_block(P, f"up_blocks.{i}.resnets.{j}", x, emb, groups, eps)
            if btype == "CrossAttnUpBlock2D":
                x = transformer_2d(P, f"up_blocks.{i}.attentions.{j}", x, rev_heads[i], rev_tlayers[i], groups,
                                   cfg["use_linear_projection"], enc, attention_mask, encoder_attention_mask,
                                   processor)
            elif btype != "UpBlock2D":
                raise NotImplementedError(btype)
        if i != n_blocks - 1:
            # latents that are not multiples of 2^(number of upsamplers) (:900-906, :1165-1169): the next skip's size is forced
            x = upsample(P, f"up_blocks.{i}.upsamplers.0", x, skips[-1].shape[2:] if forward_upsample_size else None)
        if taps is not None:
            taps[f"up_{i}"] = x

    # post (:1193-1196)
    x = group_norm(P, "conv_norm_out", x, groups, eps)
    x = F.silu(x)
    x = conv2d(P, "conv_out", x)
    return x


# --------------------------------------------------------------------------
# parameter inventory: name -> shape for a config, mirroring the ctor order
# (PPD/models/unet_2d_condition.py:287-631, unet_2d_blocks.py ctors)
# --------------------------------------------------------------------------
def unet_param_shapes(config: dict) -> Dict[str, tuple]:
    cfg = normalize_config(config)
    boc = cfg["block_out_channels"]
    S: Dict[str, tuple] = {}
    ted = boc[0] * 4

    def lin(name, i, o, bias=True):
        S[name + ".weight"] = (i, o)
        if bias:
            S[name + ".bias"] = (o,)

    def conv(name, i, o, k):
        S[name + ".weight"] = (o, i, k, k)
        S[name + ".bias"] = (o,)

    def norm(name, c):
        S[name + ".weight"] = (c,)
        S[name + ".bias"] = (c,)

    def resnet(name, cin, cout):
        norm(name + ".norm1", cin)
        conv(name + ".conv1", cin, cout, 3)
        lin(name + ".time_emb_proj", ted * (2 if cfg["class_embeddings_concat"] else 1), cout)
        norm(name + ".norm2", cout)
        conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(name + ".conv_shortcut", cin, cout, 1)

    ip = cfg["encoder_hid_dim_type"] == "ip_image_proj"

    def attn(name, dim, cross):
        lin(name + ".to_q", dim, dim, bias=False)
        lin(name + ".to_k", cross, dim, bias=False)
        lin(name + ".to_v", cross, dim, bias=False)
        lin(name + ".to_out.0", dim, dim)
        if ip and name.endswith(".attn2"):             # IPAdapterAttnProcessor sublayer (attention_processor.py:1816-1817)
            lin(name + ".processor.to_k_ip", cross, dim, bias=False)
            lin(name + ".processor.to_v_ip", cross, dim, bias=False)

    def transformer(name, c, layers, cross):
        norm(name + ".norm", c)
        if cfg["use_linear_projection"]:
            lin(name + ".proj_in", c, c)
        else:
            conv(name + ".proj_in", c, c, 1)
        for l in range(layers):
            b = f"{name}.transformer_blocks.{l}"
            norm(b + ".norm1", c)
            attn(b + ".attn1", c, c)
            norm(b + ".norm2", c)
            attn(b + ".attn2", c, cross)
            norm(b + ".norm3", c)
            lin(b + ".ff.net.0.proj", c, 8 * c)
            lin(b + ".ff.net.2", 4 * c, c)
        if cfg["use_linear_projection"]:
            lin(name + ".proj_out", c, c)
        else:
            conv(name + ".proj_out", c, c, 1)

    conv("conv_in", cfg["in_channels"], boc[0], 3)
    lin("time_embedding.linear_1", boc[0], ted)
    if cfg["time_cond_proj_dim"] is not None:   # TimestepEmbedding.cond_proj, no bias (embeddings.py:265-266)
        lin("time_embedding.cond_proj", cfg["time_cond_proj_dim"], boc[0], bias=False)
    lin("time_embedding.linear_2", ted, ted)
    class_embedding_shapes(cfg, S)
    if cfg["addition_embed_type"] == "text_time":
        lin("add_embedding.linear_1", cfg["projection_class_embeddings_input_dim"], ted)
        lin("add_embedding.linear_2", ted, ted)

    n = len(boc)
    out_c = boc[0]
    for i, btype in enumerate(cfg["down_block_types"]):
        in_c, out_c = out_c, boc[i]
        for j in range(cfg["layers_per_block"][i]):
            resnet(f"down_blocks.{i}.resnets.{j}", in_c if j == 0 else out_c, out_c)
            if btype == "CrossAttnDownBlock2D":
                transformer(f"down_blocks.{i}.attentions.{j}", out_c, cfg["transformer_layers_per_block"][i],
                            cfg["cross_attention_dim"][i])
        if i != n - 1:
            conv(f"down_blocks.{i}.downsamplers.0.conv", out_c, out_c, 3)

    if cfg["mid_block_type"] == "UNetMidBlock2DCrossAttn":
        resnet("mid_block.resnets.0", boc[-1], boc[-1])
        transformer("mid_block.attentions.0", boc[-1], cfg["transformer_layers_per_block"][-1],
                    cfg["cross_attention_dim"][-1])
        resnet("mid_block.resnets.1", boc[-1], boc[-1])

    rboc = tuple(reversed(boc))
    rlayers = tuple(reversed(cfg["layers_per_block"]))
    rtl = tuple(reversed(cfg["transformer_layers_per_block"]))
    rcross = tuple(reversed(cfg["cross_attention_dim"]))
    out_c = rboc[0]
    for i, btype in enumerate(cfg["up_block_types"]):
        prev_out, out_c = out_c, rboc[i]
        in_c = rboc[min(i + 1, n - 1)]
        nl = rlayers[i] + 1
        for j in range(nl):
            skip_c = in_c if j == nl - 1 else out_c
            rin = prev_out if j == 0 else out_c
            resnet(f"up_blocks.{i}.resnets.{j}", rin + skip_c, out_c)
            if btype == "CrossAttnUpBlock2D":
                transformer(f"up_blocks.{i}.attentions.{j}", out_c, rtl[i], rcross[i])
        if i != n - 1:
            conv(f"up_blocks.{i}.upsamplers.0.conv", out_c, out_c, 3)

    norm("conv_norm_out", boc[0])
    conv("conv_out", boc[0], cfg["out_channels"], 3)
    if ip:   # ImageProjection (embeddings.py:507-518); listed last so the other parameters keep their synthetic draws
        dx = cfg["cross_attention_dim"][0]
        lin("encoder_hid_proj.image_embeds", cfg["encoder_hid_dim"], cfg["ip_adapter_num_tokens"] * dx)
        norm("encoder_hid_proj.norm", dx)
    return S


def synth_unet_params(config: dict, seed: int = 1234, dtype=torch.float32) -> Params:
    """Synthetic weights per SURVEY.md 8(d): conv/linear N(0, 1/fan_in), biases N(0, 0.02^2),
    norm gamma = 1 + N(0, 0.02^2), beta = N(0, 0.02^2); one Generator seeded ``seed``,
    parameters drawn in ``unet_param_shapes`` order."""
    g = torch.Generator().manual_seed(seed)
    P: Params = {}
    for name, shape in unet_param_shapes(config).items():
        if name.endswith(".bias"):
            t = torch.randn(shape, generator=g) * 0.02
        elif len(shape) == 1:  # norm gamma
            t = 1.0 + torch.randn(shape, generator=g) * 0.02
        elif len(shape) == 2:  # linear [in, out]
            t = torch.randn(shape, generator=g) / math.sqrt(shape[0])
        else:  # conv OIHW
            fan_in = shape[1] * shape[2] * shape[3]
            t = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        P[name] = t.to(dtype)
    return P


# --------------------------------------------------------------------------
# ControlNetModel  (PPD/models/controlnet.py:116-877): the UNet's conv_in / time embedding / down blocks / mid block with the
# conditioning embedding added after conv_in and a 1x1 "zero convolution" on every skip tensor and on the mid output
# --------------------------------------------------------------------------
CONTROLNET_EXTRA_DEFAULTS = dict(conditioning_channels=3, conditioning_embedding_out_channels=(16, 32, 96, 256),
                                 controlnet_conditioning_channel_order="rgb", global_pool_conditions=False)


def _split_controlnet_config(config: dict):
    extra = dict(CONTROLNET_EXTRA_DEFAULTS)
    base = {}
    for k, v in config.items():
        (extra if k in CONTROLNET_EXTRA_DEFAULTS else base)[k] = v
    extra["conditioning_embedding_out_channels"] = tuple(extra["conditioning_embedding_out_channels"])
    return base, extra


def controlnet_param_shapes(config: dict) -> Dict[str, tuple]:
    """construction order of ControlNetModel.__init__ restricted to what forward reads: the UNet's encoder half, then the
    conditioning embedding and the zero convolutions (controlnet.py:262-417)"""
    base, extra = _split_controlnet_config(config)
    cfg = normalize_config(base)
    boc = cfg["block_out_channels"]
    S = {k: v for k, v in unet_param_shapes(base).items()
         if not k.startswith(("up_blocks.", "conv_norm_out.", "conv_out."))}
    ch = extra["conditioning_embedding_out_channels"]
    S["controlnet_cond_embedding.conv_in.weight"], S["controlnet_cond_embedding.conv_in.bias"] = (ch[0], extra["conditioning_channels"], 3, 3), (ch[0],)
    for i in range(len(ch) - 1):
        S[f"controlnet_cond_embedding.blocks.{2 * i}.weight"], S[f"controlnet_cond_embedding.blocks.{2 * i}.bias"] = (ch[i], ch[i], 3, 3), (ch[i],)
        S[f"controlnet_cond_embedding.blocks.{2 * i + 1}.weight"], S[f"controlnet_cond_embedding.blocks.{2 * i + 1}.bias"] = (ch[i + 1], ch[i], 3, 3), (ch[i + 1],)
    S["controlnet_cond_embedding.conv_out.weight"], S["controlnet_cond_embedding.conv_out.bias"] = (boc[0], ch[-1], 3, 3), (boc[0],)
    k, c = 0, boc[0]
    S[f"controlnet_down_blocks.{k}.weight"], S[f"controlnet_down_blocks.{k}.bias"] = (c, c, 1, 1), (c,)
    for i in range(len(boc)):
        c = boc[i]
        for _ in range(cfg["layers_per_block"][i] + (1 if i != len(boc) - 1 else 0)):
            k += 1
            S[f"controlnet_down_blocks.{k}.weight"], S[f"controlnet_down_blocks.{k}.bias"] = (c, c, 1, 1), (c,)
    S["controlnet_mid_block.weight"], S["controlnet_mid_block.bias"] = (boc[-1], boc[-1], 1, 1), (boc[-1],)
    return S


def controlnet_forward(P: Params, config: dict, sample: Tensor, timestep, encoder_hidden_states: Tensor,
                       controlnet_cond: Tensor, conditioning_scale: float = 1.0, guess_mode: bool = False,
                       added_cond_kwargs: Optional[dict] = None, class_labels=None):
    """-> (down_block_res_samples tuple, mid_block_res_sample) (controlnet.py:671-877; scaling :854-869)"""
    base, extra = _split_controlnet_config(config)
    if extra["controlnet_conditioning_channel_order"] == "bgr":
        controlnet_cond = torch.flip(controlnet_cond, dims=[1])
    elif extra["controlnet_conditioning_channel_order"] != "rgb":
        raise ValueError(f"unknown `controlnet_conditioning_channel_order`: {extra['controlnet_conditioning_channel_order']}")
    if extra["global_pool_conditions"]:
        raise NotImplementedError("oracle: global_pool_conditions")
    downs, mid = unet_forward(P, base, sample, timestep, encoder_hidden_states, added_cond_kwargs=added_cond_kwargs,
                              class_labels=class_labels, _controlnet_cond=controlnet_cond)
    if guess_mode:
        scales = torch.logspace(-1, 0, len(downs) + 1) * conditioning_scale
        return tuple(d * sc for d, sc in zip(downs, scales)), mid * scales[-1]
    return tuple(d * conditioning_scale for d in downs), mid * conditioning_scale
