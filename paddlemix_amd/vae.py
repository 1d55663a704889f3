"""MI355X-native AutoencoderKL: the ``vae.decode(latents / scaling_factor)`` call that closes every Stable-Diffusion
pipeline (pipeline_stable_diffusion.py:911) and the ``vae.encode(image).latent_dist`` call that opens the
img2img / inpaint ones (pipeline_stable_diffusion_img2img.py prepare_latents).

Mirrors ``AutoencoderKL.decode(z, return_dict, generator)`` (PPD/models/autoencoder_kl.py:302-333) and the slicing
switch ``enable_slicing`` (:196-208); the computation is ``Decoder.forward`` (PPD/models/vae.py:282-343) =
conv_in -> UNetMidBlock2D (resnet, 1-head attention, resnet; unet_2d_blocks.py:558-648) -> UpDecoderBlock2D x n
(:2530-2584) -> GroupNorm + SiLU -> conv_out, expressed as the same static program of C-ABI launches as the UNet:
every kernel is the UNet's (implicit-GEMM conv3x3 with the nearest-2x upsample folded into the gather and the shortcut /
residual in the epilogue, GroupNorm statistics + fused normalise+SiLU), plus two small ones -- the NCHW 1x1
post_quant_conv and the fp32 -> bf16 row softmax of the 512-wide single-head mid-block attention, which runs as
GEMM(Q K^T) -> softmax -> GEMM(P V) per image because one head of width C does not fit the fused attention kernel.
V is produced already transposed ([C][S], the [N][K] operand of the P V GEMM) by swapping the operands of its
projection GEMM; its bias is folded into the output projection (softmax rows sum to one: P (X Wv + 1 bv) Wo + bo =
(P X Wv) Wo + (bv Wo + bo)).

``encode`` (autoencoder_kl.py:250-283) is ``Encoder.forward`` (PPD/models/vae.py:146-180) = conv_in -> DownEncoderBlock2D x n
(resnets, then a stride-2 conv padded at the bottom / right only: Downsample2D(padding=0), resnet.py:277-279 =
MI355X_SD_PAD_BR) -> the same mid block -> GroupNorm + SiLU -> conv_out, with quant_conv (a 1x1 conv on conv_out's output, both
linear) folded into conv_out's weights at load time; the moments leave the conv GEMM as fp32 rows and one small kernel turns
them into the DiagonalGaussianDistribution's NCHW mean / clipped logvar / sample (vae.py:744-763). Encoder parameters are
optional: a checkpoint without them gives a decode-only model whose ``encode`` raises.

Tiling (``enable_tiling``) and the training-only ``kl`` / ``nll`` are not built. There is no CPU fallback.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, List, Mapping, Optional

import torch

from . import _lib
from ._lib import OUT_F32, PAD_BR
from .checkpoint import PretrainedMixin, load_pretrained
from .program import DeviceProgram, _Plan, _Ref, _V

Tensor = torch.Tensor

VAE_DEFAULTS = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                    layers_per_block=2, norm_num_groups=32, act_fn="silu", scaling_factor=0.18215,
                    use_post_quant_conv=True, use_quant_conv=True, sample_size=512, force_upcast=True,
                    # read by the pipelines off `vae.config` (autoencoder_kl.py:83-86 defaults): SD3's latent shift, SDXL's per-channel statistics
                    shift_factor=None, latents_mean=None, latents_std=None)
_MAX_ELEMS = 1 << 30   # largest activation (elements) one launch may address; bigger batches are decoded in slices


def normalize_config(config: Mapping) -> dict:
    cfg = dict(VAE_DEFAULTS)
    cfg.update({k: v for k, v in config.items() if not k.startswith("_")})
    cfg["block_out_channels"] = tuple(cfg["block_out_channels"])
    if cfg["act_fn"] not in ("silu", "swish"):
        raise ValueError(f"act_fn {cfg['act_fn']!r} is not supported (silu only)")
    if any(c % 8 for c in cfg["block_out_channels"]) or cfg["out_channels"] > 4 or cfg["latent_channels"] > 16:
        raise ValueError("unsupported decoder geometry: channels must be multiples of 8, out_channels <= 4, "
                         "latent_channels <= 16")
    return cfg


def decoder_param_shapes(config: Mapping) -> Dict[str, tuple]:
    """name -> shape (Paddle layouts) of every parameter the decode path reads, in construction order."""
    cfg = normalize_config(config)
    boc, lc = cfg["block_out_channels"], cfg["latent_channels"]
    S: Dict[str, tuple] = {}

    def conv(name, i, o, k):
        S[name + ".weight"], S[name + ".bias"] = (o, i, k, k), (o,)

    def vec2(name, c):
        S[name + ".weight"], S[name + ".bias"] = (c,), (c,)

    def resnet(name, cin, cout):
        vec2(name + ".norm1", cin)
        conv(name + ".conv1", cin, cout, 3)
        vec2(name + ".norm2", cout)
        conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(name + ".conv_shortcut", cin, cout, 1)

    if cfg["use_post_quant_conv"]:
        conv("post_quant_conv", lc, lc, 1)
    top = boc[-1]
    conv("decoder.conv_in", lc, top, 3)
    resnet("decoder.mid_block.resnets.0", top, top)
    a = "decoder.mid_block.attentions.0"
    vec2(a + ".group_norm", top)
    for nm in ("to_q", "to_k", "to_v", "to_out.0"):
        S[f"{a}.{nm}.weight"], S[f"{a}.{nm}.bias"] = (top, top), (top,)
    resnet("decoder.mid_block.resnets.1", top, top)
    rev = list(reversed(boc))
    out_c = rev[0]
    for i, c in enumerate(rev):
        prev, out_c = out_c, c
        for j in range(cfg["layers_per_block"] + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else out_c, out_c)
        if i != len(boc) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", out_c, out_c, 3)
    vec2("decoder.conv_norm_out", boc[0])
    conv("decoder.conv_out", boc[0], cfg["out_channels"], 3)
    return S


def encoder_param_shapes(config: Mapping) -> Dict[str, tuple]:
    """name -> shape (Paddle layouts) of every parameter the encode path reads (Encoder.__init__, vae.py:76-144; quant_conv,
    autoencoder_kl.py:120), in construction order."""
    cfg = normalize_config(config)
    boc, lc = cfg["block_out_channels"], cfg["latent_channels"]
    S: Dict[str, tuple] = {}

    def conv(name, i, o, k):
        S[name + ".weight"], S[name + ".bias"] = (o, i, k, k), (o,)

    def vec2(name, c):
        S[name + ".weight"], S[name + ".bias"] = (c,), (c,)

    def resnet(name, cin, cout):
        vec2(name + ".norm1", cin)
        conv(name + ".conv1", cin, cout, 3)
        vec2(name + ".norm2", cout)
        conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(name + ".conv_shortcut", cin, cout, 1)

    conv("encoder.conv_in", cfg["in_channels"], boc[0], 3)
    out_c = boc[0]
    for i, c in enumerate(boc):
        prev, out_c = out_c, c
        for j in range(cfg["layers_per_block"]):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", prev if j == 0 else out_c, out_c)
        if i != len(boc) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", out_c, out_c, 3)
    top = boc[-1]
    resnet("encoder.mid_block.resnets.0", top, top)
    a = "encoder.mid_block.attentions.0"
    vec2(a + ".group_norm", top)
    for nm in ("to_q", "to_k", "to_v", "to_out.0"):
        S[f"{a}.{nm}.weight"], S[f"{a}.{nm}.bias"] = (top, top), (top,)
    resnet("encoder.mid_block.resnets.1", top, top)
    vec2("encoder.conv_norm_out", top)
    conv("encoder.conv_out", top, 2 * lc, 3)
    if cfg["use_quant_conv"]:
        conv("quant_conv", 2 * lc, 2 * lc, 1)
    return S


def vae_param_shapes(config: Mapping) -> Dict[str, tuple]:
    """encoder + decoder"""
    return {**encoder_param_shapes(config), **decoder_param_shapes(config)}


def synth_decoder_params(config: Mapping, seed: int = 1234, device="cpu", dtype=torch.float32,
                         shapes=None) -> Dict[str, Tensor]:
    """Random-init parameters (N(0, 1/fan_in) matrices, small biases, gamma ~ 1), drawn on `device`."""
    g = torch.Generator(device=device).manual_seed(seed)
    P: Dict[str, Tensor] = {}
    for name, shape in (shapes or decoder_param_shapes)(config).items():
        r = torch.randn(shape, generator=g, device=device)
        if name.endswith(".bias"):
            t = r * 0.02
        elif len(shape) == 1:
            t = 1.0 + r * 0.02
        elif len(shape) == 2:
            t = r / shape[0] ** 0.5
        else:
            t = r / (shape[1] * shape[2] * shape[3]) ** 0.5
        P[name] = t.to(dtype)
    return P


def synth_vae_params(config: Mapping, seed: int = 1234, device="cpu", dtype=torch.float32) -> Dict[str, Tensor]:
    return synth_decoder_params(config, seed, device, dtype, shapes=vae_param_shapes)


class DecoderOutput(SimpleNamespace):
    """``.sample`` holder (PPD/models/vae.py:40-49)."""


class AutoencoderKLOutput(SimpleNamespace):
    """``.latent_dist`` holder (PPD/models/modeling_outputs.py AutoencoderKLOutput)."""


class DiagonalGaussianDistribution:
    """Posterior of ``AutoencoderKL.encode`` (PPD/models/vae.py:744-795): ``mean`` / ``logvar`` (clipped to [-30, 20]) as
    fp32 [B, L, h, w] GPU tensors, ``mode()`` and ``sample(generator)``. The moments stay on the device as the fp32 rows
    the encoder wrote; ``sample`` is one launch of mi355x_sd_latent_dist over them with freshly drawn noise."""

    def __init__(self, owner: "AutoencoderKL", moments: Tensor, B: int, L: int, h: int, w: int, mean: Tensor, logvar: Tensor):
        self._owner, self._moments, self._shape = owner, moments, (B, L, h, w)
        self.mean, self.logvar = mean, logvar
        self.deterministic = False

    @property
    def std(self) -> Tensor:
        return torch.exp(0.5 * self.logvar)

    @property
    def var(self) -> Tensor:
        return torch.exp(self.logvar)

    def mode(self) -> Tensor:
        return self.mean

    def sample(self, generator=None, *, noise: Optional[Tensor] = None, out_scale: float = 1.0) -> Tensor:
        """mean + std * randn (vae.py:755-763). ``noise`` / ``out_scale`` (extensions): caller-supplied noise, and the
        pipelines' ``* vae.config.scaling_factor`` folded into the same launch."""
        B, L, h, w = self._shape
        dev = self.mean.device
        if noise is None:
            noise = torch.randn(self._shape, generator=generator, device=dev, dtype=torch.float32)
        if tuple(noise.shape) != self._shape:
            raise ValueError(f"noise: expected {self._shape}, got {tuple(noise.shape)}")
        noise = noise.to(device=dev, dtype=torch.float32).contiguous()
        return self._owner._sample_posterior(self._moments, noise, self._shape, out_scale)


class AutoencoderKL(DeviceProgram, PretrainedMixin):
    _param_shapes = staticmethod(decoder_param_shapes)

    def __init__(self, config: Mapping, params: Mapping[str, Tensor], device="cuda", use_graph: bool = True,
                 profile: bool = False):
        self._init_backend(device, use_graph, profile)
        self.cfg = normalize_config(config)
        self.config = SimpleNamespace(**self.cfg)
        self.use_slicing = False
        self._load_weights(params)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, subfolder: Optional[str] = None, **kwargs):
        """encoder + decoder when the checkpoint holds both (every released SD VAE does), decoder alone otherwise"""
        import os
        if not os.path.isdir(pretrained_model_name_or_path):
            raise OSError(f"{pretrained_model_name_or_path} is not a local directory (there is no hub access here)")
        try:
            config, params = load_pretrained(pretrained_model_name_or_path, vae_param_shapes, subfolder)
        except KeyError:
            config, params = load_pretrained(pretrained_model_name_or_path, decoder_param_shapes, subfolder)
        return cls(config, params, **kwargs)

    def enable_slicing(self) -> None:
        """decode one image per launch sequence (autoencoder_kl.py:196-201)"""
        self.use_slicing = True

    def disable_slicing(self) -> None:
        self.use_slicing = False

    def enable_tiling(self, *a, **k):
        raise NotImplementedError("tiled encode / decode (autoencoder_kl.py:335-449) is not built; use enable_slicing")

    # ------------------------------------------------------------------ weights
    def _load_weights(self, params: Mapping[str, Tensor]) -> None:
        cfg, dev, W = self.cfg, self.device, self.w
        dshapes, eshapes = decoder_param_shapes(cfg), encoder_param_shapes(cfg)
        missing = [k for k in dshapes if k not in params]
        if missing:
            raise KeyError(f"missing parameters: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        enc_present = [k for k in eshapes if k in params]
        self.has_encoder = len(enc_present) == len(eshapes)
        if enc_present and not self.has_encoder:
            lost = [k for k in eshapes if k not in params]
            raise KeyError(f"incomplete encoder: missing {lost[:5]}{'...' if len(lost) > 5 else ''}")
        shapes = {**(eshapes if self.has_encoder else {}), **dshapes}

        def get(name):
            t = params[name]
            if tuple(t.shape) != shapes[name]:
                raise ValueError(f"{name}: expected shape {shapes[name]}, got {tuple(t.shape)}")
            return t.to(device=dev, dtype=torch.float32)

        bf = lambda t: t.to(_lib.elem_dtype()).contiguous()  # noqa: E731

        def put_conv(key, w=None, b=None):  # OIHW -> [O][kh][kw][I]
            w = get(key + ".weight") if w is None else w
            W[key + ".w"] = bf(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1))
            W[key + ".b"] = (get(key + ".bias") if b is None else b).contiguous()

        def put_norm(key):
            W[key + ".g"] = get(key + ".weight").contiguous()
            W[key + ".b"] = get(key + ".bias").contiguous()

        special = ("post_quant_conv", "quant_conv", "decoder.conv_in", "encoder.conv_in", "encoder.conv_out")
        for name in shapes:
            if not name.endswith(".weight"):
                continue
            key = name[:-7]
            if len(shapes[name]) == 1:
                put_norm(key)
            elif len(shapes[name]) == 4 and key not in special:
                put_conv(key)
        if cfg["use_post_quant_conv"]:
            w = get("post_quant_conv.weight")
            W["post_quant_conv.w"] = bf(w.reshape(w.shape[0], w.shape[1]))
            W["post_quant_conv.b"] = get("post_quant_conv.bias").contiguous()
        sides = ["decoder"] + (["encoder"] if self.has_encoder else [])
        for side in sides:
            w = get(side + ".conv_in.weight")  # -> [ky][kx][ci][O]
            W[side + ".conv_in.w"] = bf(w.permute(2, 3, 1, 0).reshape(-1, w.shape[0]))
            W[side + ".conv_in.b"] = get(side + ".conv_in.bias").contiguous()
            a = side + ".mid_block.attentions.0"
            for nm in ("to_q", "to_k", "to_v", "to_out.0"):   # Paddle [in, out] -> [out, in]
                W[f"{a}.{nm}.w"] = bf(get(f"{a}.{nm}.weight").t())
            W[a + ".to_q.b"] = get(a + ".to_q.bias").contiguous()
            W[a + ".to_k.b"] = get(a + ".to_k.bias").contiguous()
            # value bias folded through the output projection (uses the bf16-rounded Wo the GEMM multiplies with)
            wo = W[a + ".to_out.0.w"].float()
            W[a + ".to_out.0.b"] = (get(a + ".to_out.0.bias") + wo @ get(a + ".to_v.bias")).contiguous()
        if self.has_encoder:
            # quant_conv o conv_out: both linear, so moments = (Wq Wc) * x + (Wq bc + bq) (autoencoder_kl.py:274-277)
            wc, bc = get("encoder.conv_out.weight"), get("encoder.conv_out.bias")
            if cfg["use_quant_conv"]:
                wq = get("quant_conv.weight").reshape(wc.shape[0], wc.shape[0])
                bc = wq @ bc + get("quant_conv.bias")
                wc = torch.einsum("om,mikl->oikl", wq, wc)
            put_conv("encoder.conv_out", wc, bc)

    # ------------------------------------------------------------------ plan
    def _build_plan(self, B: int, h: int, w_: int, in_scale: float, mode: str = "decode") -> _Plan:
        """``mode="decode"``: h x w is the latent grid; ``mode="encode"``: the image."""
        cfg, lib, dev, W = self.cfg, self._lib, self.device, self.w
        stream = self._stream_ptr
        boc, lc, groups = cfg["block_out_channels"], cfg["latent_channels"], cfg["norm_num_groups"]
        nlev = len(boc)
        plan = _Plan()
        prog: List[tuple] = []
        scratch: Dict[str, int] = {}
        keep: List[Tensor] = []

        def sc(name, nbytes):
            scratch[name] = max(scratch.get(name, 0), nbytes)
            return _Ref(name)

        def persist(shape, dtype) -> Tensor:
            t = torch.empty(shape, device=dev, dtype=dtype)
            keep.append(t)
            return t

        def wp(key):
            return W[key].data_ptr()

        def emit(fn, args, kind, flops=0.0, desc=""):
            prog.append((fn, list(args), kind if not desc else f"{kind}:{desc}", flops))

        def gemm(a_p, lda, w_p, c_p, ldc, M, N, K, bias=None, R: Optional[_V] = None, out_scale=1.0, flags=0):
            emit(lib.mi355x_sd_linear, (a_p, lda, w_p, c_p, ldc, M, N, K, bias, None, 0, 0, R.p if R else None,
                                        R.ld if R else 0, out_scale, flags, *self._gemm_ws, stream), "gemm", 2.0 * M * N * K,
                 f"{M}x{N}x{K}")

        def conv3(x: _V, hh, ww, wkey, out: _V, up=0, R: Optional[_V] = None, stride=1, flags=0):
            cout = W[wkey + ".w"].shape[0]
            pad2 = 1 if flags & PAD_BR else 2
            ho, wo = ((hh << up) + pad2 - 3) // stride + 1, ((ww << up) + pad2 - 3) // stride + 1
            emit(lib.mi355x_sd_conv3x3, (x.p, x.ld, B, hh, ww, x.C, stride, up, wp(wkey + ".w"), out.p, out.ld, cout,
                                         wp(wkey + ".b"), None, 0, R.p if R else None, R.ld if R else 0, 1.0, flags,
                                         *self._gemm_ws, stream),
                 "conv", 2.0 * B * ho * wo * cout * 9 * x.C,
                 f"{B * ho * wo}x{cout}x{9 * x.C}" + ("up" if up else "") + ("s2" if stride == 2 else ""))

        def gnorm(x: _V, hw, nkey, silu) -> _V:
            nws = lib.mi355x_sd_groupnorm_workspace_floats(B, hw, x.C)
            ws = sc("gn_ws", 4 * nws)
            ss = sc("gn_ss", 4 * B * 2 * x.C)
            y = _V(sc("gn", 2 * x.rows * x.C), x.rows, x.C)
            if lib.mi355x_sd_groupnorm_act_fits(hw, x.C, groups):   # small (batch, group) chunks: one launch (csrc/norm.hip gn_fused_kernel)
                emit(lib.mi355x_sd_groupnorm_act, (x.p, B, hw, x.C, x.ld, groups, 1e-6, wp(nkey + ".g"), wp(nkey + ".b"),
                                                   1 if silu else 0, y.p, y.ld, stream), "gn_fused")
                return y
            emit(lib.mi355x_sd_groupnorm_stats, (x.p, B, hw, x.C, x.ld, groups, 1e-6, wp(nkey + ".g"), wp(nkey + ".b"),
                                                 ws, ss, stream), "gn_stats")
            emit(lib.mi355x_sd_scale_shift_act, (x.p, B, hw, x.C, x.ld, ss, 1 if silu else 0, y.p, y.ld, stream),
                 "gn_apply")
            return y

        flip = [0]

        def main_buf(rows, C) -> _V:   # the two alternating trunk buffers
            flip[0] ^= 1
            return _V(sc(f"x{flip[0]}", 2 * rows * C), rows, C)

        def resnet(x: _V, name, hh, ww) -> _V:
            cout = W[name + ".conv1.w"].shape[0]
            g1 = gnorm(x, hh * ww, name + ".norm1", True)
            h1 = _V(sc("h1", 2 * x.rows * cout), x.rows, cout)
            conv3(g1, hh, ww, name + ".conv1", h1)
            g2 = gnorm(h1, hh * ww, name + ".norm2", True)
            R = x
            if (name + ".conv_shortcut.w") in W:
                R = _V(sc("sc", 2 * x.rows * cout), x.rows, cout)
                gemm(x.p, x.ld, wp(name + ".conv_shortcut.w"), R.p, R.ld, x.rows, cout, x.C,
                     bias=wp(name + ".conv_shortcut.b"))
            out = main_buf(x.rows, cout)
            conv3(g2, hh, ww, name + ".conv2", out, R=R)
            return out

        def mid_block(x: _V, side, hh, ww) -> _V:
            """UNetMidBlock2D: resnet, one attention head of width C over the hh*ww pixels, resnet"""
            x = resnet(x, side + ".mid_block.resnets.0", hh, ww)
            a = side + ".mid_block.attentions.0"
            S, C, rows = hh * ww, x.C, x.rows
            y = gnorm(x, S, a + ".group_norm", False)
            q = _V(sc("att_q", 2 * rows * C), rows, C)
            k = _V(sc("att_k", 2 * rows * C), rows, C)
            o = _V(sc("att_o", 2 * rows * C), rows, C)
            gemm(y.p, y.ld, wp(a + ".to_q.w"), q.p, q.ld, rows, C, C, bias=wp(a + ".to_q.b"))
            gemm(y.p, y.ld, wp(a + ".to_k.w"), k.p, k.ld, rows, C, C, bias=wp(a + ".to_k.b"))
            vt = sc("att_vt", 2 * C * S)
            scores = sc("att_s", 4 * S * S)
            probs = sc("att_p", 2 * S * S)
            for b in range(B):
                r0 = 2 * b * S * C   # byte offset of image b's rows
                gemm(wp(a + ".to_v.w"), C, y.p + r0, vt, S, C, S, C)                     # V^T [C][S] = Wv X^T
                gemm(q.p + r0, q.ld, k.p + r0, scores, S, S, S, C, out_scale=C ** -0.5, flags=OUT_F32)
                emit(lib.mi355x_sd_softmax_rows, (scores, S, probs, S, S, S, stream), "attn_softmax")
                gemm(probs, S, vt, o.p + r0, o.ld, S, C, S)
            x2 = main_buf(rows, C)
            gemm(o.p, o.ld, wp(a + ".to_out.0.w"), x2.p, x2.ld, rows, C, C, bias=wp(a + ".to_out.0.b"), R=x)
            return resnet(x2, side + ".mid_block.resnets.1", hh, ww)

        if mode == "encode":
            # ---- image -> conv_in -> down blocks -> mid -> norm/act -> (quant_conv o conv_out) -> posterior ----
            plan.x = persist((B, cfg["in_channels"], h, w_), torch.float32)
            hh, ww = h, w_
            x = main_buf(B * hh * ww, boc[0])
            emit(lib.mi355x_sd_conv_in3x3, (plan.x.data_ptr(), None, wp("encoder.conv_in.w"), wp("encoder.conv_in.b"), x.p, B,
                                            cfg["in_channels"], hh, ww, boc[0], x.ld, stream), "misc")
            for i in range(nlev):
                for j in range(cfg["layers_per_block"]):
                    x = resnet(x, f"encoder.down_blocks.{i}.resnets.{j}", hh, ww)
                if i != nlev - 1:
                    ho, wo = (hh + 1 - 3) // 2 + 1, (ww + 1 - 3) // 2 + 1
                    out = main_buf(B * ho * wo, x.C)
                    conv3(x, hh, ww, f"encoder.down_blocks.{i}.downsamplers.0.conv", out, stride=2, flags=PAD_BR)
                    x, hh, ww = out, ho, wo
            x = mid_block(x, "encoder", hh, ww)
            g = gnorm(x, hh * ww, "encoder.conv_norm_out", True)
            plan.moments = persist((B * hh * ww, 2 * lc), torch.float32)
            conv3(g, hh, ww, "encoder.conv_out", _V(plan.moments.data_ptr(), B * hh * ww, 2 * lc), flags=OUT_F32)
            plan.mean = persist((B, lc, hh, ww), torch.float32)
            plan.logvar = persist((B, lc, hh, ww), torch.float32)
            emit(lib.mi355x_sd_latent_dist, (plan.moments.data_ptr(), 2 * lc, B, lc, hh * ww, None, 1.0,
                                             plan.mean.data_ptr(), plan.logvar.data_ptr(), None, stream), "misc")
            plan.out = plan.mean
            plan.latent_hw = (hh, ww)
        else:
            # ---- z -> (1 / scaling_factor, post_quant_conv) -> conv_in -> mid -> up blocks -> norm/act -> conv_out ----
            plan.z = persist((B, lc, h, w_), torch.float32)
            plan.out = persist((B, cfg["out_channels"], h << (nlev - 1), w_ << (nlev - 1)), torch.float32)
            top = boc[-1]
            x = main_buf(B * h * w_, top)
            if cfg["use_post_quant_conv"]:
                zq = persist((B, lc, h, w_), torch.float32)
                emit(lib.mi355x_sd_conv1x1_nchw, (plan.z.data_ptr(), float(in_scale), wp("post_quant_conv.w"),
                                                  wp("post_quant_conv.b"), zq.data_ptr(), B, lc, lc, h * w_, stream), "misc")
                emit(lib.mi355x_sd_conv_in3x3, (zq.data_ptr(), None, wp("decoder.conv_in.w"), wp("decoder.conv_in.b"), x.p, B,
                                                lc, h, w_, top, x.ld, stream), "misc")
            else:
                scale_t = persist((1,), torch.float32)
                scale_t.fill_(float(in_scale))
                plan.consts = [scale_t]   # filled here (paddlemix_amd/export.py ships its contents)
                emit(lib.mi355x_sd_conv_in3x3, (plan.z.data_ptr(), scale_t.data_ptr(), wp("decoder.conv_in.w"),
                                                wp("decoder.conv_in.b"), x.p, B, lc, h, w_, top, x.ld, stream), "misc")
            x = mid_block(x, "decoder", h, w_)
            hh, ww = h, w_
            for i in range(nlev):
                for j in range(cfg["layers_per_block"] + 1):
                    x = resnet(x, f"decoder.up_blocks.{i}.resnets.{j}", hh, ww)
                if i != nlev - 1:
                    out = main_buf(4 * x.rows, x.C)
                    conv3(x, hh, ww, f"decoder.up_blocks.{i}.upsamplers.0.conv", out, up=1)
                    x, hh, ww = out, hh * 2, ww * 2
            g = gnorm(x, hh * ww, "decoder.conv_norm_out", True)
            emit(lib.mi355x_sd_conv_out3x3, (g.p, g.ld, wp("decoder.conv_out.w"), wp("decoder.conv_out.b"),
                                             plan.out.data_ptr(), B, g.C, hh, ww, cfg["out_channels"], stream), "misc")

        bufs = {nm: persist((max(nb, 16),), torch.uint8) for nm, nb in scratch.items()}
        base = {nm: t.data_ptr() for nm, t in bufs.items()}
        res = lambda v: base[v.buf] + v.off if isinstance(v, _Ref) else v  # noqa: E731
        plan.prog = [(fn, tuple(res(v) for v in args), kind, fl) for fn, args, kind, fl in prog]
        plan.keep, plan.graph = keep, None
        plan.B = B
        return plan

    def _get_plan(self, B, h, w_, in_scale, mode: str = "decode") -> _Plan:
        key = (mode, B, h, w_, float(in_scale))
        if key not in self._plans:
            self._plans[key] = self._build_plan(B, h, w_, in_scale, mode)
        return self._plans[key]

    def _slice_batch(self, B: int, h: int, w_: int, mode: str = "decode") -> int:
        if self.use_slicing:
            return 1
        n = len(self.cfg["block_out_channels"])
        if mode == "encode":
            widest = self.cfg["block_out_channels"][0] * h * w_
        else:
            widest = max(self.cfg["block_out_channels"][min(1, n - 1)], 1) * (h << (n - 1)) * (w_ << (n - 1))
        return max(1, min(B, _MAX_ELEMS // max(widest, 1)))

    def _launch(self, plan: _Plan, stage) -> None:
        """stage the inputs and run the plan on the model's stream, ordered after / before the caller's stream"""
        if self._emulated:
            stage(False)
            self._run_eager(plan)
            return
        cur = torch.cuda.current_stream(self.device)
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            stage(True)
            self.run(plan)
        cur.wait_stream(self._stream)

    def decode(self, z: Tensor, return_dict: bool = True, generator=None, *, in_scale: float = 1.0):
        """z [B, latent_channels, h, w] fp32 -> DecoderOutput(sample [B, out_channels, 8h, 8w] fp32).

        ``in_scale`` (extension): multiplies z inside the first kernel, so ``decode(latents, in_scale=1 / scaling_factor)``
        is the pipelines' ``decode(latents / scaling_factor)`` without the extra pass."""
        if z.dim() != 4 or z.shape[1] != self.cfg["latent_channels"]:
            raise ValueError(f"z: expected [B, {self.cfg['latent_channels']}, h, w], got {tuple(z.shape)}")
        if not self._emulated and not z.is_cuda:
            raise _lib.MI355XError("inputs must be GPU tensors (no CPU fallback)")
        B, _, h, w_ = z.shape
        if (h * w_) % 8:
            raise ValueError("h * w of the latent must be a multiple of 8 (mid-block attention GEMM alignment)")
        step = self._slice_batch(B, h, w_)
        outs = []
        for s in range(0, B, step):
            zs = z[s:s + step]
            plan = self._get_plan(zs.shape[0], h, w_, in_scale)
            self._launch(plan, lambda nb, plan=plan, zs=zs: plan.z.copy_(zs, non_blocking=nb))
            outs.append(plan.out.clone())
        out = outs[0] if len(outs) == 1 else torch.cat(outs, 0)
        if not return_dict:
            return (out,)
        return DecoderOutput(sample=out)

    def encode(self, x: Tensor, return_dict: bool = True):
        """x [B, in_channels, H, W] fp32 image in [-1, 1] -> AutoencoderKLOutput(latent_dist=DiagonalGaussianDistribution)
        (autoencoder_kl.py:250-283). ``enable_slicing`` encodes one image per launch sequence (:271-273)."""
        if not self.has_encoder:
            raise _lib.MI355XError("this AutoencoderKL was built without encoder parameters (decode-only checkpoint)")
        if x.dim() != 4 or x.shape[1] != self.cfg["in_channels"]:
            raise ValueError(f"x: expected [B, {self.cfg['in_channels']}, H, W], got {tuple(x.shape)}")
        if not self._emulated and not x.is_cuda:
            raise _lib.MI355XError("inputs must be GPU tensors (no CPU fallback)")
        B, _, H, Wd = x.shape
        n = len(self.cfg["block_out_channels"])
        if H % (1 << (n - 1)) or Wd % (1 << (n - 1)) or ((H >> (n - 1)) * (Wd >> (n - 1))) % 8:
            raise ValueError(f"H and W must be multiples of {1 << (n - 1)} with (H * W) / {1 << (2 * n - 2)} a multiple of 8 "
                             "(mid-block attention GEMM alignment)")
        step = self._slice_batch(B, H, Wd, "encode")
        L = self.cfg["latent_channels"]
        moments, means, logvars = [], [], []
        for s in range(0, B, step):
            xs = x[s:s + step].to(torch.float32)
            plan = self._get_plan(xs.shape[0], H, Wd, 1.0, "encode")
            self._launch(plan, lambda nb, plan=plan, xs=xs: plan.x.copy_(xs, non_blocking=nb))
            moments.append(plan.moments.clone())
            means.append(plan.mean.clone())
            logvars.append(plan.logvar.clone())
        cat = lambda ts: ts[0] if len(ts) == 1 else torch.cat(ts, 0)  # noqa: E731
        h, w_ = plan.latent_hw
        post = DiagonalGaussianDistribution(self, cat(moments), B, L, h, w_, cat(means), cat(logvars))
        if not return_dict:
            return (post,)
        return AutoencoderKLOutput(latent_dist=post)

    def _sample_posterior(self, moments: Tensor, noise: Tensor, shape, out_scale: float) -> Tensor:
        B, L, h, w_ = shape
        mean = torch.empty(shape, device=moments.device, dtype=torch.float32)
        logvar, sample = torch.empty_like(mean), torch.empty_like(mean)

        def go(_nb):
            rc = self._lib.mi355x_sd_latent_dist(moments.data_ptr(), moments.stride(0), B, L, h * w_, noise.data_ptr(),
                                                 float(out_scale), mean.data_ptr(), logvar.data_ptr(), sample.data_ptr(),
                                                 self._stream_ptr)
            if rc:
                _lib.check(rc)

        if self._emulated:
            go(False)
        else:
            cur = torch.cuda.current_stream(self.device)
            self._stream.wait_stream(cur)
            with torch.cuda.stream(self._stream):
                go(True)
            cur.wait_stream(self._stream)
        return sample
