"""MI355X-native AutoencoderKL *decoder*: the ``vae.decode(latents / scaling_factor)`` call that closes every
Stable-Diffusion pipeline (pipeline_stable_diffusion.py:911).

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

Only the decode path exists; ``encode`` raises. There is no CPU fallback.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, List, Mapping, Optional

import torch

from . import _lib
from ._lib import OUT_F32
from .checkpoint import PretrainedMixin
from .program import DeviceProgram, _Plan, _Ref, _V

Tensor = torch.Tensor

VAE_DEFAULTS = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                    layers_per_block=2, norm_num_groups=32, act_fn="silu", scaling_factor=0.18215,
                    use_post_quant_conv=True, sample_size=512, force_upcast=True)
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


def synth_decoder_params(config: Mapping, seed: int = 1234, device="cpu", dtype=torch.float32) -> Dict[str, Tensor]:
    """Random-init parameters (N(0, 1/fan_in) matrices, small biases, gamma ~ 1), drawn on `device`."""
    g = torch.Generator(device=device).manual_seed(seed)
    P: Dict[str, Tensor] = {}
    for name, shape in decoder_param_shapes(config).items():
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


class DecoderOutput(SimpleNamespace):
    """``.sample`` holder (PPD/models/vae.py:40-49)."""


class AutoencoderKL(DeviceProgram, PretrainedMixin):
    _param_shapes = staticmethod(decoder_param_shapes)

    def __init__(self, config: Mapping, params: Mapping[str, Tensor], device="cuda", use_graph: bool = True,
                 profile: bool = False, _test_backend=None):
        """``_test_backend``: test-only injection (tests/abi_emulator.py); never selected by product code."""
        self._init_backend(device, use_graph, profile, _test_backend)
        self.cfg = normalize_config(config)
        self.config = SimpleNamespace(**self.cfg)
        self.use_slicing = False
        self._load_weights(params)

    def enable_slicing(self) -> None:
        """decode one image per launch sequence (autoencoder_kl.py:196-201)"""
        self.use_slicing = True

    def disable_slicing(self) -> None:
        self.use_slicing = False

    def encode(self, *a, **k):
        raise NotImplementedError("AutoencoderKL(mi355x) implements the decode path only (SURVEY.md 8f.1)")

    # ------------------------------------------------------------------ weights
    def _load_weights(self, params: Mapping[str, Tensor]) -> None:
        cfg, dev, W = self.cfg, self.device, self.w
        shapes = decoder_param_shapes(cfg)
        missing = [k for k in shapes if k not in params]
        if missing:
            raise KeyError(f"missing parameters: {missing[:5]}{'...' if len(missing) > 5 else ''}")

        def get(name):
            t = params[name]
            if tuple(t.shape) != shapes[name]:
                raise ValueError(f"{name}: expected shape {shapes[name]}, got {tuple(t.shape)}")
            return t.to(device=dev, dtype=torch.float32)

        bf = lambda t: t.to(_lib.elem_dtype()).contiguous()  # noqa: E731

        def put_conv(key):  # OIHW -> [O][kh][kw][I]
            w = get(key + ".weight")
            W[key + ".w"] = bf(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1))
            W[key + ".b"] = get(key + ".bias").contiguous()

        def put_norm(key):
            W[key + ".g"] = get(key + ".weight").contiguous()
            W[key + ".b"] = get(key + ".bias").contiguous()

        for name in shapes:
            if not name.endswith(".weight"):
                continue
            key = name[:-7]
            if len(shapes[name]) == 1:
                put_norm(key)
            elif len(shapes[name]) == 4 and key not in ("post_quant_conv", "decoder.conv_in"):
                put_conv(key)
        if cfg["use_post_quant_conv"]:
            w = get("post_quant_conv.weight")
            W["post_quant_conv.w"] = bf(w.reshape(w.shape[0], w.shape[1]))
            W["post_quant_conv.b"] = get("post_quant_conv.bias").contiguous()
        w = get("decoder.conv_in.weight")  # -> [ky][kx][ci][O]
        W["decoder.conv_in.w"] = bf(w.permute(2, 3, 1, 0).reshape(-1, w.shape[0]))
        W["decoder.conv_in.b"] = get("decoder.conv_in.bias").contiguous()
        a = "decoder.mid_block.attentions.0"
        for nm in ("to_q", "to_k", "to_v", "to_out.0"):   # Paddle [in, out] -> [out, in]
            W[f"{a}.{nm}.w"] = bf(get(f"{a}.{nm}.weight").t())
        W[a + ".to_q.b"] = get(a + ".to_q.bias").contiguous()
        W[a + ".to_k.b"] = get(a + ".to_k.bias").contiguous()
        # value bias folded through the output projection (uses the bf16-rounded Wo the GEMM multiplies with)
        wo = W[a + ".to_out.0.w"].float()
        W[a + ".to_out.0.b"] = (get(a + ".to_out.0.bias") + wo @ get(a + ".to_v.bias")).contiguous()

    # ------------------------------------------------------------------ plan
    def _build_plan(self, B: int, h: int, w_: int, in_scale: float) -> _Plan:
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
                                        R.ld if R else 0, out_scale, flags, stream), "gemm", 2.0 * M * N * K,
                 f"{M}x{N}x{K}")

        def conv3(x: _V, hh, ww, wkey, out: _V, up=0, R: Optional[_V] = None):
            cout = W[wkey + ".w"].shape[0]
            ho, wo = hh << up, ww << up
            emit(lib.mi355x_sd_conv3x3, (x.p, x.ld, B, hh, ww, x.C, 1, up, wp(wkey + ".w"), out.p, out.ld, cout,
                                         wp(wkey + ".b"), None, 0, R.p if R else None, R.ld if R else 0, 1.0, 0, stream),
                 "conv", 2.0 * B * ho * wo * cout * 9 * x.C, f"{B * ho * wo}x{cout}x{9 * x.C}" + ("up" if up else ""))

        def gnorm(x: _V, hw, nkey, silu) -> _V:
            nws = lib.mi355x_sd_groupnorm_workspace_floats(B, hw, x.C)
            ws = sc("gn_ws", 4 * nws)
            ss = sc("gn_ss", 4 * B * 2 * x.C)
            y = _V(sc("gn", 2 * x.rows * x.C), x.rows, x.C)
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

        # ---- inputs: z -> (1 / scaling_factor, post_quant_conv) -> conv_in ----
        plan.z = persist((B, lc, h, w_), torch.float32)
        plan.out = persist((B, cfg["out_channels"], h << (nlev - 1), w_ << (nlev - 1)), torch.float32)
        top = boc[-1]
        rows = B * h * w_
        x = main_buf(rows, top)
        if cfg["use_post_quant_conv"]:
            zq = persist((B, lc, h, w_), torch.float32)
            emit(lib.mi355x_sd_conv1x1_nchw, (plan.z.data_ptr(), float(in_scale), wp("post_quant_conv.w"),
                                              wp("post_quant_conv.b"), zq.data_ptr(), B, lc, lc, h * w_, stream), "misc")
            emit(lib.mi355x_sd_conv_in3x3, (zq.data_ptr(), None, wp("decoder.conv_in.w"), wp("decoder.conv_in.b"), x.p, B,
                                            lc, h, w_, top, x.ld, stream), "misc")
        else:
            scale_t = persist((1,), torch.float32)
            scale_t.fill_(float(in_scale))
            emit(lib.mi355x_sd_conv_in3x3, (plan.z.data_ptr(), scale_t.data_ptr(), wp("decoder.conv_in.w"),
                                            wp("decoder.conv_in.b"), x.p, B, lc, h, w_, top, x.ld, stream), "misc")

        # ---- mid block ----
        x = resnet(x, "decoder.mid_block.resnets.0", h, w_)
        a = "decoder.mid_block.attentions.0"
        S, C = h * w_, top
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
        x = resnet(x2, "decoder.mid_block.resnets.1", h, w_)

        # ---- up blocks ----
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

    def _get_plan(self, B, h, w_, in_scale) -> _Plan:
        key = (B, h, w_, float(in_scale))
        if key not in self._plans:
            self._plans[key] = self._build_plan(B, h, w_, in_scale)
        return self._plans[key]

    def _slice_batch(self, B: int, h: int, w_: int) -> int:
        if self.use_slicing:
            return 1
        n = len(self.cfg["block_out_channels"])
        widest = max(self.cfg["block_out_channels"][min(1, n - 1)], 1) * (h << (n - 1)) * (w_ << (n - 1))
        return max(1, min(B, _MAX_ELEMS // max(widest, 1)))

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
            if self._emulated:
                plan.z.copy_(zs)
                self._run_eager(plan)
                outs.append(plan.out.clone())
            else:
                cur = torch.cuda.current_stream(self.device)
                self._stream.wait_stream(cur)
                with torch.cuda.stream(self._stream):
                    plan.z.copy_(zs, non_blocking=True)
                    outs.append(self.run(plan).clone())
                cur.wait_stream(self._stream)
        out = outs[0] if len(outs) == 1 else torch.cat(outs, 0)
        if not return_dict:
            return (out,)
        return DecoderOutput(sample=out)
