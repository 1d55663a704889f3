"""MI355X-native T5 encoder (``T5EncoderModel``): the third text encoder of Stable Diffusion 3
(``_get_t5_prompt_embeds``, pipelines/stable_diffusion_3/pipeline_stable_diffusion_3.py:213-262).

Mirrors PPD/transformers/t5/modeling.py (T5Stack.forward :946-1113 for the encoder, gated-gelu feed-forward): token
embedding, per block RMS-norm -> fused QKV projection (no bias) -> attention WITHOUT 1/sqrt(d) scaling and with the
shared relative position bias as the attention kernel's additive mask -> output projection + residual -> RMS-norm ->
fused [wi_0 | wi_1] projection -> gelu_new(wi_0) * wi_1 -> wo + residual; final RMS-norm. The relative position bias
[1, heads, S, S] (``compute_bias`` :293-306) is built on the host once per sequence length -- it is a table lookup of
S x S bucket indices. Runs once per prompt; no CPU fallback.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, List, Mapping, Optional

import torch

from . import _lib
from .checkpoint import PretrainedMixin, Table
from .program import DeviceProgram, _Plan

Tensor = torch.Tensor

T5_DEFAULTS = dict(vocab_size=32128, d_model=512, d_kv=64, d_ff=1024, num_layers=8, num_heads=6,
                   relative_attention_num_buckets=32, relative_attention_max_distance=128, layer_norm_epsilon=1e-6,
                   feed_forward_proj="gated-gelu")


def normalize_config(config: Mapping) -> dict:
    cfg = dict(T5_DEFAULTS)
    cfg.update({k: v for k, v in config.items() if not k.startswith("_")})
    if cfg["feed_forward_proj"] != "gated-gelu":
        raise NotImplementedError(f"feed_forward_proj={cfg['feed_forward_proj']!r} (gated-gelu is implemented)")
    if cfg["d_kv"] % 8 or cfg["d_kv"] > 160 or cfg["d_model"] % 8 or cfg["d_model"] > 4096 or cfg["d_ff"] % 8:
        raise ValueError("unsupported geometry: d_kv multiple of 8 and <= 160, d_model <= 4096")
    return cfg


def t5_param_shapes(config: Mapping) -> Dict[str, tuple]:
    """name -> shape (Paddle layouts) of the encoder's parameters; names as in the reference checkpoints."""
    cfg = normalize_config(config)
    D, inner, Fd = cfg["d_model"], cfg["num_heads"] * cfg["d_kv"], cfg["d_ff"]
    S: Dict[str, tuple] = {"shared.weight": Table((cfg["vocab_size"], D))}
    for i in range(cfg["num_layers"]):
        b = f"encoder.block.{i}"
        for n in ("q", "k", "v"):
            S[f"{b}.layer.0.SelfAttention.{n}.weight"] = (D, inner)
        S[f"{b}.layer.0.SelfAttention.o.weight"] = (inner, D)
        if i == 0:
            S[f"{b}.layer.0.SelfAttention.relative_attention_bias.weight"] = Table((cfg["relative_attention_num_buckets"],
                                                                                     cfg["num_heads"]))   # nn.Embedding
        S[f"{b}.layer.0.layer_norm.weight"] = (D,)
        S[f"{b}.layer.1.DenseReluDense.wi_0.weight"] = (D, Fd)
        S[f"{b}.layer.1.DenseReluDense.wi_1.weight"] = (D, Fd)
        S[f"{b}.layer.1.DenseReluDense.wo.weight"] = (Fd, D)
        S[f"{b}.layer.1.layer_norm.weight"] = (D,)
    S["encoder.final_layer_norm.weight"] = (D,)
    return S


def synth_t5_params(config: Mapping, seed: int = 1234, device="cpu") -> Dict[str, Tensor]:
    g = torch.Generator(device=device).manual_seed(seed)
    P: Dict[str, Tensor] = {}
    for name, shape in t5_param_shapes(config).items():
        r = torch.randn(shape, generator=g, device=device)
        if name == "shared.weight":
            t = r
        elif name.endswith("relative_attention_bias.weight"):
            t = r * 0.5
        elif len(shape) == 1:
            t = 1.0 + r * 0.05
        else:
            t = r / shape[0] ** 0.5
        P[name] = t
    return P


def relative_position_bucket(relative_position: Tensor, num_buckets: int = 32, max_distance: int = 128) -> Tensor:
    """bidirectional ``_relative_position_bucket`` (modeling.py:246-291)"""
    num_buckets //= 2
    buckets = (relative_position > 0).long() * num_buckets
    rp = relative_position.abs()
    max_exact = num_buckets // 2
    large = max_exact + (torch.log(rp.float() / max_exact) / math.log(max_distance / max_exact)
                         * (num_buckets - max_exact)).long()
    large = torch.minimum(large, torch.full_like(large, num_buckets - 1))
    return buckets + torch.where(rp < max_exact, rp, large)


class T5EncoderOutput(SimpleNamespace):
    def __getitem__(self, i):
        return (self.last_hidden_state,)[i]


class T5EncoderModel(DeviceProgram, PretrainedMixin):
    _param_shapes = staticmethod(t5_param_shapes)

    def __init__(self, config: Mapping, params: Mapping[str, Tensor], device="cuda", use_graph: bool = True,
                 profile: bool = False):
        self._init_backend(device, use_graph, profile)
        self.cfg = normalize_config(config)
        self.config = SimpleNamespace(**self.cfg)
        self._load_weights(params)

    def _load_weights(self, params: Mapping[str, Tensor]) -> None:
        cfg, dev, W = self.cfg, self.device, self.w
        shapes = t5_param_shapes(cfg)
        if "shared.weight" not in params and "encoder.embed_tokens.weight" in params:   # tied weights (:1120)
            params = dict(params, **{"shared.weight": params["encoder.embed_tokens.weight"]})
        missing = [k for k in shapes if k not in params]
        if missing:
            raise KeyError(f"missing parameters: {missing[:5]}{'...' if len(missing) > 5 else ''}")

        def get(name):
            t = params[name]
            if tuple(t.shape) != shapes[name]:
                raise ValueError(f"{name}: expected shape {shapes[name]} (Paddle layout), got {tuple(t.shape)}")
            return t.to(device=dev, dtype=torch.float32)

        bf = lambda t: t.to(_lib.elem_dtype()).contiguous()  # noqa: E731
        W["tok"] = bf(get("shared.weight"))
        for i in range(cfg["num_layers"]):
            b = f"encoder.block.{i}"
            a = b + ".layer.0.SelfAttention."
            W[f"l{i}.qkv.w"] = bf(torch.cat([get(a + n + ".weight").t() for n in ("q", "k", "v")], 0))
            W[f"l{i}.o.w"] = bf(get(a + "o.weight").t())
            f = b + ".layer.1.DenseReluDense."
            W[f"l{i}.wi.w"] = bf(torch.cat([get(f + "wi_0.weight").t(), get(f + "wi_1.weight").t()], 0))
            W[f"l{i}.wo.w"] = bf(get(f + "wo.weight").t())
            W[f"l{i}.ln0"] = get(b + ".layer.0.layer_norm.weight").contiguous()
            W[f"l{i}.ln1"] = get(b + ".layer.1.layer_norm.weight").contiguous()
        W["lnf"] = get("encoder.final_layer_norm.weight").contiguous()
        W["rel_bias"] = get("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight").contiguous()

    def _position_bias(self, S: int) -> Tensor:
        cfg = self.cfg
        ctx = torch.arange(S, device=self.device)[:, None]
        mem = torch.arange(S, device=self.device)[None, :]
        bucket = relative_position_bucket(mem - ctx, cfg["relative_attention_num_buckets"],
                                          cfg["relative_attention_max_distance"])
        return self.w["rel_bias"][bucket].permute(2, 0, 1).contiguous()   # [heads, S, S] fp32

    def _build_plan(self, B: int, S: int) -> _Plan:
        cfg, lib, dev, W = self.cfg, self._lib, self.device, self.w
        stream = self._stream_ptr
        D, H, dk, Fd, n = cfg["d_model"], cfg["num_heads"], cfg["d_kv"], cfg["d_ff"], cfg["num_layers"]
        inner, eps = H * dk, float(cfg["layer_norm_epsilon"])
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

        def linear(a: Tensor, lda, wkey, out: Tensor, ldc, R: Optional[Tensor] = None):
            w = W[wkey]
            N, K = w.shape
            emit(lib.mi355x_sd_linear, (a.data_ptr(), lda, w.data_ptr(), out.data_ptr(), ldc, rows, N, K, None, None, 0, 0,
                                        R.data_ptr() if R is not None else None, N if R is not None else 0, 1.0, 0, *self._gemm_ws, stream),
                 "gemm", 2.0 * rows * N * K, f"{rows}x{N}x{K}")

        def rms(x: Tensor, wkey, out: Tensor):
            emit(lib.mi355x_sd_rmsnorm, (x.data_ptr(), rows, D, D, W[wkey].data_ptr(), eps, out.data_ptr(), D, stream), "ln")

        plan.ids = persist((rows,), torch.int32)
        bias = persist((H, S, S), torch.float32)
        bias.copy_(self._position_bias(S))
        plan.consts = [bias]       # filled here, read by every run (paddlemix_amd/export.py ships its contents)
        xa, xb = persist((rows, D), _lib.elem_dtype()), persist((rows, D), _lib.elem_dtype())
        h = persist((rows, D), _lib.elem_dtype())
        qkv = persist((rows, 3 * inner), _lib.elem_dtype())
        ao = persist((rows, inner), _lib.elem_dtype())
        wi = persist((rows, 2 * Fd), _lib.elem_dtype())
        ff = persist((rows, Fd), _lib.elem_dtype())
        plan.last = persist((rows, D), _lib.elem_dtype())
        emit(lib.mi355x_sd_embed_tokens, (plan.ids.data_ptr(), rows, S, W["tok"].data_ptr(), None, D, xa.data_ptr(), D,
                                          stream), "misc")
        for i in range(n):
            rms(xa, f"l{i}.ln0", h)
            linear(h, D, f"l{i}.qkv.w", qkv, 3 * inner)
            qp = qkv.data_ptr()
            emit(lib.mi355x_sd_sdpa, (qp, qp + 2 * inner, qp + 4 * inner, bias.data_ptr(), ao.data_ptr(), B, H, S, S, dk,
                                      S * 3 * inner, 3 * inner, S * 3 * inner, 3 * inner, S * 3 * inner, 3 * inner,
                                      S * inner, inner, 0, S * S, S, 1.0, stream),
                 "attn", 4.0 * B * H * S * S * dk, f"{B}x{H}x{S}x{S}x{dk}")
            linear(ao, inner, f"l{i}.o.w", xb, D, R=xa)
            rms(xb, f"l{i}.ln1", h)
            linear(h, D, f"l{i}.wi.w", wi, 2 * Fd)
            emit(lib.mi355x_sd_gated_activation, (wi.data_ptr(), 2 * Fd, ff.data_ptr(), Fd, rows, Fd, 3, stream), "misc")
            linear(ff, Fd, f"l{i}.wo.w", xa, D, R=xb)
        rms(xa, "lnf", plan.last)
        plan.prog, plan.keep, plan.graph = prog, keep, None
        plan.out = plan.last
        return plan

    def forward(self, input_ids: Tensor, attention_mask=None, output_hidden_states=None, return_dict: Optional[bool] = True):
        if input_ids is None:
            raise ValueError("You have to specify input_ids")
        if attention_mask is not None or output_hidden_states:
            raise NotImplementedError("attention_mask / output_hidden_states are not implemented (SD3 passes neither)")
        cfg = self.cfg
        ids = input_ids.reshape(-1, input_ids.shape[-1])
        B, S = ids.shape
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
        last = plan.last.reshape(B, S, cfg["d_model"]).float()
        if not return_dict:
            return (last,)
        return T5EncoderOutput(last_hidden_state=last)

    __call__ = forward
