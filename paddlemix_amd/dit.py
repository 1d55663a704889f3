"""MI355X-native class-conditional DiT: drop-in for the transformer callable of ``DiTPipeline``.

Mirrors ``Transformer2DModel.forward(hidden_states, timestep=..., class_labels=..., return_dict=...)`` on its
patched-input / ``norm_type="ada_norm_zero"`` branch (ppdiffusers/ppdiffusers/models/transformer_2d.py:272-509; called at
pipelines/dit/pipeline_dit.py:192-195). DiT-XL/2: 28 blocks, 16 heads x 72 (D = 1152), patch 2, 1000 classes + the CFG
null class, out_channels 8 (learned sigma). Built from the kernels of the SD3 path:

  * PatchEmbed (embeddings.py:209-247) = ``mi355x_sd_patchify`` + one GEMM whose residual input is the sincos table;
  * every block carries its OWN conditioning embedder (AdaLayerNormZero.emb, normalization.py:59-66): the 256-wide sinusoid
    feeds ONE GEMM for all blocks' ``timestep_embedder.linear_1`` (+SiLU), the class rows of all blocks come from ONE
    gather over the concatenated tables, then per block linear_2 (+ class row as residual) and, after one SiLU pass,
    the 6 modulation vectors (``norm1.linear``);
  * adaLN-Zero (normalization.py:84-86) and ``norm3`` + modulation (attention.py:463-466) = ``mi355x_sd_adaln`` (eps 1e-6 /
    ``norm_eps``); ``gate * (.) + x`` is the epilogue of ``attn1.to_out`` / ``ff.net.2``; tanh-GELU is the ``ff.net.0`` epilogue;
  * the final ``norm_out`` modulation reuses block 0's conditioning (:480-485), then ``proj_out_2`` and ``mi355x_sd_unpatchify``.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, List, Mapping, Optional

import numpy as np
import torch

from . import _lib
from ._lib import GELU_TANH, OUT_F32, SILU
from .checkpoint import PretrainedMixin, Table
from .program import DeviceProgram, _Plan, _Ref, _V
from .sd3 import Transformer2DModelOutput

Tensor = torch.Tensor

DIT_DEFAULTS = dict(num_attention_heads=16, attention_head_dim=72, in_channels=4, out_channels=8, num_layers=28,
                    sample_size=32, patch_size=2, num_embeds_ada_norm=1000, activation_fn="gelu-approximate",
                    attention_bias=True, norm_type="ada_norm_zero", norm_elementwise_affine=False, norm_eps=1e-5,
                    cross_attention_dim=None, dropout=0.0)


def normalize_config(config: Mapping) -> dict:
    cfg = dict(DIT_DEFAULTS)
    cfg.update({k: v for k, v in config.items() if not k.startswith("_")})
    if cfg["norm_type"] != "ada_norm_zero" or cfg["patch_size"] is None:
        raise NotImplementedError("Transformer2DModel(mi355x): only the patched ada_norm_zero (DiT) branch is implemented")
    if cfg["norm_elementwise_affine"] or cfg["cross_attention_dim"] is not None or not cfg["attention_bias"]:
        raise NotImplementedError("Transformer2DModel(mi355x): DiT configuration only (norm_elementwise_affine=False, "
                                  "no cross-attention, attention_bias=True)")
    if cfg["activation_fn"] != "gelu-approximate":
        raise NotImplementedError(f"activation_fn={cfg['activation_fn']!r}")
    if cfg["sample_size"] is None:
        raise ValueError("Transformer2DModel over patched input must provide sample_size")   # transformer_2d.py:183
    cfg["inner_dim"] = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    if cfg["out_channels"] is None:
        cfg["out_channels"] = cfg["in_channels"]
    return cfg


def dit_param_shapes(config: Mapping) -> Dict[str, tuple]:
    """name -> shape (Paddle layouts) in construction order (transformer_2d.py:182-254, attention.py:230-373)."""
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
        S[b + ".norm1.emb.class_embedder.embedding_table.weight"] = Table((cfg["num_embeds_ada_norm"] + 1, D))
        lin(b + ".norm1.linear", D, 6 * D)
        for nm in ("to_q", "to_k", "to_v", "to_out.0"):
            lin(b + ".attn1." + nm, D, D)
        lin(b + ".ff.net.0.proj", D, 4 * D)
        lin(b + ".ff.net.2", 4 * D, D)
    lin("proj_out_1", D, 2 * D)
    lin("proj_out_2", D, p * p * cfg["out_channels"])
    return S


def synth_dit_params(config: Mapping, seed: int = 1234, device="cpu", dtype=torch.float32) -> Dict[str, Tensor]:
    """Random-init parameters (same recipe as the oracle's synth_dit_params)."""
    g = torch.Generator(device=device).manual_seed(seed)
    P: Dict[str, Tensor] = {}
    for name, shape in dit_param_shapes(config).items():
        r = torch.randn(shape, generator=g, device=device)
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
        P[name] = t.to(dtype)
    return P


def sincos_pos_embed(embed_dim: int, grid: int, base_size: int, interpolation_scale: float = 1.0) -> np.ndarray:
    """get_2d_sincos_pos_embed (embeddings.py:67-120) for a square grid; [grid*grid, embed_dim]"""
    def one(dim, pos):
        omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)
    g = np.arange(grid, dtype=np.float32) / (grid / base_size) / interpolation_scale
    mesh = np.stack(np.meshgrid(g, g), axis=0).reshape([2, 1, grid, grid])   # w first, like the reference
    return np.concatenate([one(embed_dim // 2, mesh[0]), one(embed_dim // 2, mesh[1])], axis=1)


class DiTTransformer2DModel(DeviceProgram, PretrainedMixin):
    """``Transformer2DModel`` of the DiT pipeline (class name kept distinct from the UNet's inner Transformer2DModel)."""
    _param_shapes = staticmethod(dit_param_shapes)

    def __init__(self, config: Mapping, params: Mapping[str, Tensor], device="cuda", use_graph: bool = True,
                 profile: bool = False):
        self._init_backend(device, use_graph, profile)
        self.cfg = normalize_config(config)
        self.config = SimpleNamespace(**self.cfg)
        self._load_weights(params)

    # ------------------------------------------------------------------ weights
    def _load_weights(self, params: Mapping[str, Tensor]) -> None:
        cfg, dev, W = self.cfg, self.device, self.w
        shapes = dit_param_shapes(cfg)
        missing = [k for k in shapes if k not in params]
        if missing:
            raise KeyError(f"missing parameters: {missing[:5]}{'...' if len(missing) > 5 else ''}")

        def get(name):
            t = params[name]
            if tuple(t.shape) != shapes[name]:
                raise ValueError(f"{name}: shape {tuple(t.shape)} != expected {shapes[name]} (Paddle layout)")
            return t.to(device=dev, dtype=torch.float32)

        el = lambda t: t.to(_lib.elem_dtype()).contiguous()  # noqa: E731

        def put(key, name):
            W[key + ".w"] = el(get(name + ".weight").t())
            W[key + ".b"] = get(name + ".bias").contiguous()

        n = cfg["num_layers"]
        w = get("pos_embed.proj.weight")
        W["patch.w"] = el(w.reshape(w.shape[0], -1))   # [D, C*p*p], columns (c, py, px)
        W["patch.b"] = get("pos_embed.proj.bias").contiguous()
        t1w, t1b, tables = [], [], []
        for i in range(n):
            b = f"transformer_blocks.{i}"
            t1w.append(get(b + ".norm1.emb.timestep_embedder.linear_1.weight").t())
            t1b.append(get(b + ".norm1.emb.timestep_embedder.linear_1.bias"))
            tables.append(get(b + ".norm1.emb.class_embedder.embedding_table.weight"))
            put(b + ".t2", b + ".norm1.emb.timestep_embedder.linear_2")
            put(b + ".mod", b + ".norm1.linear")
            W[b + ".qkv.w"] = el(torch.cat([get(f"{b}.attn1.{x}.weight").t() for x in ("to_q", "to_k", "to_v")], 0))
            W[b + ".qkv.b"] = torch.cat([get(f"{b}.attn1.{x}.bias") for x in ("to_q", "to_k", "to_v")], 0).contiguous()
            put(b + ".out", b + ".attn1.to_out.0")
            put(b + ".ff1", b + ".ff.net.0.proj")
            put(b + ".ff2", b + ".ff.net.2")
        W["t1_all.w"] = el(torch.cat(t1w, 0))          # [n*D, 256]: every block's timestep_embedder.linear_1
        W["t1_all.b"] = torch.cat(t1b, 0).contiguous()
        W["class_tables"] = el(torch.cat(tables, 0))   # [n*(classes+1), D]: block i's rows start at i*(classes+1)
        put("proj_out_1", "proj_out_1")
        put("proj_out_2", "proj_out_2")

    # ------------------------------------------------------------------ plan
    def _build_plan(self, B: int, H: int, Wd: int) -> _Plan:
        cfg, lib, dev, W = self.cfg, self._lib, self.device, self.w
        stream = self._stream_ptr
        D, heads, p, n = cfg["inner_dim"], cfg["num_attention_heads"], cfg["patch_size"], cfg["num_layers"]
        if H % p or Wd % p or H != Wd:
            raise ValueError(f"DiT latents must be square and a multiple of patch_size {p} (the reference un-patchifies "
                             f"with int(sqrt(tokens)), transformer_2d.py:495-496), got {H}x{Wd}")
        hp = H // p
        S = hp * hp
        plan = _Plan()
        prog: List[tuple] = []
        scratch: Dict[str, int] = {}
        keep: List[Tensor] = []
        ET = _lib.elem_dtype()

        def sc(name, nbytes):
            scratch[name] = max(scratch.get(name, 0), nbytes)
            return _Ref(name)

        def persist(shape, dtype):
            t = torch.empty(shape, device=dev, dtype=dtype)
            keep.append(t)
            return t

        def emit(fn, args, kind, flops=0.0, desc=""):
            prog.append((fn, list(args), kind if not desc else f"{kind}:{desc}", flops))

        def linear(a: _V, wkey: str, out: _V, *, flags=0, R: Optional[_V] = None, gate=None, ld_gate=0, rpb=0):
            w = W[wkey + ".w"]
            N, K = w.shape
            assert K == a.C, (wkey, K, a.C)
            emit(lib.mi355x_sd_linear_ex,
                 (a.p, a.ld, 0, 0, w.data_ptr(), None, out.p, out.ld, 0, 0, a.rows, N, K, W[wkey + ".b"].data_ptr(), None, 0,
                  gate, ld_gate, rpb, R.p if R else None, R.ld if R else 0, 1.0, flags, *self._gemm_ws, stream), "gemm",
                 2.0 * a.rows * N * K, f"{a.rows}x{N}x{K}")

        def adaln(x: _V, scale_ptr, shift_ptr, ld_mod, eps, out: _V):
            emit(lib.mi355x_sd_adaln, (x.p, x.rows, x.C, x.ld, scale_ptr, shift_ptr, ld_mod, S, eps, out.p, out.ld, stream),
                 "ln")

        # ---- inputs ----
        plan.sample = persist((B, cfg["in_channels"], H, Wd), torch.float32)
        plan.t = persist((B,), torch.float32)
        plan.class_ids = persist((n * B,), torch.int32)   # label + i*(classes+1) for block i (staged by the host)
        plan.out = persist((B, cfg["out_channels"], H, Wd), torch.float32)

        # ---- PatchEmbed: conv p x p / p as a GEMM over patch rows, + the sincos position table (embeddings.py:209-247) ----
        base = cfg["sample_size"] // p
        pos = torch.from_numpy(sincos_pos_embed(D, hp, base, max(cfg["sample_size"] // 64, 1))).float()
        pos_t = persist((B * S, D), ET)
        pos_t.copy_(pos.reshape(1, S, D).expand(B, S, D).reshape(B * S, D))
        plan.consts = [pos_t]      # filled here, read by every run (paddlemix_amd/export.py ships its contents)
        kp = cfg["in_channels"] * p * p
        patches = persist((B * S, kp), ET)
        emit(lib.mi355x_sd_patchify, (plan.sample.data_ptr(), B, cfg["in_channels"], H, Wd, p, patches.data_ptr(), kp, stream),
             "misc")
        x_t = persist((B * S, D), ET)
        x = _V(x_t.data_ptr(), B * S, D)
        linear(_V(patches.data_ptr(), B * S, kp), "patch", x, R=_V(pos_t.data_ptr(), B * S, D))

        # ---- conditioning of every block (normalization.py:81-84, embeddings.py:557-565) ----
        tproj = persist((B, 256), ET)
        emit(lib.mi355x_sd_timestep_embedding, (plan.t.data_ptr(), B, B, 256, 1, 1, 1.0, 1.0, 10000.0, tproj.data_ptr(), 256,
                                                stream), "misc")
        h1 = persist((B, n * D), ET)        # SiLU(linear_1) of all blocks side by side
        linear(_V(tproj.data_ptr(), B, 256), "t1_all", _V(h1.data_ptr(), B, n * D), flags=SILU)
        cls = persist((n * B, D), ET)       # class rows: block i at rows i*B ..
        emit(lib.mi355x_sd_embed_tokens, (plan.class_ids.data_ptr(), n * B, 1, W["class_tables"].data_ptr(), None, D,
                                          cls.data_ptr(), D, stream), "misc")
        cond = persist((n * B, D), ET)      # conditioning = timestep embedding + class embedding, per block
        for i in range(n):
            b = f"transformer_blocks.{i}"
            linear(_V(h1.data_ptr() + 2 * i * D, B, D, n * D), b + ".t2", _V(cond.data_ptr() + 2 * i * B * D, B, D),
                   R=_V(cls.data_ptr() + 2 * i * B * D, B, D))
        scond = persist((n * B, D), ET)
        emit(lib.mi355x_sd_silu, (cond.data_ptr(), scond.data_ptr(), n * B * D, 0, 0, stream), "misc")
        mod = persist((n + 1, B, 6 * D), torch.float32)   # [block][batch][shift_msa scale_msa gate_msa shift_mlp scale_mlp gate_mlp]
        LDM = 6 * D
        for i in range(n):
            linear(_V(scond.data_ptr() + 2 * i * B * D, B, D), f"transformer_blocks.{i}.mod",
                   _V(mod.data_ptr() + 4 * i * B * LDM, B, LDM), flags=OUT_F32)
        # final modulation from block 0's conditioning (transformer_2d.py:480-483): [shift | scale]
        linear(_V(scond.data_ptr(), B, D), "proj_out_1", _V(mod.data_ptr() + 4 * n * B * LDM, B, 2 * D, LDM), flags=OUT_F32)
        m_at = lambda i, chunk: mod.data_ptr() + 4 * (i * B * LDM + chunk * D)  # noqa: E731

        # ---- blocks (attention.py:376-490, ada_norm_zero path) ----
        nx = _V(sc("nx", 2 * B * S * D), B * S, D)
        qkv = sc("qkv", 2 * B * S * 3 * D)
        ao = _V(sc("attn_out", 2 * B * S * D), B * S, D)
        ff = _V(sc("ff", 2 * B * S * 4 * D), B * S, 4 * D)
        d = D // heads
        for i in range(n):
            b = f"transformer_blocks.{i}"
            adaln(x, m_at(i, 1), m_at(i, 0), LDM, 1e-6, nx)                      # norm1: LN * (1 + scale_msa) + shift_msa
            linear(nx, b + ".qkv", _V(qkv, B * S, 3 * D))
            emit(lib.mi355x_sd_sdpa, (qkv, qkv + 2 * D, qkv + 4 * D, None, ao.p, B, heads, S, S, d, S * 3 * D, 3 * D,
                                      S * 3 * D, 3 * D, S * 3 * D, 3 * D, S * D, D, 0, 0, 0, d ** -0.5, stream),
                 "attn", 4.0 * B * heads * S * S * d, f"{B}x{heads}x{S}x{S}x{d}")
            linear(ao, b + ".out", x, R=x, gate=m_at(i, 2), ld_gate=LDM, rpb=S)   # x += gate_msa * attn
            adaln(x, m_at(i, 4), m_at(i, 3), LDM, float(cfg["norm_eps"]), nx)     # norm3 + (1 + scale_mlp), shift_mlp
            linear(nx, b + ".ff1", ff, flags=GELU_TANH)
            linear(ff, b + ".ff2", x, R=x, gate=m_at(i, 5), ld_gate=LDM, rpb=S)   # x += gate_mlp * ff

        # ---- output (transformer_2d.py:478-503) ----
        adaln(x, m_at(n, 1), m_at(n, 0), LDM, 1e-6, nx)
        po = p * p * cfg["out_channels"]
        proj = persist((B * S, po), ET)
        linear(nx, "proj_out_2", _V(proj.data_ptr(), B * S, po))
        emit(lib.mi355x_sd_unpatchify, (proj.data_ptr(), po, B, cfg["out_channels"], H, Wd, p, plan.out.data_ptr(), stream),
             "misc")

        bufs = {nm: persist((max(nb, 16),), torch.uint8) for nm, nb in scratch.items()}
        basep = {nm: t.data_ptr() for nm, t in bufs.items()}
        res = lambda a: basep[a.buf] + a.off if isinstance(a, _Ref) else a  # noqa: E731
        plan.prog = [(fn, tuple(res(a) for a in args), kind, fl) for fn, args, kind, fl in prog]
        plan.keep, plan.graph = keep, None
        plan.B, plan.H, plan.W = B, H, Wd
        return plan

    def _get_plan(self, B, H, W) -> _Plan:
        key = (B, H, W)
        if key not in self._plans:
            self._plans[key] = self._build_plan(B, H, W)
        return self._plans[key]

    def stage_inputs(self, plan: _Plan, hidden_states, timestep, class_labels) -> None:
        n, B = self.cfg["num_layers"], plan.B
        t = timestep if torch.is_tensor(timestep) else torch.as_tensor(float(timestep))
        t = t.reshape(-1).to(torch.float32)
        plan.t.copy_(t.expand(B) if t.numel() == 1 else t, non_blocking=True)
        plan.sample.copy_(hidden_states, non_blocking=True)
        labels = class_labels.reshape(-1).to(torch.int64)
        if labels.numel() != B:
            raise ValueError(f"class_labels of shape {tuple(class_labels.shape)}, expected ({B},)")
        off = torch.arange(n, device=labels.device, dtype=torch.int64) * (self.cfg["num_embeds_ada_norm"] + 1)
        plan.class_ids.copy_((labels[None, :] + off[:, None]).reshape(-1).to(torch.int32), non_blocking=True)

    def forward(self, hidden_states, encoder_hidden_states=None, timestep=None, added_cond_kwargs=None, class_labels=None,
                cross_attention_kwargs=None, attention_mask=None, encoder_attention_mask=None, return_dict: bool = True):
        for nm, v in (("encoder_hidden_states", encoder_hidden_states), ("attention_mask", attention_mask),
                      ("encoder_attention_mask", encoder_attention_mask)):
            if v is not None:
                raise NotImplementedError(f"Transformer2DModel(mi355x, DiT): `{nm}` is not used on this path")
        if timestep is None or class_labels is None:
            raise ValueError("the ada_norm_zero (DiT) branch needs `timestep` and `class_labels`")
        if not self._emulated and not hidden_states.is_cuda:
            raise _lib.MI355XError("inputs must be GPU tensors (no CPU fallback)")
        B, _, H, W = hidden_states.shape
        plan = self._get_plan(B, H, W)
        if self._emulated:
            self.stage_inputs(plan, hidden_states, timestep, class_labels)
            self._run_eager(plan)
            out = plan.out.clone()
        else:
            cur = torch.cuda.current_stream(self.device)
            self._stream.wait_stream(cur)
            with torch.cuda.stream(self._stream):
                self.stage_inputs(plan, hidden_states, timestep, class_labels)
                out = self.run(plan).clone()
            cur.wait_stream(self._stream)
        if not return_dict:
            return (out,)
        return Transformer2DModelOutput(sample=out)

    __call__ = forward
