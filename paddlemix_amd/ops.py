"""Tensor-level wrappers over the C ABI (one call = one HIP kernel launch on torch's current stream).

torch is used only for device memory and streams; every function here raises if the tensors are not on a
GPU or the library is missing.  Activations are bf16 rows ``[rows, C]`` (== NHWC) whose row stride may exceed C.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import _lib
from ._lib import GEGLU, GELU_TANH, OUT_F32, SILU, check

Tensor = torch.Tensor


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


_workspaces = {}


def _workspace(dev: torch.device, nbytes: int = 32 << 20):
    """(pointer, bytes) of the split-K / widening scratch handed to a GEMM-class call of the stand-alone op wrappers: one buffer per
    (device, stream) -- the scratch belongs to the call (ABI 12), launches ordered on one stream may share one, launches on two
    streams must not."""
    if dev.type != "cuda":
        # (round 5: a wrapper called with CPU tensors used to bind a 32-MB HOST buffer here before its own argument check raised --
        # and the next split-K launch that did not rebind wrote its partial sums through that pointer: a page fault on boxes
        # without XNACK, the one-off abort of profiles/r05_s14)
        raise _lib.MI355XError("tensors must live on the GPU (no CPU fallback)")
    key = (dev, _stream())
    ws = _workspaces.get(key)
    if ws is None:
        ws = _workspaces[key] = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    return ws.data_ptr(), ws.numel()


def _p(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _rows(t: Tensor, name: str, dtype=None) -> int:
    """Validate a 2-D row tensor (unit inner stride; the build's element type unless `dtype` says otherwise) and return
    its row stride."""
    dtype = dtype or _lib.elem_dtype()
    if not t.is_cuda:
        raise _lib.MI355XError(f"{name}: tensor must live on the GPU (no CPU fallback)")
    if t.dtype != dtype or t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError(f"{name}: expected 2-D {dtype} rows with unit inner stride, got {t.dtype} {tuple(t.shape)} "
                         f"strides {t.stride()}")
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def _vec(t: Optional[Tensor], n: int, name: str) -> Optional[Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n or not t.is_cuda:
        raise ValueError(f"{name}: expected contiguous fp32 cuda vector of {n} elements")
    return t


def init(device: int = 0) -> None:
    check(_lib.load().mi355x_sd_init(device))


def linear(a: Tensor, w: Tensor, bias: Optional[Tensor] = None, *, rowbias: Optional[Tensor] = None,
           rows_per_batch: int = 0, residual: Optional[Tensor] = None, out: Optional[Tensor] = None,
           out_scale: float = 1.0, geglu: bool = False, silu: bool = False, out_f32: bool = False) -> Tensor:
    """out[M,N] = ((a[M,K] @ w[N,K]^T) + bias + rowbias[m // rows_per_batch] + residual) * out_scale."""
    lib = _lib.load()
    ws = _workspace(a.device)
    lda = _rows(a, "a")
    M, K = a.shape
    if w.dtype != _lib.elem_dtype() or not w.is_contiguous() or w.dim() != 2 or w.shape[1] != K:
        raise ValueError(f"w: expected contiguous bf16 [N,{K}], got {w.dtype} {tuple(w.shape)}")
    N = w.shape[0]
    n_out = N // 2 if geglu else N
    if out is None:
        out = torch.empty((M, n_out), device=a.device, dtype=torch.float32 if out_f32 else _lib.elem_dtype())
    ldc = _rows(out, "out", torch.float32 if out_f32 else _lib.elem_dtype())
    if out.shape != (M, n_out):
        raise ValueError(f"out: expected {(M, n_out)}, got {tuple(out.shape)}")
    r_f32 = residual is not None and residual.dtype == torch.float32   # rows of the fp32 residual stream (MI355X_SD_R_F32)
    ldr = _rows(residual, "residual", torch.float32 if r_f32 else None) if residual is not None else 0
    ld_rb = 0
    if rowbias is not None:
        if rowbias.dtype != torch.float32 or rowbias.dim() != 2 or rowbias.stride(1) != 1 or rowbias.shape[1] != N:
            raise ValueError("rowbias: expected fp32 [batches, N] rows")
        ld_rb = rowbias.stride(0)
    flags = (GEGLU if geglu else 0) | (OUT_F32 if out_f32 else 0) | (SILU if silu else 0) | (_lib.R_F32 if r_f32 else 0)
    check(lib.mi355x_sd_linear(a.data_ptr(), lda, w.data_ptr(), out.data_ptr(), ldc, M, N, K,
                               _p(_vec(bias, N, "bias")), _p(rowbias), rows_per_batch, ld_rb, _p(residual), ldr,
                               float(out_scale), flags, *ws, _stream()))
    return out


def conv3x3(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, *, stride: int = 1, upsample: bool = False,
            rowbias: Optional[Tensor] = None, residual: Optional[Tensor] = None, out: Optional[Tensor] = None,
            out_scale: float = 1.0, pad_br: bool = False, kb64: bool = False) -> Tensor:
    """x NHWC view [B,H,W,C] (pixel stride >= C), w [Cout, 9*Cin] -> rows [B*Ho*Wo, Cout]. ``pad_br`` (stride 2): zero
    padding at the bottom / right only -- Downsample2D(padding=0), resnet.py:277-279. ``kb64``: w is packed
    [Cout][Cin/64][3][3][64] (MI355X_SD_CONV_KB64) instead of [Cout][3][3][Cin]."""
    lib = _lib.load()
    ws = _workspace(x.device)
    if x.dim() != 4 or x.dtype != _lib.elem_dtype() or x.stride(3) != 1 or not x.is_cuda:
        raise ValueError("x: expected bf16 cuda NHWC [B,H,W,C]")
    B, H, W, C = x.shape
    ldx = x.stride(2)
    if x.stride(1) != W * ldx or x.stride(0) != H * W * ldx:
        raise ValueError("x: pixels must be densely packed with a common stride")
    Cout = w.shape[0]
    if w.dtype != _lib.elem_dtype() or not w.is_contiguous() or w.shape[1] != 9 * C:
        raise ValueError(f"w: expected contiguous bf16 [Cout, {9 * C}]")
    up = 1 if upsample else 0
    pad2 = 1 if pad_br else 2
    Ho = ((H << up) + pad2 - 3) // stride + 1
    Wo = ((W << up) + pad2 - 3) // stride + 1
    M = B * Ho * Wo
    if out is None:
        out = torch.empty((M, Cout), device=x.device, dtype=_lib.elem_dtype())
    ldc = _rows(out, "out")
    ldr = _rows(residual, "residual") if residual is not None else 0
    ld_rb = rowbias.stride(0) if rowbias is not None else 0
    check(lib.mi355x_sd_conv3x3(x.data_ptr(), ldx, B, H, W, C, stride, up, w.data_ptr(), out.data_ptr(), ldc, Cout,
                                _p(_vec(bias, Cout, "bias")), _p(rowbias), ld_rb, _p(residual), ldr, float(out_scale),
                                (_lib.PAD_BR if pad_br else 0) | (_lib.CONV_KB64 if kb64 else 0), *ws, _stream()))
    return out


def latent_dist(moments: Tensor, B: int, L: int, noise: Optional[Tensor] = None, out_scale: float = 1.0,
                want_sample: bool = True):
    """moments fp32 rows [B*HW, >= 2L] -> (mean, logvar, sample | None), NCHW-flattened fp32 [B, L, HW]
    (DiagonalGaussianDistribution, PPD/models/vae.py:744-763)."""
    lib = _lib.load()
    if not moments.is_cuda or moments.dtype != torch.float32 or moments.dim() != 2 or moments.stride(1) != 1:
        raise ValueError("moments: expected fp32 GPU rows")
    if moments.shape[0] % B or moments.shape[1] < 2 * L:
        raise ValueError("moments: expected [B*HW, >= 2L]")
    HW = moments.shape[0] // B
    mk = lambda: torch.empty((B, L, HW), device=moments.device, dtype=torch.float32)  # noqa: E731
    mean, logvar = mk(), mk()
    sample = mk() if want_sample else None
    if noise is not None and (noise.dtype != torch.float32 or noise.numel() != B * L * HW or not noise.is_contiguous()):
        raise ValueError("noise: expected contiguous fp32 [B, L, HW]")
    check(lib.mi355x_sd_latent_dist(moments.data_ptr(), moments.stride(0), B, L, HW, _p(noise), float(out_scale),
                                    mean.data_ptr(), logvar.data_ptr(), _p(sample), _stream()))
    return mean, logvar, sample


def sdpa(q: Tensor, k: Tensor, v: Tensor, bias: Optional[Tensor] = None, scale: Optional[float] = None,
         out: Optional[Tensor] = None, accum: Optional[float] = None, log2: bool = False) -> Tensor:
    """q [B,Sq,H,D], k/v [B,Skv,H,D] (head-contiguous token rows, arbitrary token/batch strides) -> [B,Sq,H,D].
    bias: optional fp32 additive mask broadcastable to [B,H,Sq,Skv] with unit inner stride. log2: q already carries
    scale * log2(e) (MI355X_SD_SDPA_LOG2): out = softmax_2(q k^T) v."""
    lib = _lib.load()
    for name, t in (("q", q), ("k", k), ("v", v)):
        if t.dim() != 4 or t.dtype != _lib.elem_dtype() or not t.is_cuda or t.stride(3) != 1 or t.stride(2) != t.shape[3]:
            raise ValueError(f"{name}: expected bf16 cuda [B,S,H,D] with heads packed inside a token row")
    B, Sq, H, D = q.shape
    Skv = k.shape[1]
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    if out is None:
        out = torch.empty((B, Sq, H, D), device=q.device, dtype=_lib.elem_dtype())
    bb = bh = bq = 0
    if bias is not None:
        if bias.dtype != torch.float32 or bias.dim() != 4 or bias.shape[-1] != Skv or bias.stride(3) != 1:
            raise ValueError("bias: expected fp32 [B|1,H|1,Sq|1,Skv]")
        bb = bias.stride(0) if bias.shape[0] > 1 else 0
        bh = bias.stride(1) if bias.shape[1] > 1 else 0
        bq = bias.stride(2) if bias.shape[2] > 1 else 0
    args = (q.data_ptr(), k.data_ptr(), v.data_ptr(), _p(bias), out.data_ptr(), B, H, Sq, Skv, D,
            q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
            out.stride(0), out.stride(1), bb, bh, bq, float(scale))
    if log2:
        if accum is not None:
            raise ValueError("log2 and accum cannot be combined")
        check(lib.mi355x_sd_sdpa_ex(*args, _lib.SDPA_LOG2, _stream()))
    elif accum is None:
        check(lib.mi355x_sd_sdpa(*args, _stream()))
    else:   # out += accum * attention (`out` given and already holding a first attention's result)
        check(lib.mi355x_sd_sdpa_accum(*args, float(accum), _stream()))
    return out


def groupnorm_act(x: Tensor, gamma: Tensor, beta: Tensor, groups: int, eps: float, silu: bool = False) -> Tensor:
    """GroupNorm (+SiLU) of x [B,HW,C] in one launch (mi355x_sd_groupnorm_act); raises MI355XError where a (batch, group) chunk does
    not fit a block's registers (`groupnorm_act_fits`): use groupnorm_scale_shift + scale_shift_act there."""
    lib = _lib.load()
    B, HW, C = x.shape
    if x.dtype != _lib.elem_dtype() or x.stride(2) != 1 or x.stride(0) != HW * x.stride(1) or not x.is_cuda:
        raise ValueError("x: expected bf16 cuda [B,HW,C] rows")
    y = torch.empty((B, HW, C), device=x.device, dtype=_lib.elem_dtype())
    check(lib.mi355x_sd_groupnorm_act(x.data_ptr(), B, HW, C, x.stride(1), groups, float(eps), _vec(gamma, C, "gamma").data_ptr(),
                                      _vec(beta, C, "beta").data_ptr(), 1 if silu else 0, y.data_ptr(), C, _stream()))
    return y


def groupnorm_act_fits(HW: int, C: int, groups: int) -> bool:
    return bool(_lib.load().mi355x_sd_groupnorm_act_fits(HW, C, groups))


def groupnorm_scale_shift(x: Tensor, gamma: Tensor, beta: Tensor, groups: int, eps: float) -> Tensor:
    """x [B,HW,C] (row stride >= C) -> scale_shift fp32 [B,2,C]."""
    lib = _lib.load()
    B, HW, C = x.shape
    if x.dtype != _lib.elem_dtype() or x.stride(2) != 1 or x.stride(0) != HW * x.stride(1) or not x.is_cuda:
        raise ValueError("x: expected bf16 cuda [B,HW,C] rows")
    ws = torch.empty(max(1, lib.mi355x_sd_groupnorm_workspace_floats(B, HW, C)), device=x.device, dtype=torch.float32)
    ss = torch.empty((B, 2, C), device=x.device, dtype=torch.float32)
    check(lib.mi355x_sd_groupnorm_stats(x.data_ptr(), B, HW, C, x.stride(1), groups, float(eps),
                                        _vec(gamma, C, "gamma").data_ptr(), _vec(beta, C, "beta").data_ptr(),
                                        ws.data_ptr(), ss.data_ptr(), _stream()))
    return ss


def scale_shift_act(x: Tensor, scale_shift: Tensor, silu: bool, out: Optional[Tensor] = None) -> Tensor:
    lib = _lib.load()
    B, HW, C = x.shape
    if out is None:
        out = torch.empty((B, HW, C), device=x.device, dtype=_lib.elem_dtype())
    check(lib.mi355x_sd_scale_shift_act(x.data_ptr(), B, HW, C, x.stride(1), scale_shift.data_ptr(), 1 if silu else 0,
                                        out.data_ptr(), out.stride(1), _stream()))
    return out


def group_norm(x: Tensor, gamma: Tensor, beta: Tensor, groups: int, eps: float, silu: bool = False,
               out: Optional[Tensor] = None) -> Tensor:
    return scale_shift_act(x, groupnorm_scale_shift(x, gamma, beta, groups, eps), silu, out)


def layer_norm(x: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], eps: float = 1e-5,
               out: Optional[Tensor] = None) -> Tensor:
    lib = _lib.load()
    ldx = _rows(x, "x")
    rows, C = x.shape
    if out is None:
        out = torch.empty((rows, C), device=x.device, dtype=_lib.elem_dtype())
    check(lib.mi355x_sd_layernorm(x.data_ptr(), rows, C, ldx, _p(_vec(gamma, C, "gamma")), _p(_vec(beta, C, "beta")),
                                  float(eps), out.data_ptr(), _rows(out, "out"), _stream()))
    return out


def timestep_embedding(t: Tensor, dim: int, flip_sin_to_cos: bool = False, downscale_freq_shift: float = 1.0,
                       scale: float = 1.0, max_period: float = 10000.0, n: Optional[int] = None) -> Tensor:
    """t: fp32 cuda [t_count]; returns bf16 [n, dim] (t broadcast cyclically when n > t_count)."""
    lib = _lib.load()
    if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
        raise ValueError("t: expected contiguous fp32 cuda tensor")
    n = t.numel() if n is None else n
    out = torch.empty((n, dim), device=t.device, dtype=_lib.elem_dtype())
    check(lib.mi355x_sd_timestep_embedding(t.data_ptr(), t.numel(), n, dim, 1, 1 if flip_sin_to_cos else 0,
                                           float(downscale_freq_shift), float(scale), float(max_period),
                                           out.data_ptr(), dim, _stream()))
    return out


def silu(x: Tensor, out_dtype=None) -> Tensor:
    lib = _lib.load()
    out_dtype = out_dtype or x.dtype
    if not x.is_contiguous() or not x.is_cuda:
        raise ValueError("x: expected contiguous cuda tensor")
    out = torch.empty(x.shape, device=x.device, dtype=out_dtype)
    check(lib.mi355x_sd_silu(x.data_ptr(), out.data_ptr(), x.numel(), int(x.dtype == torch.float32),
                             int(out_dtype == torch.float32), _stream()))
    return out


def conv_in3x3(x_nchw: Tensor, w: Tensor, bias: Optional[Tensor], in_scale: Optional[Tensor] = None,
               out: Optional[Tensor] = None) -> Tensor:
    """x fp32 NCHW, w bf16 [9*Cin, Cout] -> bf16 rows [B*H*W, Cout]."""
    lib = _lib.load()
    B, Cin, H, W = x_nchw.shape
    Cout = w.shape[1]
    if x_nchw.dtype != torch.float32 or not x_nchw.is_contiguous() or not x_nchw.is_cuda:
        raise ValueError("x: expected contiguous fp32 cuda NCHW")
    if out is None:
        out = torch.empty((B * H * W, Cout), device=x_nchw.device, dtype=_lib.elem_dtype())
    check(lib.mi355x_sd_conv_in3x3(x_nchw.data_ptr(), _p(in_scale), w.data_ptr(), _p(_vec(bias, Cout, "bias")),
                                   out.data_ptr(), B, Cin, H, W, Cout, _rows(out, "out"), _stream()))
    return out


def conv_out3x3(x: Tensor, w: Tensor, bias: Optional[Tensor], B: int, H: int, W: int) -> Tensor:
    """x bf16 rows [B*H*W, Cin], w bf16 [Cout, 9*Cin] -> fp32 NCHW [B,Cout,H,W]."""
    lib = _lib.load()
    ldx = _rows(x, "x")
    Cin = x.shape[1]
    Cout = w.shape[0]
    out = torch.empty((B, Cout, H, W), device=x.device, dtype=torch.float32)
    check(lib.mi355x_sd_conv_out3x3(x.data_ptr(), ldx, w.data_ptr(), _p(_vec(bias, Cout, "bias")), out.data_ptr(), B,
                                    Cin, H, W, Cout, _stream()))
    return out


def axpby(x: Tensor, y: Tensor, coef: Tensor, out: Optional[Tensor] = None) -> Tensor:
    lib = _lib.load()
    if out is None:
        out = torch.empty_like(x)
    check(lib.mi355x_sd_axpby(x.data_ptr(), y.data_ptr(), out.data_ptr(), coef.data_ptr(), x.numel(), _stream()))
    return out


def probe_layouts(device="cuda") -> Tensor:
    lib = _lib.load()
    out = torch.zeros((64, 24), device=device, dtype=torch.float32)
    check(lib.mi355x_sd_probe_layouts(out.data_ptr(), _stream()))
    return out


def linear_ex(a: Tensor, w: Tensor, bias: Optional[Tensor] = None, *, w_scale: Optional[Tensor] = None,
              gate: Optional[Tensor] = None,
              rows_per_batch: int = 0, residual: Optional[Tensor] = None, out: Optional[Tensor] = None,
              gelu_tanh: bool = False, silu: bool = False, out_f32: bool = False, a_rows_per_batch: int = 0,
              a_batch_stride: int = 0, c_rows_per_batch: int = 0, c_batch_stride: int = 0, M: Optional[int] = None) -> Tensor:
    """mi355x_sd_linear_ex: out = residual + gate[m // rows_per_batch] * (a @ w^T + bias), optional row remaps of a / out
    (then `a` / `out` are the flat base tensors and M is given explicitly)."""
    lib = _lib.load()
    ws = _workspace(a.device)
    N, K = w.shape
    if a_rows_per_batch:
        lda = K if a.dim() == 1 else a.stride(0)
        assert M is not None
    else:
        lda = _rows(a, "a")
        M = a.shape[0]
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32 if out_f32 else _lib.elem_dtype())
    ldc = N if c_rows_per_batch else _rows(out, "out", torch.float32 if out_f32 else _lib.elem_dtype())
    ld_gate = gate.stride(0) if gate is not None else 0
    flags = (OUT_F32 if out_f32 else 0) | (SILU if silu else 0) | (GELU_TANH if gelu_tanh else 0)
    check(lib.mi355x_sd_linear_ex(a.data_ptr(), lda, a_rows_per_batch, a_batch_stride, w.data_ptr(), _p(w_scale), out.data_ptr(), ldc,
                                  c_rows_per_batch, c_batch_stride, M, N, K, _p(_vec(bias, N, "bias")), None, 0, _p(gate),
                                  ld_gate, rows_per_batch, _p(residual), _rows(residual, "residual") if residual is not None else 0,
                                  1.0, flags, *ws, _stream()))
    return out


def _mod_vectors(device, *ts):
    """Modulation / affine vectors as the C ABI takes them -> (tensors, ld_mod of the first 2-D group, mod_dtype). All in the build's
    16-bit element type (the reference's case: chunks of a 16-bit linear output, possibly strided views of it) -> MOD_ELEM, passed
    as they are when they share a row stride and sit on 16-byte boundaries; anything else -> contiguous fp32 copies, MOD_F32."""
    ed = _lib.elem_dtype()
    given = [t for t in ts if t is not None]
    two_d = [t for t in given if t.dim() == 2]
    if all(t.dtype == ed and t.is_cuda and t.stride(-1) == 1 and t.data_ptr() % 16 == 0 for t in given) and \
            len({t.stride(0) for t in two_d}) <= 1 and all(t.stride(0) % 8 == 0 for t in two_d):
        return list(ts), (two_d[0].stride(0) if two_d else 0), _lib.MOD_ELEM
    conv = [None if t is None else t.to(device=device, dtype=torch.float32).contiguous() for t in ts]
    two_d = [t for t in conv if t is not None and t.dim() == 2]
    return conv, (two_d[0].stride(0) if two_d else 0), _lib.MOD_F32


def adaln(x: Tensor, scale: Tensor, shift: Tensor, rows_per_batch: int, eps: float = 1e-6,
          out: Optional[Tensor] = None) -> Tensor:
    """y = LN_noaffine(x) * (1 + scale[b]) + shift[b]; scale/shift [batches, C] rows sharing one row stride, fp32 or the build's
    16-bit element type (mi355x_sd_adaln_ex)."""
    lib = _lib.load()
    ldx = _rows(x, "x")
    rows, C = x.shape
    if out is None:
        out = torch.empty((rows, C), device=x.device, dtype=_lib.elem_dtype())
    if scale.dtype == torch.float32 and shift.dtype == torch.float32:
        assert scale.stride(0) == shift.stride(0)
        check(lib.mi355x_sd_adaln(x.data_ptr(), rows, C, ldx, scale.data_ptr(), shift.data_ptr(), scale.stride(0),
                                  rows_per_batch, float(eps), out.data_ptr(), _rows(out, "out"), _stream()))
        return out
    (sc, sh), ld_mod, md = _mod_vectors(x.device, scale, shift)
    check(lib.mi355x_sd_adaln_ex(x.data_ptr(), rows, C, ldx, sc.data_ptr(), sh.data_ptr(), ld_mod, md, rows_per_batch, float(eps),
                                 out.data_ptr(), _rows(out, "out"), _stream()))
    return out


def adaptive_layer_norm(x: Tensor, scale: Tensor, shift: Tensor, weight: Optional[Tensor] = None, bias: Optional[Tensor] = None,
                        epsilon: float = 1e-05) -> Tensor:
    """The reference's fused op with its own signature (paddlemix/triton_ops/triton_ops.py:1030-1139; unfused definition
    :1084-1088): ``layer_norm(x, weight, bias, epsilon) * (1 + scale[:, None]) + shift[:, None]``; x [B, S, C] in the build's 16-bit
    type, scale / shift [B, C] in x.dtype as the reference passes them (fp32 also accepted)."""
    assert x.dim() == 3, "x should be 3-dim [batch_size, seq_size, feature_dim]"
    assert scale.shape == shift.shape and scale.dim() == 2, "scale and shift should be 2-dim [batch_size, feature_dim]"
    B, S, C = x.shape
    assert scale.shape[0] == B and scale.shape[1] == C, "x, scale and shift should have same batch_size / feature_dim"
    if x.dtype != _lib.elem_dtype() or not x.is_cuda:
        raise ValueError(f"x: expected a {_lib.elem_dtype()} cuda tensor")
    xx = x.contiguous()
    if weight is None and bias is None:
        return adaln(xx.view(B * S, C), scale, shift, S, epsilon).view(B, S, C)
    # with an affine LayerNorm: the two-output op with a zero gate (resi_out = x is discarded)
    zero = torch.zeros_like(scale)
    return fused_adaLN_scale_residual(xx, xx, zero, scale, shift, weight, bias, epsilon)[1]


def patchify(x_nchw: Tensor, patch: int) -> Tensor:
    lib = _lib.load()
    B, C, H, W = x_nchw.shape
    out = torch.empty((B * (H // patch) * (W // patch), C * patch * patch), device=x_nchw.device, dtype=_lib.elem_dtype())
    check(lib.mi355x_sd_patchify(x_nchw.data_ptr(), B, C, H, W, patch, out.data_ptr(), out.shape[1], _stream()))
    return out


def unpatchify(x: Tensor, B: int, C: int, H: int, W: int, patch: int) -> Tensor:
    lib = _lib.load()
    out = torch.empty((B, C, H, W), device=x.device, dtype=torch.float32)
    check(lib.mi355x_sd_unpatchify(x.data_ptr(), _rows(x, "x"), B, C, H, W, patch, out.data_ptr(), _stream()))
    return out


def conv1x1_nchw(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, in_scale: float = 1.0) -> Tensor:
    """post_quant_conv: x fp32 [B, Cin, H, W], w bf16 [Cout, Cin] -> fp32 [B, Cout, H, W]."""
    lib = _lib.load()
    if not x.is_cuda or x.dtype != torch.float32 or not x.is_contiguous() or x.dim() != 4:
        raise ValueError("x: expected a contiguous fp32 NCHW GPU tensor")
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    if w.dtype != _lib.elem_dtype() or tuple(w.shape) != (Cout, Cin) or not w.is_contiguous():
        raise ValueError(f"w: expected contiguous bf16 [Cout, {Cin}]")
    y = torch.empty((B, Cout, H, W), device=x.device, dtype=torch.float32)
    check(lib.mi355x_sd_conv1x1_nchw(x.data_ptr(), float(in_scale), w.data_ptr(), _p(_vec(bias, Cout, "bias")),
                                     y.data_ptr(), B, Cin, Cout, H * W, _stream()))
    return y


def softmax_rows(x: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """bf16 softmax over the last dim of an fp32 [rows, n] matrix."""
    lib = _lib.load()
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("x: expected an fp32 [rows, n] GPU tensor with unit inner stride")
    rows, n = x.shape
    if out is None:
        out = torch.empty((rows, n), device=x.device, dtype=_lib.elem_dtype())
    check(lib.mi355x_sd_softmax_rows(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), rows, n, _stream()))
    return out


def row_stats(x: Tensor, eps: float = 1e-5) -> Tensor:
    """fp32 [rows, 2] = (rstd, -mean * rstd) per row of a bf16 [rows, C] view."""
    lib = _lib.load()
    ldx = _rows(x, "x")
    rows, C = x.shape
    st = torch.empty((rows, 2), device=x.device, dtype=torch.float32)
    check(lib.mi355x_sd_row_stats(x.data_ptr(), rows, C, ldx, float(eps), st.data_ptr(), _stream()))
    return st


def linear_ln(a: Tensor, stats: Tensor, w: Tensor, w_rowsum: Tensor, bias: Optional[Tensor] = None, *,
              geglu: bool = False, out: Optional[Tensor] = None) -> Tensor:
    """LayerNorm-folded projection: out = rstd * (a @ w^T) - mean * rstd * w_rowsum + bias (optionally GEGLU)."""
    lib = _lib.load()
    ws = _workspace(a.device)
    lda = _rows(a, "a")
    M, K = a.shape
    N = w.shape[0]
    if w.dtype != _lib.elem_dtype() or not w.is_contiguous() or tuple(w.shape) != (N, K):
        raise ValueError(f"w: expected contiguous bf16 [N,{K}]")
    if stats.dtype != torch.float32 or tuple(stats.shape) != (M, 2) or not stats.is_contiguous():
        raise ValueError("stats: expected contiguous fp32 [M, 2]")
    n_out = N // 2 if geglu else N
    if out is None:
        out = torch.empty((M, n_out), device=a.device, dtype=_lib.elem_dtype())
    ldc = _rows(out, "out")
    check(lib.mi355x_sd_linear_ln(a.data_ptr(), lda, stats.data_ptr(), w.data_ptr(), _vec(w_rowsum, N, "w_rowsum").data_ptr(),
                                  out.data_ptr(), ldc, M, N, K, _p(_vec(bias, N, "bias")), GEGLU if geglu else 0, *ws, _stream()))
    return out


def embed_tokens(ids: Tensor, token_table: Tensor, position_table: Tensor, seq_len: int) -> Tensor:
    """CLIPTextEmbeddings: bf16 rows token_table[ids] + position_table[i % seq_len]; ids int32 [n]."""
    lib = _lib.load()
    if ids.dtype != torch.int32 or not ids.is_cuda or not ids.is_contiguous():
        raise ValueError("ids: expected a contiguous int32 GPU tensor")
    V, D = token_table.shape
    if int(ids.min()) < 0 or int(ids.max()) >= V or seq_len > position_table.shape[0]:
        raise ValueError("ids / seq_len out of range of the embedding tables")
    out = torch.empty((ids.numel(), D), device=ids.device, dtype=_lib.elem_dtype())
    check(lib.mi355x_sd_embed_tokens(ids.data_ptr(), ids.numel(), seq_len, token_table.data_ptr(),
                                     position_table.data_ptr(), D, out.data_ptr(), D, _stream()))
    return out


def activation(x: Tensor, kind: str) -> Tensor:
    """quick_gelu | gelu | silu on a contiguous bf16 tensor."""
    lib = _lib.load()
    kinds = {"quick_gelu": 0, "gelu": 1, "silu": 2}
    if x.dtype != _lib.elem_dtype() or not x.is_cuda or not x.is_contiguous():
        raise ValueError("x: expected a contiguous bf16 GPU tensor")
    y = torch.empty_like(x)
    check(lib.mi355x_sd_activation(x.data_ptr(), y.data_ptr(), x.numel(), kinds[kind], _stream()))
    return y


def rms_norm(x: Tensor, weight: Tensor, eps: float = 1e-6) -> Tensor:
    """T5LayerNorm: weight * x * rsqrt(mean(x^2) + eps) on bf16 rows."""
    lib = _lib.load()
    ldx = _rows(x, "x")
    rows, C = x.shape
    out = torch.empty((rows, C), device=x.device, dtype=_lib.elem_dtype())
    check(lib.mi355x_sd_rmsnorm(x.data_ptr(), rows, C, ldx, _vec(weight, C, "weight").data_ptr(), float(eps), out.data_ptr(),
                                C, _stream()))
    return out


def gated_activation(x: Tensor, kind: str = "gelu_new") -> Tensor:
    """[rows, 2F] -> [rows, F]: act(x[:, :F]) * x[:, F:]."""
    lib = _lib.load()
    kinds = {"quick_gelu": 0, "gelu": 1, "silu": 2, "gelu_new": 3}
    ldx = _rows(x, "x")
    rows, F2 = x.shape
    out = torch.empty((rows, F2 // 2), device=x.device, dtype=_lib.elem_dtype())
    check(lib.mi355x_sd_gated_activation(x.data_ptr(), ldx, out.data_ptr(), F2 // 2, rows, F2 // 2, kinds[kind], _stream()))
    return out


def quantize_rows(x: Tensor):
    """bf16 [rows, C] -> (uint8 e4m3 [rows, C], fp32 scale [rows]) with value = scale * q, scale = absmax / 448."""
    lib = _lib.load()
    ldx = _rows(x, "x")
    rows, C = x.shape
    y = torch.empty((rows, C), device=x.device, dtype=torch.uint8)
    sc = torch.empty((rows,), device=x.device, dtype=torch.float32)
    check(lib.mi355x_sd_quantize_rows(x.data_ptr(), rows, C, ldx, 0, 0, y.data_ptr(), C, sc.data_ptr(), _stream()))
    return y, sc


def adaln_f8(x: Tensor, scale: Tensor, shift: Tensor, rows_per_batch: int, eps: float = 1e-6):
    """adaln() with fused e4m3 quantisation -> (uint8 [rows, C], fp32 scale [rows], fp32 row L2 norm [rows])."""
    lib = _lib.load()
    ldx = _rows(x, "x")
    rows, C = x.shape
    y = torch.empty((rows, C), device=x.device, dtype=torch.uint8)
    sc = torch.empty((rows,), device=x.device, dtype=torch.float32)
    l2 = torch.empty((rows,), device=x.device, dtype=torch.float32)
    check(lib.mi355x_sd_adaln_f8(x.data_ptr(), rows, C, ldx, scale.data_ptr(), shift.data_ptr(), scale.stride(0), rows_per_batch,
                                 float(eps), y.data_ptr(), C, sc.data_ptr(), l2.data_ptr(), _stream()))
    return y, sc, l2


def linear_f8_q(a8: Tensor, a_scale: Tensor, a_l2: Tensor, w8: Tensor, w_scale: Tensor, w_norm_max: float,
                bias: Optional[Tensor] = None, bias_abs_max: float = 0.0, *, gelu_tanh: bool = False):
    """W8A8 GEMM with an e4m3 output -> (uint8 [M, N], fp32 scale [M]); row scale = 1.1 * (a_l2 * w_norm_max +
    bias_abs_max) / 448 (include/mi355x_sd.h mi355x_sd_linear_f8_q)."""
    lib = _lib.load()
    M, K = a8.shape
    N = w8.shape[0]
    for t, nm in ((a8, "a8"), (w8, "w8")):
        if t.dtype != torch.uint8 or not t.is_cuda or t.stride(1) != 1:
            raise ValueError(f"{nm}: expected uint8 (e4m3 bytes) cuda rows")
    out = torch.empty((M, N), device=a8.device, dtype=torch.uint8)
    sc = torch.empty((M,), device=a8.device, dtype=torch.float32)
    check(lib.mi355x_sd_linear_f8_q(a8.data_ptr(), a8.stride(0), _vec(a_scale, M, "a_scale").data_ptr(),
                                    _vec(a_l2, M, "a_l2").data_ptr(), w8.data_ptr(), _vec(w_scale, N, "w_scale").data_ptr(),
                                    float(w_norm_max), out.data_ptr(), N, sc.data_ptr(), M, N, K, _p(_vec(bias, N, "bias")),
                                    float(bias_abs_max), GELU_TANH if gelu_tanh else 0, _stream()))
    return out, sc


def linear_f8(a8: Tensor, a_scale: Tensor, w8: Tensor, w_scale: Tensor, bias: Optional[Tensor] = None, *,
              gate: Optional[Tensor] = None, rows_per_batch: int = 0, residual: Optional[Tensor] = None,
              gelu_tanh: bool = False, out: Optional[Tensor] = None) -> Tensor:
    """W8A8 GEMM on the fp8 matrix pipe: (a8 @ w8^T) * a_scale[:, None] * w_scale[None, :] + bias, bf16 out."""
    lib = _lib.load()
    M, K = a8.shape
    N = w8.shape[0]
    for t, nm in ((a8, "a8"), (w8, "w8")):
        if t.dtype != torch.uint8 or not t.is_cuda or t.stride(1) != 1:
            raise ValueError(f"{nm}: expected uint8 (e4m3 bytes) cuda rows")
    if out is None:
        out = torch.empty((M, N), device=a8.device, dtype=_lib.elem_dtype())
    check(lib.mi355x_sd_linear_f8(a8.data_ptr(), a8.stride(0), 0, 0, _vec(a_scale, M, "a_scale").data_ptr(), w8.data_ptr(),
                                  _vec(w_scale, N, "w_scale").data_ptr(), out.data_ptr(), _rows(out, "out"), 0, 0, M, N, K,
                                  _p(_vec(bias, N, "bias")), _p(gate), gate.stride(0) if gate is not None else 0, rows_per_batch,
                                  _p(residual), _rows(residual, "residual") if residual is not None else 0,
                                  GELU_TANH if gelu_tanh else 0, _stream()))
    return out


def fused_adaLN_scale_residual(x: Tensor, mha_out: Tensor, gate_msa: Tensor, scale_mlp: Tensor, shift_mlp: Tensor,
                               weight: Optional[Tensor] = None, bias: Optional[Tensor] = None, epsilon: float = 1e-05):
    """The reference's fused op with its own signature and shape checks (paddlemix/triton_ops/triton_ops.py:758-920):
    ``resi_out = mha_out * gate_msa[:, None] + x``; ``adaLN_out = layer_norm(resi_out, weight, bias, epsilon) * (1 + scale_mlp[:, None])
    + shift_mlp[:, None]`` -> ``(resi_out, adaLN_out)``. x / mha_out [B, S, C] in the build's 16-bit type; gate / scale / shift [B, C]
    and weight / bias [C] in x.dtype as the reference passes them (:777-786) or fp32."""
    assert x.shape == mha_out.shape, "x and mha_out should have same shape"
    assert gate_msa.shape == scale_mlp.shape == shift_mlp.shape, "gate_msa, scale_mlp and shift_mlp should have same shape"
    assert x.dim() == 3, "x should be 3-dim [batch_size, seq_size, feature_dim]"
    B, S, C = x.shape
    if weight is not None:
        assert weight.dim() == 1 and weight.shape[-1] == C, "x and weight should have same shape[-1] == feature_dim"
    if bias is not None:
        assert bias.dim() == 1 and bias.shape[-1] == C, "x and bias should have same shape[-1] == feature_dim"
    assert scale_mlp.dim() == 2 and shift_mlp.dim() == 2, "scale and shift should be 2-dim [batch_size, feature_dim]"
    assert scale_mlp.shape[0] == B and scale_mlp.shape[1] == C, "x, scale and shift should have same batch_size / feature_dim"
    ed = _lib.elem_dtype()
    if x.dtype != ed or mha_out.dtype != ed or not x.is_cuda:
        raise ValueError(f"x / mha_out: expected {ed} cuda tensors")
    xx, mm = x.contiguous(), mha_out.contiguous()
    # gate / scale / shift / weight / bias go down in the type they come in: x.dtype in the reference (chunks of a 16-bit linear
    # output: strided views are passed as they are), fp32 from this library's own programs
    (g, sc, sh, w, b), ld_mod, md = _mod_vectors(x.device, gate_msa, scale_mlp, shift_mlp, weight, bias)
    resi, out = torch.empty_like(xx), torch.empty_like(xx)
    check(_lib.load().mi355x_sd_fused_adaln_scale_residual_ex(xx.data_ptr(), C, mm.data_ptr(), C, g.data_ptr(), sc.data_ptr(),
                                                              sh.data_ptr(), ld_mod, md, S, _p(w), _p(b), float(epsilon), B * S, C,
                                                              resi.data_ptr(), C, out.data_ptr(), C, _stream()))
    return resi, out


def split_concat(x: Tensor, y: Tensor):
    """The reference's ``split_concat(x [B,S1,3C], y [B,S2,3C]) -> (q, k, v)`` each [B, S1+S2, C] (triton_ops.py:1692-1752): what
    JointAttnProcessor does with the fused image / context QKV projections before attention (simplified_sd3.py:96-113)."""
    assert x.dim() == 3 and y.dim() == 3
    assert x.shape[0] == y.shape[0] and x.shape[2] == y.shape[2]
    B, S1, H3 = x.shape
    S2, C = y.shape[1], H3 // 3
    ed = _lib.elem_dtype()
    if x.dtype != ed or y.dtype != ed or not x.is_cuda or H3 % 3:
        raise ValueError(f"x / y: expected {ed} cuda tensors [B, S, 3C]")
    xx, yy = x.contiguous(), y.contiguous()
    outs = [torch.empty((B, S1 + S2, C), device=x.device, dtype=ed) for _ in range(3)]
    check(_lib.load().mi355x_sd_split_concat(xx.data_ptr(), yy.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(),
                                             B, S1, S2, C, _stream()))
    return tuple(outs)
