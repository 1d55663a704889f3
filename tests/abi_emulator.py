"""Host-memory interpreter of the C ABI in include/mi355x_sd.h -- TEST INFRASTRUCTURE ONLY.

It executes the same (function, raw-pointer args) program that paddlemix_amd.unet emits for the HIP library, but on
CPU memory with plain torch fp32 math, so the *host logic* (weight repacking, GEGLU interleave, fused QKV,
concat-by-construction strides, program order) can be checked against the oracle without a GPU.  It is not a
fallback: product code never constructs it (tests push it through paddlemix_amd.program._BACKEND_OVERRIDE, see on_emulator below).

``round_bf16=False`` keeps intermediate activations in fp32 buffers?  No -- buffers are bf16 by ABI; the emulator
therefore shows exactly the rounding points of the device path (bf16 stores, fp32 accumulation).
"""
from __future__ import annotations

import ctypes
import math

import os

import torch
import torch.nn.functional as F

from paddlemix_amd import _lib

GEGLU, OUT_F32, SILU, GELU_TANH, R_F32 = 1, 2, 4, 8, 32
_ES = {torch.bfloat16: 2, torch.float16: 2, torch.float32: 4}


def _flat(ptr: int, n: int, dtype) -> torch.Tensor:
    buf = (ctypes.c_char * (n * _ES[dtype])).from_address(ptr)
    return torch.frombuffer(buf, dtype=dtype, count=n)


def _rows(ptr: int, rows: int, C: int, ld: int, dtype=None) -> torch.Tensor:
    dtype = dtype or _lib.elem_dtype()   # the element type of the selected library build (bf16 | fp16)
    n = (rows - 1) * ld + C
    return _flat(ptr, n, dtype).as_strided((rows, C), (ld, 1))


class Emulator:
    IS_TEST_BACKEND = True   # what paddlemix_amd.program accepts in its override slot (anything else is refused)

    def __init__(self):
        self.calls = []

    # ---- bookkeeping ----
    def mi355x_sd_init(self, device):
        return 0

    def mi355x_sd_last_error(self):
        return b"emulator"

    def mi355x_sd_groupnorm_workspace_floats(self, B, HW, C):
        # the library's own answer (csrc/norm.hip groupnorm_partial_floats): a program planned here is replayed on the device
        # (tests/test_gpu_export.py), whose statistics kernel writes B * nblk * 128 partial sums. Until round 4 this returned 64:
        # the CPU-planned VAE program under-allocated its workspace 4x and passed or failed with whatever lay behind it.
        if C <= 0 or C % 8:
            return 0
        cv = C // 8
        ppp = 256 // cv if cv <= 256 else 1
        ppb = ppp * 16
        return B * ((HW + ppb - 1) // ppb) * 2 * 64

    def mi355x_sd_groupnorm_act_fits(self, HW, C, groups):
        cpg = C // groups
        # (csrc/norm.hip groupnorm_act_fits: even group width <= 128 -- the kernel's LDS table of a group's gamma / beta --, the
        # (batch, group) chunk within a block's registers)
        return int(cpg % 2 == 0 and cpg <= 128 and C % 8 == 0 and HW * (cpg // 2) <= 1024 * 24 and not os.environ.get("MI355X_SD_NO_GN_FUSED"))

    def mi355x_sd_groupnorm_act(self, x, B, HW, C, ldx, groups, eps, gamma, beta, silu, y, ldy, stream):
        ss = torch.empty(B * 2 * C, dtype=torch.float32)
        self.mi355x_sd_groupnorm_stats(x, B, HW, C, ldx, groups, eps, gamma, beta, None, ss.data_ptr(), stream)
        self.calls[-1] = "gn_fused"
        rc = self.mi355x_sd_scale_shift_act(x, B, HW, C, ldx, ss.data_ptr(), silu, y, ldy, stream)
        self.calls.pop()
        return rc

    # ---- GEMM family ----
    def _epilogue(self, acc, N, bias, rowbias, rpb, ld_rb, R, ldr, out_scale, flags, C, ldc):
        M = acc.shape[0]
        if bias:
            acc = acc + _flat(bias, N, torch.float32)
        if flags & GEGLU:
            a = acc.reshape(M, N // 32, 2, 16)
            acc = (a[:, :, 0] * F.gelu(a[:, :, 1])).reshape(M, N // 2)
            N = N // 2
        else:
            if rowbias:
                nb = (M + rpb - 1) // rpb
                rb = _rows(rowbias, nb, N, ld_rb, torch.float32)
                acc = acc + rb.repeat_interleave(rpb, 0)[:M]
            if R:
                acc = acc + _rows(R, M, N, ldr, torch.float32 if flags & R_F32 else None).float()
            acc = acc * out_scale
            if flags & SILU:
                acc = F.silu(acc)
        dt = torch.float32 if flags & OUT_F32 else _lib.elem_dtype()
        _rows(C, M, N, ldc, dt).copy_(acc.to(dt))

    def mi355x_sd_linear(self, A, lda, W, C, ldc, M, N, K, bias, rowbias, rpb, ld_rb, R, ldr, out_scale, flags, ws, ws_bytes, stream):
        self.calls.append("linear")
        assert K % 8 == 0 and N % 4 == 0 and lda % 8 == 0 and ldc % 4 == 0
        a = _rows(A, M, K, lda).float()
        w = _rows(W, N, K, K).float()
        self._epilogue(a @ w.t(), N, bias, rowbias, rpb, ld_rb, R, ldr, out_scale, flags, C, ldc)
        return 0

    def mi355x_sd_linear_ex(self, A, lda, a_rpb, a_bs, W, w_scale, C, ldc, c_rpb, c_bs, M, N, K, bias, rowbias, ld_rb, gate,
                            ld_gate, rpb, R, ldr, out_scale, flags, ws, ws_bytes, stream):
        self.calls.append("linear_ex")
        assert K % 8 == 0 and N % 4 == 0 and lda % 8 == 0 and ldc % 4 == 0
        assert not (flags & GEGLU) and not rowbias

        def remap_rows(ptr, rows, cols, ld, r_pb, bstride, dtype):
            if not r_pb:
                return _rows(ptr, rows, cols, ld, dtype)
            nb = rows // r_pb
            assert nb * r_pb == rows
            n = (nb - 1) * bstride + (r_pb - 1) * ld + cols
            return _flat(ptr, n, dtype).as_strided((nb, r_pb, cols), (bstride, ld, 1))

        a = remap_rows(A, M, K, lda, a_rpb, a_bs, _lib.elem_dtype()).float().reshape(M, K)
        if w_scale:
            buf = (ctypes.c_char * (N * K)).from_address(W)
            q = torch.frombuffer(buf, dtype=torch.uint8, count=N * K).reshape(N, K).view(torch.float8_e4m3fn).float()
            acc = (a @ q.t()) * _flat(w_scale, N, torch.float32)
        else:
            w = _rows(W, N, K, K).float()
            acc = a @ w.t()
        if bias:
            acc = acc + _flat(bias, N, torch.float32)
        if gate:
            g = _rows(gate, M // rpb, N, ld_gate, torch.float32)
            acc = acc * g.repeat_interleave(rpb, 0)
        if R:
            acc = acc + _rows(R, M, N, ldr).float()
        acc = acc * out_scale
        if flags & SILU:
            acc = F.silu(acc)
        if flags & GELU_TANH:
            acc = F.gelu(acc, approximate="tanh")
        dt = torch.float32 if flags & OUT_F32 else _lib.elem_dtype()
        out = remap_rows(C, M, N, ldc, c_rpb, c_bs, dt)
        out.copy_(acc.to(dt).reshape(out.shape))
        return 0

    def mi355x_sd_row_stats(self, x, rows, C, ldx, eps, stats, stream):
        xi = _rows(x, rows, C, ldx).float()
        mean = xi.mean(-1)
        rstd = torch.rsqrt(xi.var(-1, unbiased=False) + eps)
        _flat(stats, 2 * rows, torch.float32).copy_(torch.stack([rstd, -mean * rstd], 1).reshape(-1))
        return 0

    def mi355x_sd_linear_ln(self, A, lda, row_stats, W, w_rowsum, C, ldc, M, N, K, bias, flags, ws, ws_bytes, stream):
        self.calls.append("linear_ln")
        a = _rows(A, M, K, lda).float()
        w = _rows(W, N, K, K).float()
        st = _flat(row_stats, 2 * M, torch.float32).reshape(M, 2)
        acc = st[:, :1] * (a @ w.t()) + st[:, 1:] * _flat(w_rowsum, N, torch.float32)[None, :]
        assert not flags & GELU_TANH
        self._epilogue(acc, N, bias, None, 0, 0, None, 0, 1.0, flags, C, ldc)
        return 0

    # ---- W8A8 ----
    @staticmethod
    def _q8(v):
        sc = v.abs().amax(dim=1).clamp_min(1e-12) * (1.0 / 448.0)
        return (v / sc[:, None]).to(torch.float8_e4m3fn), sc

    @staticmethod
    def _u8(ptr, rows, C, ld):
        buf = (ctypes.c_char * ((rows - 1) * ld + C)).from_address(ptr)
        return torch.frombuffer(buf, dtype=torch.uint8, count=(rows - 1) * ld + C).as_strided((rows, C), (ld, 1))

    def mi355x_sd_adaln_f8(self, x, rows, C, ldx, scale, shift, ld_mod, rpb, eps, y8, ldy, y_scale, y_l2, stream):
        xv = _rows(x, rows, C, ldx).float()
        nb = (rows + rpb - 1) // rpb
        sc = _rows(scale, nb, C, ld_mod, torch.float32).repeat_interleave(rpb, 0)[:rows]
        sh = _rows(shift, nb, C, ld_mod, torch.float32).repeat_interleave(rpb, 0)[:rows]
        v = F.layer_norm(xv, (C,), eps=eps) * (1 + sc) + sh
        q, s = self._q8(v)
        self._u8(y8, rows, C, ldy).copy_(q.view(torch.uint8))
        _flat(y_scale, rows, torch.float32).copy_(s)
        if y_l2:
            _flat(y_l2, rows, torch.float32).copy_(v.norm(dim=1))
        return 0

    def mi355x_sd_linear_f8_q(self, A8, lda, a_scale, a_l2, W8, w_scale, w_norm_max, C8, ldc, c_scale, M, N, K, bias,
                              bias_abs_max, flags, stream):
        self.calls.append("linear_f8_q")
        assert K % 128 == 0 and ldc % 4 == 0
        a = self._u8(A8, M, K, lda).view(torch.float8_e4m3fn).float()
        w = self._u8(W8, N, K, K).view(torch.float8_e4m3fn).float()
        acc = (a @ w.t()) * _flat(a_scale, M, torch.float32)[:, None] * _flat(w_scale, N, torch.float32)[None, :]
        if bias:
            acc = acc + _flat(bias, N, torch.float32)
        if flags & GELU_TANH:
            acc = F.gelu(acc, approximate="tanh")
        sc = (1.1 * (_flat(a_l2, M, torch.float32) * w_norm_max + bias_abs_max)).clamp_min(1e-12) * (1.0 / 448.0)
        self._u8(C8, M, N, ldc).copy_((acc / sc[:, None]).to(torch.float8_e4m3fn).view(torch.uint8))
        _flat(c_scale, M, torch.float32).copy_(sc)
        return 0

    def mi355x_sd_quantize_rows(self, x, rows, C, ldx, x_rpb, x_bs, y8, ldy, y_scale, stream):
        if x_rpb:
            nb = (rows + x_rpb - 1) // x_rpb
            n = (nb - 1) * x_bs + (x_rpb - 1) * ldx + C
            xv = _flat(x, n, _lib.elem_dtype()).as_strided((nb, x_rpb, C), (x_bs, ldx, 1)).reshape(nb * x_rpb, C)[:rows].float()
        else:
            xv = _rows(x, rows, C, ldx).float()
        q, s = self._q8(xv)
        self._u8(y8, rows, C, ldy).copy_(q.view(torch.uint8))
        _flat(y_scale, rows, torch.float32).copy_(s)
        return 0

    def mi355x_sd_linear_f8(self, A8, lda, a_rpb, a_bs, a_scale, W8, w_scale, C, ldc, c_rpb, c_bs, M, N, K, bias, gate, ld_gate,
                            rpb, R, ldr, flags, stream):
        self.calls.append("linear_f8")
        assert K % 128 == 0 and not a_rpb
        a = self._u8(A8, M, K, lda).view(torch.float8_e4m3fn).float()
        w = self._u8(W8, N, K, K).view(torch.float8_e4m3fn).float()
        acc = (a @ w.t()) * _flat(a_scale, M, torch.float32)[:, None] * _flat(w_scale, N, torch.float32)[None, :]
        if bias:
            acc = acc + _flat(bias, N, torch.float32)
        if gate:
            nb = (M + rpb - 1) // rpb
            acc = acc * _rows(gate, nb, N, ld_gate, torch.float32).repeat_interleave(rpb, 0)[:M]
        if R:
            acc = acc + _rows(R, M, N, ldr).float()
        if flags & GELU_TANH:
            acc = F.gelu(acc, approximate="tanh")
        if c_rpb:
            nb = (M + c_rpb - 1) // c_rpb
            n = (nb - 1) * c_bs + (c_rpb - 1) * ldc + N
            dst = _flat(C, n, _lib.elem_dtype()).as_strided((nb, c_rpb, N), (c_bs, ldc, 1))
            dst.copy_(acc.reshape(nb, c_rpb, N).to(_lib.elem_dtype()))
        else:
            _rows(C, M, N, ldc).copy_(acc.to(_lib.elem_dtype()))
        return 0

    def mi355x_sd_adaln(self, x, rows, C, ldx, scale, shift, ld_mod, rpb, eps, y, ldy, stream):
        self.calls.append("adaln")
        nb = rows // rpb
        xv = F.layer_norm(_rows(x, rows, C, ldx).float(), (C,), None, None, eps)
        sc = _rows(scale, nb, C, ld_mod, torch.float32).repeat_interleave(rpb, 0)
        sh = _rows(shift, nb, C, ld_mod, torch.float32).repeat_interleave(rpb, 0)
        _rows(y, rows, C, ldy).copy_((xv * (1 + sc) + sh).to(_lib.elem_dtype()))
        return 0

    def mi355x_sd_patchify(self, x, B, C, H, W, p, out, ldo, stream):
        self.calls.append("patchify")
        xs = _flat(x, B * C * H * W, torch.float32).reshape(B, C, H // p, p, W // p, p)
        rows = xs.permute(0, 2, 4, 1, 3, 5).reshape(B * (H // p) * (W // p), C * p * p)
        _rows(out, rows.shape[0], rows.shape[1], ldo).copy_(rows.to(_lib.elem_dtype()))
        return 0

    def mi355x_sd_unpatchify(self, x, ldx, B, C, H, W, p, out, stream):
        self.calls.append("unpatchify")
        h, w = H // p, W // p
        rows = _rows(x, B * h * w, p * p * C, ldx).float().reshape(B, h, w, p, p, C)
        _flat(out, B * C * H * W, torch.float32).reshape(B, C, H, W).copy_(rows.permute(0, 5, 1, 3, 2, 4).reshape(B, C, H, W))
        return 0

    def mi355x_sd_conv3x3(self, X, ldx, B, Hs, Ws, Cin, stride, up, W, C, ldc, Cout, bias, rowbias, ld_rb, R, ldr,
                          out_scale, flags, ws, ws_bytes, stream):
        self.calls.append("conv3x3")
        assert Cin % 8 == 0 and ldx % 8 == 0
        x = _rows(X, B * Hs * Ws, Cin, ldx).float().reshape(B, Hs, Ws, Cin).permute(0, 3, 1, 2)
        if up:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        if flags & 64:   # MI355X_SD_CONV_KB64: [O][Cin/64][3][3][64]
            assert Cin % 64 == 0
            w = _rows(W, Cout, 9 * Cin, 9 * Cin).float().reshape(Cout, Cin // 64, 3, 3, 64).permute(0, 1, 4, 2, 3).reshape(Cout, Cin, 3, 3)
        else:
            w = _rows(W, Cout, 9 * Cin, 9 * Cin).float().reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
        if flags & 16:   # MI355X_SD_PAD_BR
            assert stride == 2 and not up
            y = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, None, stride=2, padding=0)
        else:
            y = F.conv2d(x, w, None, stride=stride, padding=1)
        Ho, Wo = y.shape[2], y.shape[3]
        acc = y.permute(0, 2, 3, 1).reshape(B * Ho * Wo, Cout)
        self._epilogue(acc, Cout, bias, rowbias, Ho * Wo, ld_rb, R, ldr, out_scale, flags, C, ldc)
        return 0

    def mi355x_sd_sdpa_accum(self, q, k, v, bias, out, B, H, Sq, Skv, D, q_bs, q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts,
                             bias_bs, bias_hs, bias_qs, scale, out_scale, stream):
        return self.mi355x_sd_sdpa(q, k, v, bias, out, B, H, Sq, Skv, D, q_bs, q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts,
                                   bias_bs, bias_hs, bias_qs, scale, stream, _accum=out_scale)

    def mi355x_sd_sdpa_ex(self, q, k, v, bias, out, B, H, Sq, Skv, D, q_bs, q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts,
                          bias_bs, bias_hs, bias_qs, scale, flags, stream):
        if flags & 1:   # MI355X_SD_SDPA_LOG2: scores are base-2 exponents -> softmax(x ln 2)
            assert not bias and D == 64
            scale = math.log(2.0)
        return self.mi355x_sd_sdpa(q, k, v, bias, out, B, H, Sq, Skv, D, q_bs, q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts,
                                   bias_bs, bias_hs, bias_qs, scale, stream)

    def mi355x_sd_sdpa(self, q, k, v, bias, out, B, H, Sq, Skv, D, q_bs, q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts,
                       bias_bs, bias_hs, bias_qs, scale, stream, _accum=None):
        self.calls.append("sdpa")

        def view(p, S, bs, ts):
            n = (B - 1) * bs + (S - 1) * ts + H * D
            return _flat(p, n, _lib.elem_dtype()).as_strided((B, S, H, D), (bs, ts, D, 1))

        qq, kk, vv = view(q, Sq, q_bs, q_ts).float(), view(k, Skv, k_bs, k_ts).float(), view(v, Skv, v_bs, v_ts).float()
        s = torch.einsum("bqhd,bkhd->bhqk", qq, kk) * scale
        if bias:
            # general additive mask: bias[b*bias_bs + h*bias_hs + q*bias_qs + kv] (0 strides broadcast)
            n = (B - 1) * bias_bs + (H - 1) * bias_hs + (Sq - 1) * bias_qs + Skv
            s = s + _flat(bias, n, torch.float32).as_strided((B, H, Sq, Skv), (bias_bs, bias_hs, bias_qs, 1))
        p = torch.softmax(s, -1)
        o = torch.einsum("bhqk,bkhd->bqhd", p, vv)
        if _accum is not None:   # out += out_scale * attention, on the 16-bit values already in `out`
            o = view(out, Sq, o_bs, o_ts).float() + _accum * o
        view(out, Sq, o_bs, o_ts).copy_(o.to(_lib.elem_dtype()))
        return 0

    # ---- norms ----
    def mi355x_sd_groupnorm_stats_ex(self, x, B, HW, C, ldx, groups, eps, gamma, beta, ws, ss, x_f32, stream):
        return self.mi355x_sd_groupnorm_stats(x, B, HW, C, ldx, groups, eps, gamma, beta, ws, ss, stream, _f32=bool(x_f32))

    def mi355x_sd_scale_shift_act_ex(self, x, B, HW, C, ldx, ss, silu, y, ldy, x_f32, raw16, ld_raw, stream):
        if raw16:
            _rows(raw16, B * HW, C, ld_raw).copy_(_rows(x, B * HW, C, ldx, torch.float32 if x_f32 else None).to(_lib.elem_dtype()))
        return self.mi355x_sd_scale_shift_act(x, B, HW, C, ldx, ss, silu, y, ldy, stream, _f32=bool(x_f32))

    def mi355x_sd_layernorm_ex(self, x, rows, C, ldx, gamma, beta, eps, y, ldy, x_f32, stream):
        return self.mi355x_sd_layernorm(x, rows, C, ldx, gamma, beta, eps, y, ldy, stream, _f32=bool(x_f32))

    def mi355x_sd_cast_rows(self, x, ldx, y, ldy, rows, C, stream):
        self.calls.append("cast_rows")
        _rows(y, rows, C, ldy).copy_(_rows(x, rows, C, ldx, torch.float32).to(_lib.elem_dtype()))
        return 0

    def mi355x_sd_groupnorm_stats(self, x, B, HW, C, ldx, groups, eps, gamma, beta, ws, ss, stream, _f32=False):
        self.calls.append("gn_stats")
        xv = _rows(x, B * HW, C, ldx, torch.float32 if _f32 else None).float().reshape(B, HW, groups, C // groups)
        mean = xv.mean(dim=(1, 3))
        var = xv.var(dim=(1, 3), unbiased=False)
        rstd = (var + eps).rsqrt()
        g, b = _flat(gamma, C, torch.float32), _flat(beta, C, torch.float32)
        cpg = C // groups
        scale = g[None] * rstd.repeat_interleave(cpg, 1)
        shift = b[None] - mean.repeat_interleave(cpg, 1) * scale
        _flat(ss, B * 2 * C, torch.float32).reshape(B, 2, C).copy_(torch.stack([scale, shift], 1))
        return 0

    def mi355x_sd_scale_shift_act(self, x, B, HW, C, ldx, ss, silu, y, ldy, stream, _f32=False):
        self.calls.append("scale_shift_act")
        xv = _rows(x, B * HW, C, ldx, torch.float32 if _f32 else None).float().reshape(B, HW, C)
        s = _flat(ss, B * 2 * C, torch.float32).reshape(B, 2, C)
        o = xv * s[:, 0:1] + s[:, 1:2]
        if silu:
            o = F.silu(o)
        _rows(y, B * HW, C, ldy).copy_(o.reshape(B * HW, C).to(_lib.elem_dtype()))
        return 0

    def mi355x_sd_layernorm(self, x, rows, C, ldx, gamma, beta, eps, y, ldy, stream, _f32=False):
        self.calls.append("layernorm")
        g = _flat(gamma, C, torch.float32) if gamma else None
        b = _flat(beta, C, torch.float32) if beta else None
        o = F.layer_norm(_rows(x, rows, C, ldx, torch.float32 if _f32 else None).float(), (C,), g, b, eps)
        _rows(y, rows, C, ldy).copy_(o.to(_lib.elem_dtype()))
        return 0

    # ---- small ops ----
    def mi355x_sd_timestep_embedding(self, t, t_count, n, dim, group, flip, freq_shift, scale, max_period, out, ldo,
                                     stream):
        self.calls.append("timestep_embedding")
        tt = _flat(t, t_count, torch.float32)
        idx = torch.arange(n) % t_count
        half = dim // 2
        exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / (half - freq_shift)
        emb = scale * (tt[idx][:, None] * torch.exp(exponent)[None])
        e = torch.cat([torch.cos(emb), torch.sin(emb)], -1) if flip else torch.cat([torch.sin(emb), torch.cos(emb)], -1)
        nb = (n + group - 1) // group
        o = _rows(out, nb, group * dim, ldo)
        o.copy_(e.reshape(nb, group * dim).to(_lib.elem_dtype()))
        return 0

    def mi355x_sd_silu(self, x, y, n, in_f32, out_f32, stream):
        self.calls.append("silu")
        xi = _flat(x, n, torch.float32 if in_f32 else _lib.elem_dtype()).float()
        _flat(y, n, torch.float32 if out_f32 else _lib.elem_dtype()).copy_(F.silu(xi))
        return 0

    def mi355x_sd_conv_in3x3_ex(self, x, in_scale, w, bias, y, B, Cin, H, W, Cout, ldy, out_f32, stream):
        return self.mi355x_sd_conv_in3x3(x, in_scale, w, bias, y, B, Cin, H, W, Cout, ldy, stream, _f32=bool(out_f32))

    def mi355x_sd_conv_in3x3(self, x, in_scale, w, bias, y, B, Cin, H, W, Cout, ldy, stream, _f32=False):
        self.calls.append("conv_in")
        xs = _flat(x, B * Cin * H * W, torch.float32).reshape(B, Cin, H, W)
        if in_scale:
            xs = xs * _flat(in_scale, 1, torch.float32)
        xs = xs.to(_lib.elem_dtype()).float()
        wt = _flat(w, 9 * Cin * Cout, _lib.elem_dtype()).float().reshape(3, 3, Cin, Cout).permute(3, 2, 0, 1)
        o = F.conv2d(xs, wt, _flat(bias, Cout, torch.float32) if bias else None, padding=1)
        dt = torch.float32 if _f32 else _lib.elem_dtype()
        _rows(y, B * H * W, Cout, ldy, dt).copy_(o.permute(0, 2, 3, 1).reshape(B * H * W, Cout).to(dt))
        return 0

    def mi355x_sd_conv_out3x3(self, x, ldx, w, bias, y, B, Cin, H, W, Cout, stream):
        self.calls.append("conv_out")
        xs = _rows(x, B * H * W, Cin, ldx).float().reshape(B, H, W, Cin).permute(0, 3, 1, 2)
        wt = _flat(w, Cout * 9 * Cin, _lib.elem_dtype()).float().reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
        o = F.conv2d(xs, wt, _flat(bias, Cout, torch.float32) if bias else None, padding=1)
        _flat(y, B * Cout * H * W, torch.float32).reshape(B, Cout, H, W).copy_(o)
        return 0

    def mi355x_sd_add_nchw_ex(self, x, ldx, r, B, C, HW, x_f32, stream):
        return self.mi355x_sd_add_nchw(x, ldx, r, B, C, HW, stream, _f32=bool(x_f32))

    def mi355x_sd_add_nchw(self, x, ldx, r, B, C, HW, stream, _f32=False):
        xv = _rows(x, B * HW, C, ldx, torch.float32 if _f32 else None)
        rv = _flat(r, B * C * HW, torch.float32).reshape(B, C, HW).permute(0, 2, 1).reshape(B * HW, C)
        xv.copy_((xv.float() + rv).to(xv.dtype))
        return 0

    def mi355x_sd_latent_dist(self, m, ld, B, L, HW, noise, out_scale, mean, logvar, sample, stream):
        mv = _flat(m, B * HW * ld, torch.float32).reshape(B, HW, ld)
        mu = mv[:, :, :L].permute(0, 2, 1).reshape(-1)
        lv = mv[:, :, L:2 * L].permute(0, 2, 1).reshape(-1).clamp(-30.0, 20.0)
        _flat(mean, B * L * HW, torch.float32).copy_(mu)
        _flat(logvar, B * L * HW, torch.float32).copy_(lv)
        if sample:
            x = mu + (torch.exp(0.5 * lv) * _flat(noise, B * L * HW, torch.float32) if noise else 0.0)
            _flat(sample, B * L * HW, torch.float32).copy_(x * out_scale)
        return 0

    def mi355x_sd_embed_tokens(self, ids, n_tokens, seq_len, tok, pos, D, out, ldo, stream):
        buf = (ctypes.c_char * (4 * n_tokens)).from_address(ids)
        idx = torch.frombuffer(buf, dtype=torch.int32, count=n_tokens).long()
        V = int(idx.max()) + 1
        t = _rows(tok, V, D, D).float()[idx]
        if pos:
            t = t + _rows(pos, seq_len, D, D).float()[torch.arange(n_tokens) % seq_len]
        _rows(out, n_tokens, D, ldo).copy_(t.to(_lib.elem_dtype()))
        return 0

    def mi355x_sd_rmsnorm(self, x, rows, C, ldx, weight, eps, y, ldy, stream):
        xv = _rows(x, rows, C, ldx).float()
        o = xv * torch.rsqrt(xv.pow(2).mean(-1, keepdim=True) + eps) * _flat(weight, C, torch.float32)
        _rows(y, rows, C, ldy).copy_(o.to(_lib.elem_dtype()))
        return 0

    def mi355x_sd_gated_activation(self, x, ldx, y, ldy, rows, Fd, kind, stream):
        xv = _rows(x, rows, 2 * Fd, ldx).float()
        a, b = xv[:, :Fd], xv[:, Fd:]
        g = (a * torch.sigmoid(1.702 * a) if kind == 0 else F.gelu(a) if kind == 1 else F.silu(a) if kind == 2
             else F.gelu(a, approximate="tanh"))
        _rows(y, rows, Fd, ldy).copy_((g * b).to(_lib.elem_dtype()))
        return 0

    def mi355x_sd_activation(self, x, y, n, kind, stream):
        v = _flat(x, n, _lib.elem_dtype()).float()
        o = v * torch.sigmoid(1.702 * v) if kind == 0 else (F.gelu(v) if kind == 1 else F.silu(v))
        _flat(y, n, _lib.elem_dtype()).copy_(o.to(_lib.elem_dtype()))
        return 0

    def mi355x_sd_conv1x1_nchw(self, x, in_scale, w, bias, y, B, Cin, Cout, HW, stream):
        assert Cin <= 16 and Cout <= 16
        xi = (_flat(x, B * Cin * HW, torch.float32).reshape(B, Cin, HW) * in_scale).to(_lib.elem_dtype()).float()
        wt = _flat(w, Cout * Cin, _lib.elem_dtype()).reshape(Cout, Cin).float()
        out = torch.einsum("oc,bcp->bop", wt, xi)
        if bias:
            out = out + _flat(bias, Cout, torch.float32)[None, :, None]
        _flat(y, B * Cout * HW, torch.float32).copy_(out.reshape(-1))
        return 0

    def mi355x_sd_softmax_rows(self, x, ldx, y, ldy, rows, n, stream):
        assert n % 4 == 0 and ldx % 4 == 0 and ldy % 4 == 0
        xi = _rows(x, rows, n, ldx, torch.float32)
        _rows(y, rows, n, ldy).copy_(torch.softmax(xi, -1).to(_lib.elem_dtype()))
        return 0

    def mi355x_sd_copy_rows(self, x, ldx, y, ldy, rows, C, stream):
        self.calls.append("copy_rows")
        _rows(y, rows, C, ldy).copy_(_rows(x, rows, C, ldx))
        return 0

    def mi355x_sd_cfg_axpby(self, x, eu, et, out, coef, gs, n, stream):
        a, b = _flat(coef, 2, torch.float32).tolist()
        u, t = _flat(eu, n, torch.float32), _flat(et, n, torch.float32)
        _flat(out, n, torch.float32).copy_(a * _flat(x, n, torch.float32) + b * (u + gs * (t - u)))
        return 0

    def mi355x_sd_axpby(self, x, y, out, coef, n, stream):
        self.calls.append("axpby")
        c = _flat(coef, 2, torch.float32)
        _flat(out, n, torch.float32).copy_(c[0] * _flat(x, n, torch.float32) + c[1] * _flat(y, n, torch.float32))
        return 0


def on_emulator(ctor, *args, backend=None, **kwargs):
    """``ctor(*args, **kwargs)`` -- a model class or its ``from_pretrained`` -- with the C-ABI calls of the model it builds routed
    to ``backend`` (a fresh Emulator by default) instead of the HIP library: the one test hook of the product
    (paddlemix_amd/program.py _BACKEND_OVERRIDE), held only for the duration of the constructor call."""
    from paddlemix_amd import program
    token = program._BACKEND_OVERRIDE.set(backend if backend is not None else Emulator())   # a ContextVar: this thread only
    try:
        return ctor(*args, **kwargs)
    finally:
        program._BACKEND_OVERRIDE.reset(token)
