// Shared GEMM epilogue (fp8 weight scale / bias / time-embedding row bias / residual / scale / SiLU / GEGLU, bf16 or fp32 stores).
// Accumulator layout: acc[tn][tm] is the "swapped" 16x16 MFMA tile whose lane holds, for output row
// m = m_wave + tm*16 + (lane&15), the 4 consecutive channels n = n_wave + tn*16 + (lane>>4)*4 + {0..3}.
#pragma once
#include "common.h"
#include "kernels.h"

namespace sd {

static __device__ __attribute__((aligned(16))) const unsigned g_zero16[4] = {0u, 0u, 0u, 0u};
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[TN][TM], int m_wave, int n_wave,
                                              int lane) {
  const int nq = (lane >> 4) * 4;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int m = m_wave + tm * 16 + (lane & 15);
    if (m >= p.M) continue;
    const float* rb = p.rowbias ? p.rowbias + (size_t)(m / p.rows_per_batch) * p.ld_rowbias : nullptr;
    const float* gt = p.gate ? p.gate + (size_t)(m / p.rows_per_batch) * p.ld_gate : nullptr;
    const size_t crow = p.c_rpb ? (size_t)(m / p.c_rpb) * p.c_bstride + (size_t)(m % p.c_rpb) * p.ldc : (size_t)m * p.ldc;
    if (p.geglu) {
#pragma unroll
      for (int tp = 0; tp < TN / 2; ++tp) {
        const int n_phys = n_wave + tp * 32 + nq;  // physical (interleaved) column of the value half
        if (n_phys >= p.N) continue;               // N % 32 == 0: the gate half of the pair is inside too
        f32x4 h = acc[2 * tp][tm], g = acc[(2 * tp + 1) % TN][tm];   // (% TN: odd-TN configs never take this path)
        if (p.wscale) {
          h *= *reinterpret_cast<const f32x4*>(p.wscale + n_phys);
          g *= *reinterpret_cast<const f32x4*>(p.wscale + n_phys + 16);
        }
        if (p.bias) {
          h += *reinterpret_cast<const f32x4*>(p.bias + n_phys);
          g += *reinterpret_cast<const f32x4*>(p.bias + n_phys + 16);
        }
        const int n_out = (n_wave >> 1) + tp * 16 + nq;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = h[r] * gelu_erf_f(g[r]);
        u32x2 pk = {pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3])};
        *reinterpret_cast<u32x2*>(reinterpret_cast<bf16*>(p.C) + crow + n_out) = pk;
      }
    } else {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int n = n_wave + tn * 16 + nq;
        if (n >= p.N) continue;
        f32x4 v = acc[tn][tm];
        if (p.wscale) v *= *reinterpret_cast<const f32x4*>(p.wscale + n);
        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
        if (rb) v += *reinterpret_cast<const f32x4*>(rb + n);
        if (gt) v *= *reinterpret_cast<const f32x4*>(gt + n);
        if (p.R) {
          const bf16x4 r4 = *reinterpret_cast<const bf16x4*>(p.R + (size_t)m * p.ldr + n);
          v[0] += (float)r4[0];
          v[1] += (float)r4[1];
          v[2] += (float)r4[2];
          v[3] += (float)r4[3];
        }
        v *= p.out_scale;
        if (p.silu) {
          v[0] = silu_f(v[0]);
          v[1] = silu_f(v[1]);
          v[2] = silu_f(v[2]);
          v[3] = silu_f(v[3]);
        }
        if (p.gelu_tanh) {
          v[0] = gelu_tanh_f(v[0]);
          v[1] = gelu_tanh_f(v[1]);
          v[2] = gelu_tanh_f(v[2]);
          v[3] = gelu_tanh_f(v[3]);
        }
        if (p.out_f32) {
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + crow + n) = v;
        } else {
          u32x2 pk = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
          *reinterpret_cast<u32x2*>(reinterpret_cast<bf16*>(p.C) + crow + n) = pk;
        }
      }
    }
  }
}

// Epilogue of the LayerNorm-folded projections (mi355x_sd_linear_ln): acc <- rstd[m] * acc - mean[m] * rstd[m] * wsum[n]
// + bias[n], then GEGLU / SiLU / tanh-GELU and the store. A separate function (and separate kernel instantiations,
// template parameter LN) so the register allocation of every other GEMM stays exactly what it was. The row statistics
// of all TM row-tiles are fetched up front with clamped, branch-free addresses: the tm loop then has one load round
// trip per row-tile (wsum / bias, L1 hits) instead of two serialised ones.
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_ln(const GemmArgs& p, f32x4 (&acc)[TN][TM], int m_wave, int n_wave,
                                                 int lane) {
  const int nq = (lane >> 4) * 4;
  f32x2 rs[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
    rs[tm] = *reinterpret_cast<const f32x2*>(p.rowstat + 2 * (size_t)min(m_wave + tm * 16 + (lane & 15), p.M - 1));
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int m = m_wave + tm * 16 + (lane & 15);
    if (m >= p.M) continue;
    const size_t crow = (size_t)m * p.ldc;
    if (p.geglu) {
#pragma unroll
      for (int tp = 0; tp < TN / 2; ++tp) {
        const int n_phys = n_wave + tp * 32 + nq;
        if (n_phys >= p.N) continue;
        f32x4 h = acc[2 * tp][tm], g = acc[(2 * tp + 1) % TN][tm];
        h = rs[tm][0] * h + rs[tm][1] * *reinterpret_cast<const f32x4*>(p.wsum + n_phys);
        g = rs[tm][0] * g + rs[tm][1] * *reinterpret_cast<const f32x4*>(p.wsum + n_phys + 16);
        if (p.bias) {
          h += *reinterpret_cast<const f32x4*>(p.bias + n_phys);
          g += *reinterpret_cast<const f32x4*>(p.bias + n_phys + 16);
        }
        const int n_out = (n_wave >> 1) + tp * 16 + nq;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = h[r] * gelu_erf_f(g[r]);
        u32x2 pk = {pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3])};
        *reinterpret_cast<u32x2*>(reinterpret_cast<bf16*>(p.C) + crow + n_out) = pk;
      }
    } else {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int n = n_wave + tn * 16 + nq;
        if (n >= p.N) continue;
        f32x4 v = rs[tm][0] * acc[tn][tm] + rs[tm][1] * *reinterpret_cast<const f32x4*>(p.wsum + n);
        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
        if (p.silu) {
          v[0] = silu_f(v[0]);
          v[1] = silu_f(v[1]);
          v[2] = silu_f(v[2]);
          v[3] = silu_f(v[3]);
        }
        if (p.gelu_tanh) {
          v[0] = gelu_tanh_f(v[0]);
          v[1] = gelu_tanh_f(v[1]);
          v[2] = gelu_tanh_f(v[2]);
          v[3] = gelu_tanh_f(v[3]);
        }
        if (p.out_f32) {
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + crow + n) = v;
        } else {
          u32x2 pk = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
          *reinterpret_cast<u32x2*>(reinterpret_cast<bf16*>(p.C) + crow + n) = pk;
        }
      }
    }
  }
}

// Epilogue of the W8A8 GEMM (mi355x_sd_linear_f8): acc * ascale[m] * wscale[n] + bias, optional gate / residual /
// tanh-GELU / row-remapped C; bf16 store. Own function + own kernel instantiation (see gemm_epilogue_ln).
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_f8(const GemmArgs& p, f32x4 (&acc)[TN][TM], int m_wave, int n_wave,
                                                 int lane) {
  const int nq = (lane >> 4) * 4;
  float as[TM], oinv[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int mc = min(m_wave + tm * 16 + (lane & 15), p.M - 1);
    as[tm] = p.ascale[mc];
    oinv[tm] = 0.f;
    if (p.out_f8) {   // safe row scale for the e4m3 output (1.1: the quantised operands may exceed their fp32 norms slightly)
      const float bound = 1.1f * (p.a_l2[mc] * p.w_norm_max + p.bias_abs_max);
      oinv[tm] = 448.0f / fmaxf(bound, 1e-12f);
      if (n_wave == 0 && (lane >> 4) == 0 && m_wave + tm * 16 + (lane & 15) < p.M) p.oscale[mc] = fmaxf(bound, 1e-12f) * (1.0f / 448.0f);
    }
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int m = m_wave + tm * 16 + (lane & 15);
    if (m >= p.M) continue;
    const float* gt = p.gate ? p.gate + (size_t)(m / p.rows_per_batch) * p.ld_gate : nullptr;
    const size_t crow = p.c_rpb ? (size_t)(m / p.c_rpb) * p.c_bstride + (size_t)(m % p.c_rpb) * p.ldc : (size_t)m * p.ldc;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int n = n_wave + tn * 16 + nq;
      if (n >= p.N) continue;
      f32x4 v = acc[tn][tm] * as[tm] * *reinterpret_cast<const f32x4*>(p.wscale + n);
      if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
      if (gt) v *= *reinterpret_cast<const f32x4*>(gt + n);
      if (p.R) {
        const bf16x4 r4 = *reinterpret_cast<const bf16x4*>(p.R + (size_t)m * p.ldr + n);
        v[0] += (float)r4[0];
        v[1] += (float)r4[1];
        v[2] += (float)r4[2];
        v[3] += (float)r4[3];
      }
      if (p.gelu_tanh) {
        v[0] = gelu_tanh_f(v[0]);
        v[1] = gelu_tanh_f(v[1]);
        v[2] = gelu_tanh_f(v[2]);
        v[3] = gelu_tanh_f(v[3]);
      }
      if (p.out_f8) {
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * oinv[tm], v[1] * oinv[tm], w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * oinv[tm], v[3] * oinv[tm], w, true);
        *reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(p.C) + crow + n) = w;
      } else {
        u32x2 pk = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
        *reinterpret_cast<u32x2*>(reinterpret_cast<bf16*>(p.C) + crow + n) = pk;
      }
    }
  }
}

}  // namespace sd
