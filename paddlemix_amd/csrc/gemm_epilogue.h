// Shared GEMM epilogue (fp8 weight scale / bias / time-embedding row bias / residual / scale / SiLU / GEGLU, bf16 or fp32 stores).
// Accumulator layout: acc[tn][tm] is the "swapped" 16x16 MFMA tile whose lane holds 4 consecutive channels of output row
// m = m_wave + tm*16 + (lane&15). WHICH channels is a free choice -- MFMA row i of n-sub-tile tn is whatever weight row
// the LDS-DMA put into LDS row tn*16+i -- and it is chosen for the stores: sub-tiles are paired so that a lane owns 8
// CONSECUTIVE channels (acc[2h] | acc[2h+1]) = one 16-byte bf16 store, and the four lane groups of a row cover 64
// contiguous bytes per store instruction (scripts/probes/store_probe.hip: 5.8 vs 3.75 TB/s for the 8-byte / 32-byte-
// per-row pattern of the natural mapping). acc_col() is the single definition of the mapping; the DMA source rows
// (w_row_of_lds_row), the split-K partial sums and the epilogues all derive from it.
// Loads: each optional operand sits in its own uniform branch (load, wait, use) per 4-channel group. Batching them
// (branch-free buffer loads through zero-sized descriptors for absent operands, 8-16 channels per lane in flight) was
// built and measured (profiles/experiments/r01_gemm_epilogue_batched_loads.h.txt): no spills, parity green, but no
// gain -- SDXL step 63.4 vs 63.5 ms (box noise), SD3 W8A8 68.0 vs 65.1 ms (slower): the eight waves of a block already
// hide each other's epilogue latency, and the unconditional loads of absent operands are not free. Two lessons kept
// from it: (1) never give the accumulators two alternative consumer loops (a fast path next to a general one, or an
// e4m3 and a bf16 store loop) -- the register allocator then splits their live ranges and spills them INSIDE the K loop;
// (2) per-channel operands folded into all accumulators up front pin the whole accumulator set in VGPRs.
#pragma once
#include "common.h"
#include "kernels.h"

namespace sd {

static __device__ __attribute__((aligned(16))) const unsigned g_zero16[4] = {0u, 0u, 0u, 0u};
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// first channel (relative to the wave's n-range) of the 4 values lane-group lq holds in acc[tn]
template <int TN>
__device__ __forceinline__ int acc_col(int tn, int lq, bool geglu) {
  if (!geglu) return tn < (TN & ~1) ? (tn >> 1) * 32 + lq * 8 + (tn & 1) * 4 : tn * 16 + lq * 4;
  // GEGLU weights are stored in interleaved groups of 32 rows (16 value | 16 gate): acc[2tp] = value, acc[2tp+1] =
  // gate of the same 4 OUTPUT channels ol..ol+3 (output pairs paired again for 16-byte stores)
  const int tp = tn >> 1;
  const int ol = tp < ((TN / 2) & ~1) ? (tp >> 1) * 32 + lq * 8 + (tp & 1) * 4 : tp * 16 + lq * 4;
  return (ol >> 4) * 32 + (ol & 15) + (tn & 1) * 16;
}
// weight row (relative to the block tile) that belongs into LDS row j of the W tile
template <int TN>
__device__ __forceinline__ int w_row_of_lds_row(int j, bool geglu) {
  const int wn = j / (TN * 16), jl = j - wn * (TN * 16);
  const int i = jl & 15;
  return wn * (TN * 16) + acc_col<TN>(jl >> 4, i >> 2, geglu) + (i & 3);
}

// bias / row bias / gate / residual / scale / activation on 4 channels n..n+3 of row m
__device__ __forceinline__ f32x4 act4(const GemmArgs& p, f32x4 v) {
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
  return v;
}
__device__ __forceinline__ f32x4 epi4(const GemmArgs& p, f32x4 v, int m, int n, const float* rb, const float* gt) {
  if (p.wscale) v *= *reinterpret_cast<const f32x4*>(p.wscale + n);
  if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
  if (rb) v += *reinterpret_cast<const f32x4*>(rb + n);
  if (gt) v *= *reinterpret_cast<const f32x4*>(gt + n);
  if (p.R) {
    if (p.r_f32) {
      v += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.R) + (size_t)m * p.ldr + n);
    } else {
      const bf16x4 r4 = *reinterpret_cast<const bf16x4*>(p.R + (size_t)m * p.ldr + n);
      v[0] += (float)r4[0];
      v[1] += (float)r4[1];
      v[2] += (float)r4[2];
      v[3] += (float)r4[3];
    }
  }
  return act4(p, v * p.out_scale);
}
__device__ __forceinline__ void store4(const GemmArgs& p, size_t off, f32x4 v) {
  if (p.out_f32) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + off) = v;
  } else {
    u32x2 pk = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
    *reinterpret_cast<u32x2*>(reinterpret_cast<bf16*>(p.C) + off) = pk;
  }
}
// 8 consecutive channels; one 16-byte store when the destination allows it (p.c_wide: C, ldc, c_bstride 16-byte aligned)
__device__ __forceinline__ void store8(const GemmArgs& p, size_t off, f32x4 lo, f32x4 hi) {
  if (p.out_f32) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + off) = lo;
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + off + 4) = hi;
  } else if (p.c_wide) {
    u32x4 pk = {pack_bf16(lo[0], lo[1]), pack_bf16(lo[2], lo[3]), pack_bf16(hi[0], hi[1]), pack_bf16(hi[2], hi[3])};
    *reinterpret_cast<u32x4*>(reinterpret_cast<bf16*>(p.C) + off) = pk;
  } else {
    u32x2 p0 = {pack_bf16(lo[0], lo[1]), pack_bf16(lo[2], lo[3])}, p1 = {pack_bf16(hi[0], hi[1]), pack_bf16(hi[2], hi[3])};
    *reinterpret_cast<u32x2*>(reinterpret_cast<bf16*>(p.C) + off) = p0;
    *reinterpret_cast<u32x2*>(reinterpret_cast<bf16*>(p.C) + off + 4) = p1;
  }
}
__device__ __forceinline__ f32x4 geglu4(f32x4 h, f32x4 g) {
  f32x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = h[r] * gelu_erf_f(g[r]);
  return o;
}
// stores of one row-tile: GEGLU pairs or plain channels, 8 consecutive channels per lane where sub-tiles pair up.
// fin(tn, n) returns the finished 4 values of acc[tn] (first channel n; the physical weight column for GEGLU).
template <int TN, class F>
__device__ __forceinline__ void store_row(const GemmArgs& p, size_t crow, int n_wave, int lq, F&& fin) {
  if (p.geglu) {
#pragma unroll
    for (int hp = 0; hp < TN / 4; ++hp) {   // two output pairs = 8 consecutive output channels
      const int n_phys = n_wave + acc_col<TN>(4 * hp, lq, true);
      if (n_phys >= p.N) continue;          // N % 32 == 0: the whole interleaved group is inside
      store8(p, crow + (n_wave >> 1) + hp * 32 + lq * 8, geglu4(fin(4 * hp, n_phys), fin(4 * hp + 1, n_phys + 16)),
             geglu4(fin((4 * hp + 2) % TN, n_phys + 4), fin((4 * hp + 3) % TN, n_phys + 20)));
    }
    if constexpr ((TN / 2) & 1) {
      constexpr int tp = TN / 2 - 1;
      const int n_phys = n_wave + tp * 32 + lq * 4;
      if (n_phys < p.N)
        store4(p, crow + (n_wave >> 1) + tp * 16 + lq * 4, geglu4(fin(2 * tp, n_phys), fin((2 * tp + 1) % TN, n_phys + 16)));
    }
  } else {
#pragma unroll
    for (int h = 0; h < TN / 2; ++h) {
      const int n = n_wave + h * 32 + lq * 8;
      if (n >= p.N) continue;
      const f32x4 lo = fin(2 * h, n);
      if (n + 4 < p.N) store8(p, crow + n, lo, fin(2 * h + 1, n + 4));
      else store4(p, crow + n, lo);
    }
    if constexpr (TN & 1) {
      const int n = n_wave + (TN - 1) * 16 + lq * 4;
      if (n < p.N) store4(p, crow + n, fin(TN - 1, n));
    }
  }
}

template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[TN][TM], int m_wave, int n_wave,
                                              int lane) {
  const int lq = lane >> 4;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int m = m_wave + tm * 16 + (lane & 15);
    if (m >= p.M) continue;
    const float* rb = p.rowbias ? p.rowbias + (size_t)(m / p.rows_per_batch) * p.ld_rowbias : nullptr;
    const float* gt = p.gate ? p.gate + (size_t)(m / p.rows_per_batch) * p.ld_gate : nullptr;
    const size_t crow = p.c_rpb ? (size_t)(m / p.c_rpb) * p.c_bstride + (size_t)(m % p.c_rpb) * p.ldc : (size_t)m * p.ldc;
    store_row<TN>(p, crow, n_wave, lq, [&](int tn, int n) {
      if (!p.geglu) return epi4(p, acc[tn][tm], m, n, rb, gt);
      f32x4 v = acc[tn][tm];   // GEGLU halves: fp8 scale and bias only (launch_gemm rejects the other operands)
      if (p.wscale) v *= *reinterpret_cast<const f32x4*>(p.wscale + n);
      if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
      return v;
    });
  }
}

// ---- epilogue operands fetched EARLY (gemm_pipe.hip, residual launches) ----------------------------------------------------------
// scripts/gemm_timeline.py on the K = 1280 residual projections (8192 x 1280 x 1280 + R, 192 launches per SDXL step): K loop 21 us,
// epilogue 12.7 us -- every CU leaves its K loop at the same moment and only THEN starts fetching its 80 KB of residual, 20 dependent
// load -> wait -> add -> store rounds per wave with nothing else on the chip to overlap them. The residual (and bias) values a lane
// needs are known from the start, so they are loaded into registers during the last 1.5 K iterations (after the loop's final LDS-DMA
// wait: no interaction with the counted vmcnt of the DMA ring) and the epilogue starts with its operands in hand.
// 8 consecutive channels per sub-tile pair = one 16-byte load; 40 (bf16 residual) + 20 (bias) registers per lane at TM x TN = 4 x 5.
template <int TM, int TN>
struct EpiPre {
  u32x4 r8[TM][TN / 2 > 0 ? TN / 2 : 1];   // residual channels of the pair (acc[2h] | acc[2h+1]) of row-tile tm
  u32x2 r4[TM];                            // the unpaired last sub-tile (TN odd)
  f32x4 b[TN];                             // bias of this lane's 4 channels of sub-tile tn
};

template <int TM, int TN>
__device__ __forceinline__ void epi_prefetch(const GemmArgs& p, EpiPre<TM, TN>& pre, int m_wave, int n_wave, int lane) {
  const int lq = lane >> 4;
  // bounds-checked buffer loads: rows >= M and channels >= N read as zero (the matching stores are skipped)
  const __amdgpu_buffer_rsrc_t r_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.R), 0, (unsigned)(((size_t)(p.M - 1) * p.ldr + p.N) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? (unsigned)((size_t)p.N * 4) : 0u, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n_wave + acc_col<TN>(tn, lq, false);
    pre.b[tn] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, n < p.N ? (unsigned)n * 4u : OOB, 0, 0));
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int m = m_wave + tm * 16 + (lane & 15);
    const unsigned row = (unsigned)((size_t)m * p.ldr * 2);
#pragma unroll
    for (int h = 0; h < TN / 2; ++h) {
      const int n = n_wave + h * 32 + lq * 8;
      pre.r8[tm][h] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, (m < p.M && n + 8 <= p.N) ? row + (unsigned)n * 2u : OOB, 0, 0));
    }
    if constexpr (TN & 1) {
      const int n = n_wave + (TN - 1) * 16 + lq * 4;
      pre.r4[tm] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r_rsrc, (m < p.M && n + 4 <= p.N) ? row + (unsigned)n * 2u : OOB, 0, 0));
    }
  }
}

// gemm_epilogue for a launch whose residual and bias sit in `pre` (bf16 residual, no GEGLU / fp8 scale / gate: gemm_pipe.hip checks)
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_pre(const GemmArgs& p, f32x4 (&acc)[TN][TM], const EpiPre<TM, TN>& pre, int m_wave,
                                                  int n_wave, int lane) {
  const int lq = lane >> 4;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int m = m_wave + tm * 16 + (lane & 15);
    if (m >= p.M) continue;
    const float* rb = p.rowbias ? p.rowbias + (size_t)(m / p.rows_per_batch) * p.ld_rowbias : nullptr;
    const size_t crow = p.c_rpb ? (size_t)(m / p.c_rpb) * p.c_bstride + (size_t)(m % p.c_rpb) * p.ldc : (size_t)m * p.ldc;
    store_row<TN>(p, crow, n_wave, lq, [&](int tn, int n) {
      f32x4 v = acc[tn][tm] + pre.b[tn];        // (no bias: the descriptor was empty, the loads returned zeros)
      if (rb) v += *reinterpret_cast<const f32x4*>(rb + n);
      bf16x4 r4;
      if ((TN & 1) && tn == TN - 1) {
        r4 = __builtin_bit_cast(bf16x4, pre.r4[tm]);
      } else {
        const u32x4 q = pre.r8[tm][tn >> 1];
        r4 = __builtin_bit_cast(bf16x4, (tn & 1) ? u32x2{q[2], q[3]} : u32x2{q[0], q[1]});
      }
      v[0] += (float)r4[0];
      v[1] += (float)r4[1];
      v[2] += (float)r4[2];
      v[3] += (float)r4[3];
      return act4(p, v * p.out_scale);
    });
  }
}

// Epilogue of the LayerNorm-folded projections (mi355x_sd_linear_ln): acc <- rstd[m] * acc - mean[m] * rstd[m] * wsum[n]
// + bias[n], then GEGLU / SiLU / tanh-GELU and the store. A separate function (and separate kernel instantiations,
// template parameter LN) so the register allocation of every other GEMM is unaffected. The row statistics of all TM
// row-tiles are fetched up front with clamped, branch-free addresses.
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_ln(const GemmArgs& p, f32x4 (&acc)[TN][TM], int m_wave, int n_wave,
                                                 int lane) {
  const int lq = lane >> 4;
  f32x2 rs[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
    rs[tm] = *reinterpret_cast<const f32x2*>(p.rowstat + 2 * (size_t)min(m_wave + tm * 16 + (lane & 15), p.M - 1));
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int m = m_wave + tm * 16 + (lane & 15);
    if (m >= p.M) continue;
    store_row<TN>(p, (size_t)m * p.ldc, n_wave, lq, [&](int tn, int n) {
      f32x4 v = rs[tm][0] * acc[tn][tm] + rs[tm][1] * *reinterpret_cast<const f32x4*>(p.wsum + n);
      if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
      return p.geglu ? v : act4(p, v);
    });
  }
}

// Epilogue of the W8A8 GEMM (mi355x_sd_linear_f8 / _f8_q): acc * ascale[m] * wscale[n] + bias, optional gate / residual /
// tanh-GELU / row-remapped C; bf16 or e4m3 store. Own function + own kernel instantiation (see gemm_epilogue_ln).
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_f8(const GemmArgs& p, f32x4 (&acc)[TN][TM], int m_wave, int n_wave,
                                                 int lane) {
  static_assert(TN % 2 == 0, "fp8 GEMM tiles pair their n sub-tiles");
  const int lq = lane >> 4;
  float as[TM], oinv[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int mc = min(m_wave + tm * 16 + (lane & 15), p.M - 1);
    as[tm] = p.ascale[mc];
    oinv[tm] = 0.f;
    if (p.out_f8) {   // safe row scale for the e4m3 output (1.1: the quantised operands may exceed their fp32 norms slightly)
      const float bound = 1.1f * (p.a_l2[mc] * p.w_norm_max + p.bias_abs_max);
      oinv[tm] = 448.0f / fmaxf(bound, 1e-12f);
      if (n_wave == 0 && lq == 0 && m_wave + tm * 16 + (lane & 15) < p.M) p.oscale[mc] = fmaxf(bound, 1e-12f) * (1.0f / 448.0f);
    }
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int m = m_wave + tm * 16 + (lane & 15);
    if (m >= p.M) continue;
    const float* gt = p.gate ? p.gate + (size_t)(m / p.rows_per_batch) * p.ld_gate : nullptr;
    const size_t crow = p.c_rpb ? (size_t)(m / p.c_rpb) * p.c_bstride + (size_t)(m % p.c_rpb) * p.ldc : (size_t)m * p.ldc;
    auto fin = [&](int tn, int n) {
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
      return act4(p, v);
    };
    auto q4 = [&](f32x4 v) {   // 4 x e4m3 with the row's output scale
      v *= oinv[tm];
      int q = 0;
      q = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], q, false);
      q = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], q, true);
      return (unsigned)q;
    };
#pragma unroll
    for (int h = 0; h < TN / 2; ++h) {   // one loop for both output types: two loops over the accumulators make the
      const int n = n_wave + h * 32 + lq * 8;   // register allocator split their live ranges and spill in the K loop
      if (n >= p.N) continue;
      const bool two = n + 4 < p.N;
      const f32x4 lo = fin(2 * h, n), hi = two ? fin(2 * h + 1, n + 4) : lo;
      if (p.out_f8) {   // 8 e4m3 bytes per lane (ldc % 8, no gate / residual: launch_gemm_f8)
        unsigned char* dst = reinterpret_cast<unsigned char*>(p.C) + crow + n;
        if (two) *reinterpret_cast<u32x2*>(dst) = u32x2{q4(lo), q4(hi)};
        else *reinterpret_cast<unsigned*>(dst) = q4(lo);
      } else if (two) {
        store8(p, crow + n, lo, hi);
      } else {
        store4(p, crow + n, lo);
      }
    }
  }
}

}  // namespace sd
