// HBM-bound normalisation kernels (gfx950): GroupNorm(+SiLU) over NHWC rows and LayerNorm over token rows.
//
// Reference semantics:
//   paddle.nn.GroupNorm(num_groups, C, epsilon) + F.silu   ppdiffusers/ppdiffusers/models/resnet.py:739-741, 789-795,
//                                                          transformer_2d.py:359 (eps 1e-6), unet_2d_condition.py:1194-1195
//   paddle.nn.LayerNorm(C, epsilon=1e-5)                   ppdiffusers/ppdiffusers/models/attention.py:397, 442, 463
// Statistics are biased (divide by N), accumulated in fp32 (+ fp64 across blocks), outputs written as bf16.
//
// GroupNorm is split in two so the second half is a pure per-channel affine (+SiLU):
//   1. gn_partial_kernel:  per-(batch, pixel-strip) block sums of x and x^2 per group      (reads x once)
//      gn_finalize_kernel: mean / rstd per (batch, group) -> scale[b][c] = gamma*rstd, shift[b][c] = beta - mean*scale
//   2. scale_shift_act_kernel: y = act(x*scale + shift)                                    (reads x once, writes y once)
// All loads/stores are 16-byte (8 x bf16) per lane.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace sd {

static int ln_grid(int rows, int rows_per_block);

// 8 consecutive channels of a row as fp32: from the build's 16-bit elements (one 16-byte load) or, in the fp32
// residual-stream mode (XF32), from fp32 rows (two 16-byte loads). `x` is the row base in ELEMENTS of its own type.
// `ok == false` yields zeros. Only the raw load sits under the predicate (the conversion does not), so that the compiler
// issues the loads of an unrolled caller back to back instead of one branch + s_waitcnt per chunk.
template <bool XF32>
__device__ __forceinline__ void load8f(const void* x, size_t elem_off, float (&f)[8], bool ok = true) {
  if constexpr (XF32) {
    const float* p = reinterpret_cast<const float*>(x) + elem_off;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
      a = *reinterpret_cast<const f32x4*>(p);
      b = *reinterpret_cast<const f32x4*>(p + 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[j] = a[j]; f[4 + j] = b[j]; }
  } else {
    u32x4 raw = {0u, 0u, 0u, 0u};
    if (ok) raw = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16*>(x) + elem_off);
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(&raw);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (float)v[j];
  }
}

// pixels per thread-slot per block: statistics pass / apply pass (measured inside the SDXL step, profiles/r02_l_norm_knobs.txt:
// statistics 8 / 16 / 32 -> 1.20 / 1.01 / 0.94 ms, apply 0.90 / 0.93 / 0.99 ms; the statistics pass stays at 16 because its
// block partition is part of the result's summation order)
constexpr int GN_ITERS_STATS = 16;
constexpr int GN_ITERS_APPLY = 8;
constexpr int GN_MAXC = 4096;

struct GnGeom {
  int cv;        // 16-B chunks per pixel = C / 8
  int ppp;       // pixels processed per pass by one block
  int threads;   // active threads = ppp * cv
  int block;     // launch block size (multiple of 64)
  int ppb;       // pixels per block
  int nblk;      // blocks per batch item
};
static GnGeom gn_geom(int HW, int C, int iters = GN_ITERS_STATS) {
  GnGeom g;
  g.cv = C / 8;
  g.ppp = g.cv <= 256 ? 256 / g.cv : 1;
  g.threads = g.ppp * g.cv;
  g.block = (g.threads + 63) / 64 * 64;
  g.ppb = g.ppp * iters;
  g.nblk = (HW + g.ppb - 1) / g.ppb;
  return g;
}

template <bool XF32>
__global__ void gn_partial_kernel(const void* __restrict__ x, int HW, int C, int ldx, int groups, int cv, int ppp,
                                  int nthreads, int ppb, float* __restrict__ partial) {
  // Deterministic (fixed-order) reduction: per-thread channel partials -> LDS -> per-channel -> per-group.
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  float* t_sum = gsm;                       // [ppp][C]
  float* t_sq = gsm + (size_t)ppp * C;      // [ppp][C]
  float* ch = t_sq + (size_t)ppp * C;       // [2][C]
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  if (tid < nthreads) {
    const int cc = tid % cv, pl = tid / cv;
    const int p_begin = blockIdx.x * ppb;
    const int p_end = min(p_begin + ppb, HW);
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
    const size_t xb = (size_t)b * HW * ldx + cc * 8;
    // four pixels' loads in flight per thread (one 16-B load per trip left the kernel latency-bound at ~1.7 TB/s); the
    // accumulation order is the pixel order either way, so the partials do not depend on the unrolling
    int pix = p_begin + pl;
    for (; pix + 3 * ppp < p_end; pix += 4 * ppp) {
      float v[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) load8f<XF32>(x, xb + (size_t)(pix + u * ppp) * ldx, v[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          s[j] += v[u][j];
          q[j] = __builtin_fmaf(v[u][j], v[u][j], q[j]);
        }
      }
    }
    for (; pix < p_end; pix += ppp) {
      float v[8];
      load8f<XF32>(x, xb + (size_t)pix * ldx, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        s[j] += v[j];
        q[j] = __builtin_fmaf(v[j], v[j], q[j]);
      }
    }
    float* ts = t_sum + (size_t)pl * C + cc * 8;
    float* tq = t_sq + (size_t)pl * C + cc * 8;
    *reinterpret_cast<f32x4*>(ts) = f32x4{s[0], s[1], s[2], s[3]};
    *reinterpret_cast<f32x4*>(ts + 4) = f32x4{s[4], s[5], s[6], s[7]};
    *reinterpret_cast<f32x4*>(tq) = f32x4{q[0], q[1], q[2], q[3]};
    *reinterpret_cast<f32x4*>(tq + 4) = f32x4{q[4], q[5], q[6], q[7]};
  }
  __syncthreads();
  for (int c = tid; c < C; c += blockDim.x) {
    float a = 0.f, d = 0.f;
    for (int pl = 0; pl < ppp; ++pl) {
      a += t_sum[(size_t)pl * C + c];
      d += t_sq[(size_t)pl * C + c];
    }
    ch[c] = a;
    ch[C + c] = d;
  }
  __syncthreads();
  const int cpg = C / groups;
  if (tid < groups) {
    float a = 0.f, d = 0.f;
    for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
      a += ch[c];
      d += ch[C + c];
    }
    float* pp = partial + ((size_t)b * gridDim.x + blockIdx.x) * 2 * groups;
    pp[2 * tid] = a;
    pp[2 * tid + 1] = d;
  }
}

__global__ void gn_finalize_kernel(const float* __restrict__ partial, int nblk, int HW, int C, int groups, float eps,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ scale_shift) {
  __shared__ float mean_s[64], rstd_s[64];
  __shared__ double part_s[64][8], part_q[64][8];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int cpg = C / groups;
  {
    // 8 threads per group, each a fixed strided subset of the block partials (deterministic order)
    const int g = tid >> 3, j = tid & 7;
    if (g < groups) {
      double s = 0.0, q = 0.0;
      const float* pp = partial + (size_t)b * nblk * 2 * groups + 2 * g;
      for (int i = j; i < nblk; i += 8) {
        s += (double)pp[(size_t)i * 2 * groups];
        q += (double)pp[(size_t)i * 2 * groups + 1];
      }
      part_s[g][j] = s;
      part_q[g][j] = q;
    }
  }
  __syncthreads();
  if (tid < groups) {
    double s = 0.0, q = 0.0;
    for (int j = 0; j < 8; ++j) {
      s += part_s[tid][j];
      q += part_q[tid][j];
    }
    const double n = (double)HW * cpg;
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    mean_s[tid] = (float)mean;
    rstd_s[tid] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  float* sc = scale_shift + (size_t)b * 2 * C;
  for (int c = tid; c < C; c += blockDim.x) {
    const int g = c / cpg;
    const float a = gamma[c] * rstd_s[g];
    sc[c] = a;
    sc[C + c] = beta[c] - mean_s[g] * a;
  }
}

int groupnorm_partial_floats(int B, int HW, int C) {
  if (C <= 0 || (C & 7)) return 0;
  const GnGeom g = gn_geom(HW, C);
  return B * g.nblk * 2 * 64;
}

int launch_groupnorm_stats(const void* x, int x_f32, int B, int HW, int C, int ldx, int groups, float eps, const float* gamma,
                           const float* beta, float* partial, float* scale_shift, hipStream_t stream) {
  if (B <= 0 || HW <= 0 || C <= 0) return SD_ERR_INVALID;
  if ((C & 7) || (ldx & 7) || C > GN_MAXC || groups <= 0 || groups > 64 || C % groups) return SD_ERR_UNSUPPORTED;
  const GnGeom g = gn_geom(HW, C);
  if (g.block > 1024) return SD_ERR_UNSUPPORTED;
  const size_t lds = (size_t)(2 * g.ppp + 2) * C * sizeof(float);
  if (x_f32)
    hipLaunchKernelGGL(gn_partial_kernel<true>, dim3(g.nblk, B), dim3(g.block), lds, stream, x, HW, C, ldx, groups, g.cv,
                       g.ppp, g.threads, g.ppb, partial);
  else
    hipLaunchKernelGGL(gn_partial_kernel<false>, dim3(g.nblk, B), dim3(g.block), lds, stream, x, HW, C, ldx, groups, g.cv,
                       g.ppp, g.threads, g.ppb, partial);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(512), 0, stream, partial, g.nblk, HW, C, groups, eps, gamma,
                     beta, scale_shift);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

// XF32: x rows are fp32 (residual-stream mode); raw16 (optional, XF32 only) receives the 16-bit rounding of the raw rows
// in the same pass -- the operand of the resnet's conv_shortcut GEMM (resnet.py:797-798), which must not read fp32.
template <bool SILU, bool XF32>
__global__ void scale_shift_act_kernel(const void* __restrict__ x, int HW, int C, int ldx, int cv, int ppp, int nthreads, int ppb,
                                       const float* __restrict__ scale_shift, bf16* __restrict__ y, int ldy,
                                       bf16* __restrict__ raw16, int ld_raw) {
  // Same block geometry as gn_partial_kernel: a thread owns ONE 8-channel chunk (its scale / shift stay in registers) and
  // walks pixels, four loads in flight. (The first version decomposed a flat 64-bit chunk id with two divisions per
  // 16 bytes and re-read scale / shift every trip: ~350 VALU instructions per chunk, VALU-bound at 1.4 TB/s.)
  const int tid = threadIdx.x;
  if (tid >= nthreads) return;
  const int b = blockIdx.y;
  const int cc = tid % cv, pl = tid / cv;
  const int p_begin = blockIdx.x * ppb;
  const int p_end = min(p_begin + ppb, HW);
  const float* sc = scale_shift + (size_t)b * 2 * C + cc * 8;
  const f32x4 a0 = *reinterpret_cast<const f32x4*>(sc), a1 = *reinterpret_cast<const f32x4*>(sc + 4);
  const f32x4 b0 = *reinterpret_cast<const f32x4*>(sc + C), b1 = *reinterpret_cast<const f32x4*>(sc + C + 4);
  const size_t row0 = (size_t)b * HW;
  auto emit = [&](size_t row, const float (&v)[8]) {
    if (XF32 && raw16) {
      u32x4 rk = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
      *reinterpret_cast<u32x4*>(raw16 + row * ld_raw + cc * 8) = rk;
    }
    float o[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = __builtin_fmaf(v[j], a0[j], b0[j]);
      o[4 + j] = __builtin_fmaf(v[4 + j], a1[j], b1[j]);
    }
    if (SILU) {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = silu_f(o[j]);
    }
    u32x4 pk = {pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7])};
    *reinterpret_cast<u32x4*>(y + row * ldy + cc * 8) = pk;
  };
  int pix = p_begin + pl;
  for (; pix + 3 * ppp < p_end; pix += 4 * ppp) {
    float v[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) load8f<XF32>(x, (row0 + pix + u * ppp) * ldx + cc * 8, v[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) emit(row0 + pix + u * ppp, v[u]);
  }
  for (; pix < p_end; pix += ppp) {
    float v[8];
    load8f<XF32>(x, (row0 + pix) * ldx + cc * 8, v);
    emit(row0 + pix, v);
  }
}

int launch_scale_shift_act(const void* x, int x_f32, int B, int HW, int C, int ldx, const float* scale_shift, int silu, bf16* y,
                           int ldy, bf16* raw16, int ld_raw, hipStream_t stream) {
  if (B <= 0 || HW <= 0 || C <= 0) return SD_ERR_INVALID;
  if ((C & 7) || (ldx & 7) || (ldy & 7)) return SD_ERR_UNSUPPORTED;
  if (raw16 && (!x_f32 || (ld_raw & 7))) return SD_ERR_UNSUPPORTED;
  const GnGeom g = gn_geom(HW, C, GN_ITERS_APPLY);
  if (g.block > 1024 || B > 65535) return SD_ERR_UNSUPPORTED;
#define SD_SSA(S_, F_) \
  hipLaunchKernelGGL((scale_shift_act_kernel<S_, F_>), dim3(g.nblk, B), dim3(g.block), 0, stream, x, HW, C, ldx, g.cv, g.ppp, \
                     g.threads, g.ppb, scale_shift, y, ldy, raw16, ld_raw)
  if (silu) { if (x_f32) SD_SSA(true, true); else SD_SSA(true, false); }
  else      { if (x_f32) SD_SSA(false, true); else SD_SSA(false, false); }
#undef SD_SSA
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

// ---- GroupNorm (+SiLU) in ONE launch for small groups: the (batch, group) chunk held in registers --------------------------------
// A block of 1024 threads owns one (batch item, group): HW pixels x cpg channels, at most 24 dwords (48 elements) per thread =
// 96 KB per chunk. One read: every thread loads its dwords (2 channels each; cpg = C / groups is even for every SD width), sums
// x and x^2 in fp32; block reduction in a fixed order (deterministic), mean / rstd in double like the two-launch form; then the
// affine (+SiLU) is applied to the values still in registers and written. Three launches (statistics, finalize, apply) and the
// second read of x become one launch -- what the batch-1 SD-1.5 step needs (61 GroupNorms of 0.1-5 MB each: pure launch latency,
// 19 % of its 6.3 ms step), and the 32x32 levels of SDXL. Larger groups (HW * cpg * 2 > 96 KB) keep the split form.
constexpr int GNF_THREADS = 1024, GNF_MAXD = 24;
int groupnorm_act_fits(int HW, int C, int groups) {
  static const bool off = sd_switch("MI355X_SD_NO_GN_FUSED") != nullptr;   // A/B switch (read by the program builders too)
  if (off || groups <= 0 || C <= 0 || (C % groups) || HW <= 0) return 0;
  const int cpg = C / groups;
  if ((cpg & 1) || (C & 7) || cpg > 128) return 0;   // (128: the kernel's LDS table of a group's gamma / beta)
  return (long)HW * (cpg / 2) <= (long)GNF_THREADS * GNF_MAXD;
}

template <bool SILU>
__global__ __launch_bounds__(GNF_THREADS) void gn_fused_kernel(const bf16* __restrict__ x, int HW, int C, int ldx, int groups, float eps,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             bf16* __restrict__ y, int ldy) {
  __shared__ float red_s[GNF_THREADS / 64], red_q[GNF_THREADS / 64];
  __shared__ float stat[2];
  __shared__ float gb[2][128];   // this group's gamma / beta (cpg <= 128)
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int cpg = C / groups, dpp = cpg >> 1;   // dwords per pixel of this group
  const int nd = HW * dpp;                      // dwords of the chunk
  const bf16* xb = x + (size_t)b * HW * ldx + g * cpg;
  bf16* yb = y + (size_t)b * HW * ldy + g * cpg;
  if (tid < cpg) {
    gb[0][tid] = gamma[g * cpg + tid];
    gb[1][tid] = beta[g * cpg + tid];
  }
  unsigned v[GNF_MAXD];
  float s = 0.f, q = 0.f;
  // i / dpp == umulhi(i, ceil(2^32 / dpp)) for the i <= 24575 of a chunk; dpp == 1 (two channels per group) would need 2^32 itself
  const unsigned inv_dpp = dpp > 1 ? 0xFFFFFFFFu / (unsigned)dpp + 1u : 0u;
#pragma unroll
  for (int k = 0; k < GNF_MAXD; ++k) {
    const int i = tid + k * GNF_THREADS;
    v[k] = 0u;
    if (i < nd) {
      const int pix = dpp > 1 ? (int)__umulhi((unsigned)i, inv_dpp) : i, d = i - pix * dpp;
      v[k] = *reinterpret_cast<const unsigned*>(xb + (size_t)pix * ldx + 2 * d);
    }
  }
#pragma unroll
  for (int k = 0; k < GNF_MAXD; ++k) {   // (zeros past the chunk's end add nothing)
    const bf16x2 e = __builtin_bit_cast(bf16x2, v[k]);
    const float a = (float)e[0], c = (float)e[1];
    s += a + c;
    q = __builtin_fmaf(a, a, __builtin_fmaf(c, c, q));
  }
  s = wave_sum(s);
  q = wave_sum(q);
  if ((tid & 63) == 0) {
    red_s[tid >> 6] = s;
    red_q[tid >> 6] = q;
  }
  __syncthreads();
  if (tid == 0) {
    double sa = 0.0, sq = 0.0;
    for (int w = 0; w < GNF_THREADS / 64; ++w) {   // fixed order
      sa += (double)red_s[w];
      sq += (double)red_q[w];
    }
    const double n = (double)HW * cpg;
    const double mean = sa / n;
    double var = sq / n - mean * mean;
    if (var < 0.0) var = 0.0;
    stat[0] = (float)mean;
    stat[1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const float mean = stat[0], rstd = stat[1];
#pragma unroll
  for (int k = 0; k < GNF_MAXD; ++k) {
    const int i = tid + k * GNF_THREADS;
    if (i < nd) {
      const int pix = dpp > 1 ? (int)__umulhi((unsigned)i, inv_dpp) : i, d = i - pix * dpp;
      const bf16x2 e = __builtin_bit_cast(bf16x2, v[k]);
      // the same arithmetic as the split form: scale = gamma * rstd, shift = beta - mean * scale, y = x * scale + shift
      const float sc0 = gb[0][2 * d] * rstd, sc1 = gb[0][2 * d + 1] * rstd;
      float o0 = __builtin_fmaf((float)e[0], sc0, gb[1][2 * d] - mean * sc0), o1 = __builtin_fmaf((float)e[1], sc1, gb[1][2 * d + 1] - mean * sc1);
      if (SILU) {
        o0 = silu_f(o0);
        o1 = silu_f(o1);
      }
      *reinterpret_cast<unsigned*>(yb + (size_t)pix * ldy + 2 * d) = pack_bf16(o0, o1);
    }
  }
}

int launch_groupnorm_act(const bf16* x, int B, int HW, int C, int ldx, int groups, float eps, const float* gamma, const float* beta,
                         int silu, bf16* y, int ldy, hipStream_t stream) {
  if (B <= 0 || HW <= 0 || C <= 0) return SD_ERR_INVALID;
  if (!groupnorm_act_fits(HW, C, groups) || (ldx & 1) || (ldy & 1)) return SD_ERR_UNSUPPORTED;
  // (batch items on grid.y, 65535 per launch: larger batches go out in slices -- the predicate above is the whole contract)
  for (int b0 = 0; b0 < B; b0 += 65535) {
    const int nb = B - b0 < 65535 ? B - b0 : 65535;
    const bf16* xb = x + (size_t)b0 * HW * ldx;
    bf16* yb = y + (size_t)b0 * HW * ldy;
    if (silu) hipLaunchKernelGGL(gn_fused_kernel<true>, dim3(groups, nb), dim3(GNF_THREADS), 0, stream, xb, HW, C, ldx, groups, eps, gamma, beta, yb, ldy);
    else hipLaunchKernelGGL(gn_fused_kernel<false>, dim3(groups, nb), dim3(GNF_THREADS), 0, stream, xb, HW, C, ldx, groups, eps, gamma, beta, yb, ldy);
  }
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

// LayerNorm: persistent waves. A wave keeps gamma / beta of its channel chunks in registers and walks rows
// (ROWS at a time: all loads issued up front, the ROWS reduction chains interleave), so the per-row traffic is
// exactly one read and one write of the row. Statistics in one pass over data shifted by the row's first element
// (sum(x-K), sum((x-K)^2): no cancellation for the O(1..10) activations here), fp32.
template <int NCH, int ROWS, bool XF32 = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const void* __restrict__ x, int rows, int C, int ldx,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, bf16* __restrict__ y, int ldy) {
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (blockDim.x >> 6);
  const int cv = C >> 3;
  float g[NCH][8], bt[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int cc = lane + 64 * i;
    // two 16-byte loads per vector under ONE predicate each (element-wise selects compile to a branch + dword load per element)
    f32x4 ga = {1.f, 1.f, 1.f, 1.f}, gb = ga, ba = {0.f, 0.f, 0.f, 0.f}, bb = ba;
    if (gamma && cc < cv) {
      ga = *reinterpret_cast<const f32x4*>(gamma + cc * 8);
      gb = *reinterpret_cast<const f32x4*>(gamma + cc * 8 + 4);
    }
    if (beta && cc < cv) {
      ba = *reinterpret_cast<const f32x4*>(beta + cc * 8);
      bb = *reinterpret_cast<const f32x4*>(beta + cc * 8 + 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      g[i][j] = ga[j];
      g[i][4 + j] = gb[j];
      bt[i][j] = ba[j];
      bt[i][4 + j] = bb[j];
    }
  }
  const float invC = 1.0f / (float)C;
  for (int row0 = wave_g * ROWS; row0 < rows; row0 += nwaves * ROWS) {
    float v[ROWS][NCH][8];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int row = min(row0 + r, rows - 1);
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int cc = lane + 64 * i;
        load8f<XF32>(x, (size_t)row * ldx + cc * 8, v[r][i], cc < cv);
      }
    }
    float mean[ROWS], rstd[ROWS], s1[ROWS], s2[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const float K = __shfl(v[r][0][0], 0, 64);
      float a = 0.f, q = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        if (lane + 64 * i < cv) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float d = v[r][i][j] - K;
            a += d;
            q = __builtin_fmaf(d, d, q);
          }
        }
      }
      s1[r] = a;
      s2[r] = q;
      mean[r] = K;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        s1[r] += __shfl_xor(s1[r], o, 64);
        s2[r] += __shfl_xor(s2[r], o, 64);
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const float m = s1[r] * invC;
      const float var = fmaxf(s2[r] * invC - m * m, 0.f);
      mean[r] += m;
      rstd[r] = rsqrtf(var + eps);
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      if (row0 + r < rows) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const int cc = lane + 64 * i;
          if (cc < cv) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = __builtin_fmaf((v[r][i][j] - mean[r]) * rstd[r], g[i][j], bt[i][j]);
            u32x4 pk = {pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7])};
            *reinterpret_cast<u32x4*>(y + (size_t)(row0 + r) * ldy + cc * 8) = pk;
          }
        }
      }
    }
  }
}

// Grid of the row-normalisation kernels: blocks = clamp(needed / div, min, max). Round 1 measured, back to back in isolation,
// that waves walking 2-4 row groups beat one group per wave (scripts/ln_probe.py: 8192 x 1280 15.7 -> 13.6 us at 256 blocks) and
// used div = 2; inside the SDXL step, and with the per-wave prologue now 12 vector loads instead of 48 branches, one trip per
// wave wins (LayerNorm class 3.07 -> 2.79 ms per step, profiles/r02_k_ln_grid.txt): div = 1, at most 1024 blocks.
static int ln_grid(int rows, int rows_per_block) {
  constexpr int lo = 256, hi = 1024;   // (div = 1)
  const int needed = (rows + rows_per_block - 1) / rows_per_block;
  int blocks = needed;
  if (blocks < lo) blocks = lo;
  if (blocks > hi) blocks = hi;
  return blocks < needed ? blocks : needed;
}

int launch_layernorm(const void* x, int x_f32, int rows, int C, int ldx, const float* gamma, const float* beta, float eps, bf16* y,
                     int ldy, hipStream_t stream) {
  if (rows <= 0 || C <= 0) return SD_ERR_INVALID;
  if ((C & 7) || (ldx & 7) || (ldy & 7) || C > 2560) return SD_ERR_UNSUPPORTED;
  const int wpb = 4;
  const int cv = C >> 3;
  // rows per wave: round 1 measured 4 better than 2 (4.6 vs 5.2 ms per SDXL step) when every wave paid a 48-branch prologue; with
  // the vector prologue and one trip per wave, 2 rows (twice the waves, ~110 instead of 200 registers) is ahead again: class
  // 2.80 -> 2.65 ms per step (profiles/r02_l_norm_knobs.txt).
  const int blocks = ln_grid(rows, wpb * 2);
#define SD_LN_LAUNCH(NCH, R_, F_) \
  hipLaunchKernelGGL((layernorm_kernel<NCH, R_, F_>), dim3(blocks), dim3(64 * wpb), 0, stream, x, rows, C, ldx, gamma, beta, eps, y, ldy)
  if (x_f32) {
    if (cv <= 128) SD_LN_LAUNCH(2, 2, true);
    else if (cv <= 192) SD_LN_LAUNCH(3, 2, true);
    else SD_LN_LAUNCH(5, 2, true);
  } else {
    if (cv <= 128) SD_LN_LAUNCH(2, 2, false);
    else if (cv <= 192) SD_LN_LAUNCH(3, 2, false);
    else SD_LN_LAUNCH(5, 2, false);
  }
#undef SD_LN_LAUNCH
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

// Row statistics for the LayerNorm-folded GEMM (mi355x_sd_linear_ln): stats[row] = (rstd, -mean * rstd), the same
// shifted one-pass fp32 statistics as layernorm_kernel. The normalised row is never written: the consumer GEMM
// multiplies the raw rows by W.diag(gamma) and applies  rstd * acc - mean * rstd * rowsum(W')  in its epilogue.
template <int NCH, int ROWS>
__global__ __launch_bounds__(256) void row_stats_kernel(const bf16* __restrict__ x, int rows, int C, int ldx, float eps,
                                                        float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (blockDim.x >> 6);
  const int cv = C >> 3;
  const float invC = 1.0f / (float)C;
  for (int row0 = wave_g * ROWS; row0 < rows; row0 += nwaves * ROWS) {
    float s1[ROWS], s2[ROWS], K[ROWS];
    u32x4 raw[ROWS][NCH];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const bf16* xr = x + (size_t)min(row0 + r, rows - 1) * ldx;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int cc = lane + 64 * i;
        raw[r][i] = u32x4{0u, 0u, 0u, 0u};
        if (cc < cv) raw[r][i] = *reinterpret_cast<const u32x4*>(xr + cc * 8);
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const bf16x8 t0 = *reinterpret_cast<const bf16x8*>(&raw[r][0]);
      K[r] = __shfl((float)t0[0], 0, 64);
      float a = 0.f, q = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        if (lane + 64 * i < cv) {
          const bf16x8 t = *reinterpret_cast<const bf16x8*>(&raw[r][i]);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float d = (float)t[j] - K[r];
            a += d;
            q = __builtin_fmaf(d, d, q);
          }
        }
      }
      s1[r] = a;
      s2[r] = q;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        s1[r] += __shfl_xor(s1[r], o, 64);
        s2[r] += __shfl_xor(s2[r], o, 64);
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        if (row0 + r < rows) {
          const float m = s1[r] * invC;
          const float rstd = rsqrtf(fmaxf(s2[r] * invC - m * m, 0.f) + eps);
          *reinterpret_cast<float2*>(stats + 2 * (size_t)(row0 + r)) = make_float2(rstd, -(K[r] + m) * rstd);
        }
      }
    }
  }
}

int launch_row_stats(const bf16* x, int rows, int C, int ldx, float eps, float* stats, hipStream_t stream) {
  if (rows <= 0 || C <= 0) return SD_ERR_INVALID;
  if ((C & 7) || (ldx & 7) || C > 2560) return SD_ERR_UNSUPPORTED;
  const int wpb = 4, cv = C >> 3;
  constexpr int R = 4;
  const int blocks = ln_grid(rows, wpb * R);
#define SD_RS_LAUNCH(NCH) \
  hipLaunchKernelGGL((row_stats_kernel<NCH, R>), dim3(blocks), dim3(64 * wpb), 0, stream, x, rows, C, ldx, eps, stats)
  if (cv <= 64) SD_RS_LAUNCH(1);
  else if (cv <= 128) SD_RS_LAUNCH(2);
  else if (cv <= 192) SD_RS_LAUNCH(3);
  else SD_RS_LAUNCH(5);
#undef SD_RS_LAUNCH
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

// T5LayerNorm (RMSNorm; PPD/transformers/t5/modeling.py:86-108): y = weight * x * rsqrt(mean(x^2) + eps), no mean
// subtraction, no bias; statistics in fp32. Same wave layout as layernorm_kernel (a wave owns ROWS rows).
template <int NCH, int ROWS>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16* __restrict__ x, int rows, int C, int ldx,
                                                      const float* __restrict__ weight, float eps, bf16* __restrict__ y,
                                                      int ldy) {
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (blockDim.x >> 6);
  const int cv = C >> 3;
  const float invC = 1.0f / (float)C;
  for (int row0 = wave_g * ROWS; row0 < rows; row0 += nwaves * ROWS) {
    float v[ROWS][NCH][8], ss[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const bf16* xr = x + (size_t)min(row0 + r, rows - 1) * ldx;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int cc = lane + 64 * i;
        u32x4 raw = {0u, 0u, 0u, 0u};
        if (cc < cv) raw = *reinterpret_cast<const u32x4*>(xr + cc * 8);
        const bf16x8 t = *reinterpret_cast<const bf16x8*>(&raw);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          v[r][i][j] = (float)t[j];
          q = __builtin_fmaf(v[r][i][j], v[r][i][j], q);
        }
      }
      ss[r] = q;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int r = 0; r < ROWS; ++r) ss[r] += __shfl_xor(ss[r], o, 64);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      if (row0 + r >= rows) continue;
      const float rstd = rsqrtf(ss[r] * invC + eps);
      bf16* yr = y + (size_t)(row0 + r) * ldy;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int cc = lane + 64 * i;
        if (cc < cv) {
          float o8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) o8[j] = v[r][i][j] * rstd * weight[cc * 8 + j];
          u32x4 pk = {pack_bf16(o8[0], o8[1]), pack_bf16(o8[2], o8[3]), pack_bf16(o8[4], o8[5]), pack_bf16(o8[6], o8[7])};
          *reinterpret_cast<u32x4*>(yr + cc * 8) = pk;
        }
      }
    }
  }
}

int launch_rmsnorm(const bf16* x, int rows, int C, int ldx, const float* weight, float eps, bf16* y, int ldy,
                   hipStream_t stream) {
  if (rows <= 0 || C <= 0) return SD_ERR_INVALID;
  if ((C & 7) || (ldx & 7) || (ldy & 7) || C > 4096) return SD_ERR_UNSUPPORTED;
  const int cv = C >> 3;
  const int blocks = ln_grid(rows, 8);
#define SD_RMS_LAUNCH(NCH) \
  hipLaunchKernelGGL((rmsnorm_kernel<NCH, 2>), dim3(blocks), dim3(256), 0, stream, x, rows, C, ldx, weight, eps, y, ldy)
  if (cv <= 128) SD_RMS_LAUNCH(2);
  else if (cv <= 256) SD_RMS_LAUNCH(4);
  else SD_RMS_LAUNCH(8);
#undef SD_RMS_LAUNCH
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

// Adaptive LayerNorm of the MMDiT blocks: y = LN(x) * (1 + scale[b]) + shift[b], LN without affine, b = row / rows_per_batch.
// Reference: AdaLayerNormZero.forward (ppdiffusers/ppdiffusers/models/normalization.py:72-86), AdaLayerNormContinuous
// (:190-202), the modulated norm2 of JointTransformerBlock (attention.py:184-185) and the fused Triton op
// adaptive_layer_norm (paddlemix/triton_ops/triton_ops.py:981-1027).  Same wave layout as layernorm_kernel.
template <int NCH, int ROWS, bool M16 = false>
__global__ __launch_bounds__(256) void adaln_kernel(const bf16* __restrict__ x, int rows, int C, int ldx,
                                                    const void* __restrict__ scale, const void* __restrict__ shift,
                                                    int ld_mod, int rows_per_batch, float eps, bf16* __restrict__ y,
                                                    int ldy) {
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (blockDim.x >> 6);
  const int cv = C >> 3;
  const float invC = 1.0f / (float)C;
  for (int row0 = wave_g * ROWS; row0 < rows; row0 += nwaves * ROWS) {
    float v[ROWS][NCH][8];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int row = min(row0 + r, rows - 1);
      const bf16* xr = x + (size_t)row * ldx;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int cc = lane + 64 * i;
        u32x4 raw = {0u, 0u, 0u, 0u};
        if (cc < cv) raw = *reinterpret_cast<const u32x4*>(xr + cc * 8);
        const bf16x8 t = *reinterpret_cast<const bf16x8*>(&raw);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[r][i][j] = (float)t[j];
      }
    }
    float mean[ROWS], rstd[ROWS], s1[ROWS], s2[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const float K = __shfl(v[r][0][0], 0, 64);
      float a = 0.f, q = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        if (lane + 64 * i < cv) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float d = v[r][i][j] - K;
            a += d;
            q = __builtin_fmaf(d, d, q);
          }
        }
      }
      s1[r] = a;
      s2[r] = q;
      mean[r] = K;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        s1[r] += __shfl_xor(s1[r], o, 64);
        s2[r] += __shfl_xor(s2[r], o, 64);
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const float m = s1[r] * invC;
      const float var = fmaxf(s2[r] * invC - m * m, 0.f);
      mean[r] += m;
      rstd[r] = rsqrtf(var + eps);
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      if (row0 + r < rows) {
        const size_t mrow = (size_t)((row0 + r) / rows_per_batch) * ld_mod;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const int cc = lane + 64 * i;
          if (cc < cv) {
            float sc[8], sh[8], o[8];
            load_mod8<M16>(scale, mrow + cc * 8, sc);
            load_mod8<M16>(shift, mrow + cc * 8, sh);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = __builtin_fmaf((v[r][i][j] - mean[r]) * rstd[r], 1.0f + sc[j], sh[j]);
            u32x4 pk = {pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7])};
            *reinterpret_cast<u32x4*>(y + (size_t)(row0 + r) * ldy + cc * 8) = pk;
          }
        }
      }
    }
  }
}

// ---- W8A8 producers: rows quantised to OCP e4m3 with one fp32 scale per row (absmax / 448) ----
__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);   // bytes 0, 1
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);    // bytes 2, 3
  return (unsigned)w;
}

// AdaLayerNorm (same arithmetic as adaln_kernel) with the fp8 quantisation fused: the modulated row never exists in
// bf16. y8[row][C] bytes (row stride ldy bytes), yscale[row] = absmax / 448 (dequantised value = scale * q).
template <int NCH>
__global__ __launch_bounds__(256) void adaln_f8_kernel(const bf16* __restrict__ x, int rows, int C, int ldx,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       int ld_mod, int rows_per_batch, float eps,
                                                       unsigned char* __restrict__ y8, int ldy, float* __restrict__ yscale,
                                                       float* __restrict__ yl2) {
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (blockDim.x >> 6);
  const int cv = C >> 3;
  const float invC = 1.0f / (float)C;
  for (int row = wave_g; row < rows; row += nwaves) {
    float v[NCH][8];
    const bf16* xr = x + (size_t)row * ldx;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int cc = lane + 64 * i;
      u32x4 raw = {0u, 0u, 0u, 0u};
      if (cc < cv) raw = *reinterpret_cast<const u32x4*>(xr + cc * 8);
      const bf16x8 t = *reinterpret_cast<const bf16x8*>(&raw);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = (float)t[j];
    }
    const float K = __shfl(v[0][0], 0, 64);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (lane + 64 * i < cv) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = v[i][j] - K;
          s1 += d;
          s2 = __builtin_fmaf(d, d, s2);
        }
      }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const float m = s1 * invC;
    const float mean = K + m, rstd = rsqrtf(fmaxf(s2 * invC - m * m, 0.f) + eps);
    const int bidx = row / rows_per_batch;
    const float* sc = scale + (size_t)bidx * ld_mod;
    const float* sh = shift + (size_t)bidx * ld_mod;
    float amax = 0.f, sq = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int cc = lane + 64 * i;
      if (cc < cv) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          v[i][j] = __builtin_fmaf((v[i][j] - mean) * rstd, 1.0f + sc[cc * 8 + j], sh[cc * 8 + j]);
          amax = fmaxf(amax, fabsf(v[i][j]));
          sq = __builtin_fmaf(v[i][j], v[i][j], sq);
        }
      }
    }
    amax = wave_max(amax);
    sq = wave_sum(sq);
    const float qs = fmaxf(amax, 1e-12f) * (1.0f / 448.0f), inv = 1.0f / qs;
    if (lane == 0) {
      yscale[row] = qs;
      if (yl2) yl2[row] = sqrtf(sq);   // row L2 norm: bounds every output of the consuming GEMM (Cauchy-Schwarz)
    }
    unsigned char* yr = y8 + (size_t)row * ldy;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int cc = lane + 64 * i;
      if (cc < cv) {
        u32x2 pk = {pack_fp8x4(v[i][0] * inv, v[i][1] * inv, v[i][2] * inv, v[i][3] * inv),
                    pack_fp8x4(v[i][4] * inv, v[i][5] * inv, v[i][6] * inv, v[i][7] * inv)};
        *reinterpret_cast<u32x2*>(yr + cc * 8) = pk;
      }
    }
  }
}

int launch_adaln_f8(const bf16* x, int rows, int C, int ldx, const float* scale, const float* shift, int ld_mod,
                    int rows_per_batch, float eps, unsigned char* y8, int ldy, float* yscale, float* yl2, hipStream_t stream) {
  if (rows <= 0 || C <= 0 || rows_per_batch <= 0) return SD_ERR_INVALID;
  if ((C & 7) || (ldx & 7) || (ldy & 7) || (ld_mod & 3) || C > 2560) return SD_ERR_UNSUPPORTED;
  const int cv = C >> 3;
  int blocks = (rows + 3) / 4;
  if (blocks > 2048) blocks = 2048;
#define SD_AF8(NCH) \
  hipLaunchKernelGGL((adaln_f8_kernel<NCH>), dim3(blocks), dim3(256), 0, stream, x, rows, C, ldx, scale, shift, ld_mod, \
                     rows_per_batch, eps, y8, ldy, yscale, yl2)
  if (cv <= 128) SD_AF8(2);
  else if (cv <= 192) SD_AF8(3);
  else SD_AF8(5);
#undef SD_AF8
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

// bf16 rows -> e4m3 rows + per-row scale (the attention output in front of the MMDiT output projections). One wave per
// row, two passes (absmax, convert); the second pass re-reads the row from L2.
__global__ __launch_bounds__(256) void quantize_rows_kernel(const bf16* __restrict__ x, long rows, int C, int ldx, int x_rpb,
                                                            long x_bstride, unsigned char* __restrict__ y8, int ldy,
                                                            float* __restrict__ yscale) {
  const int lane = threadIdx.x & 63;
  const long wave_g = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
  const int cv = C >> 3;
  for (long row = wave_g; row < rows; row += nwaves) {
    // source row remap (rows of one stream inside the joint [B, S_img + S_txt, C] attention output); output is compact
    const bf16* xr = x_rpb ? x + (size_t)(row / x_rpb) * x_bstride + (size_t)(row % x_rpb) * ldx : x + (size_t)row * ldx;
    float amax = 0.f;
    for (int cc = lane; cc < cv; cc += 64) {
      const u32x4 raw = *reinterpret_cast<const u32x4*>(xr + cc * 8);
      const bf16x8 t = *reinterpret_cast<const bf16x8*>(&raw);
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf((float)t[j]));
    }
    amax = wave_max(amax);
    const float qs = fmaxf(amax, 1e-12f) * (1.0f / 448.0f), inv = 1.0f / qs;
    if (lane == 0) yscale[row] = qs;
    unsigned char* yr = y8 + (size_t)row * ldy;
    for (int cc = lane; cc < cv; cc += 64) {
      const u32x4 raw = *reinterpret_cast<const u32x4*>(xr + cc * 8);
      const bf16x8 t = *reinterpret_cast<const bf16x8*>(&raw);
      u32x2 pk = {pack_fp8x4((float)t[0] * inv, (float)t[1] * inv, (float)t[2] * inv, (float)t[3] * inv),
                  pack_fp8x4((float)t[4] * inv, (float)t[5] * inv, (float)t[6] * inv, (float)t[7] * inv)};
      *reinterpret_cast<u32x2*>(yr + cc * 8) = pk;
    }
  }
}

int launch_quantize_rows(const bf16* x, long rows, int C, int ldx, int x_rpb, long x_bstride, unsigned char* y8, int ldy,
                         float* yscale, hipStream_t stream) {
  if (rows <= 0 || C <= 0 || x_rpb < 0) return SD_ERR_INVALID;
  if ((C & 7) || (ldx & 7) || (ldy & 7) || (x_bstride & 7)) return SD_ERR_UNSUPPORTED;
  long blocks = (rows + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(quantize_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, rows, C, ldx, x_rpb, x_bstride, y8,
                     ldy, yscale);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

int launch_adaln(const bf16* x, int rows, int C, int ldx, const void* scale, const void* shift, int ld_mod, int mod16,
                 int rows_per_batch, float eps, bf16* y, int ldy, hipStream_t stream) {
  if (rows <= 0 || C <= 0 || rows_per_batch <= 0) return SD_ERR_INVALID;
  if ((C & 7) || (ldx & 7) || (ldy & 7) || (ld_mod & (mod16 ? 7 : 3)) || C > 2560) return SD_ERR_UNSUPPORTED;
  const int wpb = 4;
  const int cv = C >> 3;
  constexpr int R = 4;
  // all row groups resident (measured better here than the capped grid of ln_grid: 2.54 vs 2.77 ms per SD3 step --
  // the per-batch scale / shift vectors add a dependent load to every iteration)
  int blocks = (rows + wpb * R - 1) / (wpb * R);
  if (blocks > 256 * 8) blocks = 256 * 8;
#define SD_ADALN_LAUNCH(NCH, RR, M16)                                                                                         \
  hipLaunchKernelGGL((adaln_kernel<NCH, RR, M16>), dim3(blocks), dim3(64 * wpb), 0, stream, x, rows, C, ldx, scale, shift, ld_mod, \
                     rows_per_batch, eps, y, ldy)
  if (mod16) {   // 16-bit modulation vectors (the reference's own operand type)
    if (cv <= 128) SD_ADALN_LAUNCH(2, R, true);
    else if (cv <= 192) SD_ADALN_LAUNCH(3, R, true);
    else SD_ADALN_LAUNCH(5, 2, true);
  } else {
    if (cv <= 128) SD_ADALN_LAUNCH(2, R, false);
    else if (cv <= 192) SD_ADALN_LAUNCH(3, R, false);
    else SD_ADALN_LAUNCH(5, 2, false);
  }
#undef SD_ADALN_LAUNCH
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

}  // namespace sd
