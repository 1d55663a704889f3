// bf16 MFMA GEMM / implicit-GEMM convolution for the SD denoising hot path (gfx950).
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T )          fp32 accumulation in the matrix cores
//
// Replaces, on the reference side, every F.linear / F.conv2d reached from
//   LoRACompatibleLinear.forward  ppdiffusers/ppdiffusers/models/lora.py:453-459
//   LoRACompatibleConv.forward    ppdiffusers/ppdiffusers/models/lora.py:364-377
// plus the element-wise tails the reference runs as separate ops (bias, time-embedding broadcast add
// resnet.py:772-784, residual add + /output_scale_factor resnet.py:800-806, GEGLU gate activations.py:101-104).
//
// Data layout: activations are NHWC / token-major rows with an explicit row stride (lda / ldc), so a
// channel-concat is just two producers writing into one buffer. Weights are [N][K] with K contiguous
// (Paddle's Linear [in,out] is transposed once at load; conv OIHW is repacked to [O][kh][kw][I]).
//
// Kernel shape: 128x128x64 block tile, 256 threads = 4 waves in 2x2, each wave owns a 64x64 sub-tile as
// 4x4 v_mfma_f32_16x16x32_bf16 tiles. Operands are staged global -> registers -> LDS (double buffered,
// XOR-swizzled 16-B chunks so ds_read_b128 fragment reads are bank-conflict free), the next tile's global
// loads are in flight while the current tile is multiplied. The MFMA is issued "swapped" (W as the row
// operand) so that every lane ends up with 4 consecutive output channels of one row -> 8-byte stores.
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "kernels.h"

namespace sd {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int GEMM_THREADS = 256;

// 16 zero bytes: the source of every out-of-range chunk of the LDS-DMA loader (M/N/K tails, conv zero padding)
__device__ __attribute__((aligned(16))) const unsigned g_zero16[4] = {0u, 0u, 0u, 0u};

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// GLDS = true : operands go HBM -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave instruction, no staging
//               VGPRs, no ds_write pass); the XOR swizzle is applied on the per-lane SOURCE address because the
//               LDS destination of an LDS-DMA is lane-linear.
// GLDS = false: register-staged loader (kept as the A/B baseline; MI355X_SD_GEMM=regs selects it).
template <bool CONV, bool GLDS>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_bf16_kernel(const GemmArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (BM + BN) * BK * 2];
  unsigned char* As = smem;                       // [2][BM][BK] bf16, swizzled
  unsigned char* Ws = smem + 2 * BM * BK * 2;     // [2][BN][BK]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int ntn = (p.N + BN - 1) / BN;
  const int ntm = (p.M + BM - 1) / BM;
  const int lid = xcd_remap(blockIdx.x, ntm * ntn);
  const int tile_m = lid / ntn, tile_n = lid - tile_m * ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- loader geometry: thread owns 16-B chunk kc of rows r0 + 32*i ----
  const int kc = tid & 7;
  const int r0 = tid >> 3;  // 0..31
  const int swz = (kc ^ (r0 & 7)) << 4;

  const bf16* a_base[4];
  bool a_ok[4];
  int oy[4], ox[4];
  const bf16* w_base[4];
  bool w_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + r0 + 32 * i;
    a_ok[i] = m < p.M;
    if (CONV) {
      const int hw = p.Ho * p.Wo;
      const int mm = a_ok[i] ? m : 0;
      const int b = mm / hw;
      const int rem = mm - b * hw;
      oy[i] = rem / p.Wo;
      ox[i] = rem - oy[i] * p.Wo;
      a_base[i] = p.A + (size_t)b * p.Hs * p.Ws * p.lda;
    } else {
      a_base[i] = p.A + (size_t)(a_ok[i] ? m : 0) * p.lda + kc * 8;
      oy[i] = ox[i] = 0;
    }
    const int n = n0 + r0 + 32 * i;
    w_ok[i] = n < p.N;
    w_base[i] = p.W + (size_t)(w_ok[i] ? n : 0) * p.K + kc * 8;
  }
  // conv: running (tap, channel) of this thread's chunk
  int tap = 0, cch = kc * 8;
  if (CONV) {
    tap = cch / p.Cin;
    cch -= tap * p.Cin;
  }

  u32x4 ra[4], rw[4];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  auto load_tile = [&](int k0) {
    const bool k_ok = (k0 + kc * 8) < p.K;
    if (CONV) {
      const int ky = tap / 3, kx = tap - ky * 3;
      const int Hin = p.Hs << p.up, Win = p.Ws << p.up;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int iy = oy[i] * p.stride + ky - 1;
        const int ix = ox[i] * p.stride + kx - 1;
        const bool ok = a_ok[i] && k_ok && (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win;
        const size_t off = ((size_t)(iy >> p.up) * p.Ws + (ix >> p.up)) * p.lda + cch;
        ra[i] = ok ? *reinterpret_cast<const u32x4*>(a_base[i] + off) : zero4;
      }
      cch += BK;
      while (cch >= p.Cin) {
        cch -= p.Cin;
        ++tap;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        ra[i] = (a_ok[i] && k_ok) ? *reinterpret_cast<const u32x4*>(a_base[i] + k0) : zero4;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      rw[i] = (w_ok[i] && k_ok) ? *reinterpret_cast<const u32x4*>(w_base[i] + k0) : zero4;
  };
  auto store_tile = [&](int buf) {
    unsigned char* a = As + buf * (BM * BK * 2);
    unsigned char* w = Ws + buf * (BN * BK * 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = r0 + 32 * i;
      *reinterpret_cast<u32x4*>(a + row * (BK * 2) + swz) = ra[i];
      *reinterpret_cast<u32x4*>(w + row * (BK * 2) + swz) = rw[i];
    }
  };

  // ---- LDS-DMA loader geometry: wave w fills 1-KiB pieces 4w..4w+3 (8 rows x 128 B each) of A and of W ----
  const int g_sub = lane >> 3;                 // row inside a piece == (row & 7)
  const int g_cg = (lane & 7) ^ g_sub;         // global 16-B chunk this lane fetches (source-side swizzle)
  const bf16* ga_base[4];
  bool ga_ok[4];
  int goy[4], gox[4];
  const bf16* gw_base[4];
  bool gw_ok[4];
  int gtap = 0, gcch = g_cg * 8;
  if (GLDS) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave * 32 + i * 8 + g_sub;
      const int m = m0 + row;
      ga_ok[i] = m < p.M;
      if (CONV) {
        const int hw = p.Ho * p.Wo;
        const int mm = ga_ok[i] ? m : 0;
        const int b = mm / hw;
        const int rem = mm - b * hw;
        goy[i] = rem / p.Wo;
        gox[i] = rem - goy[i] * p.Wo;
        ga_base[i] = p.A + (size_t)b * p.Hs * p.Ws * p.lda;
      } else {
        ga_base[i] = p.A + (size_t)(ga_ok[i] ? m : 0) * p.lda + g_cg * 8;
        goy[i] = gox[i] = 0;
      }
      const int n = n0 + row;
      gw_ok[i] = n < p.N;
      gw_base[i] = p.W + (size_t)(gw_ok[i] ? n : 0) * p.K + g_cg * 8;
    }
    if (CONV) {
      gtap = gcch / p.Cin;
      gcch -= gtap * p.Cin;
    }
  }
  auto issue_tile = [&](int k0, int buf) {
    const bool k_ok = (k0 + g_cg * 8) < p.K;
    unsigned char* a = As + buf * (BM * BK * 2) + wave * 4096;
    unsigned char* w = Ws + buf * (BN * BK * 2) + wave * 4096;
    const bf16* zsrc = reinterpret_cast<const bf16*>(g_zero16);
    if (CONV) {
      const int ky = gtap / 3, kx = gtap - ky * 3;
      const int Hin = p.Hs << p.up, Win = p.Ws << p.up;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int iy = goy[i] * p.stride + ky - 1;
        const int ix = gox[i] * p.stride + kx - 1;
        const bool ok = ga_ok[i] && k_ok && (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win;
        const size_t off = ((size_t)(iy >> p.up) * p.Ws + (ix >> p.up)) * p.lda + gcch;
        const bf16* src = ok ? ga_base[i] + off : zsrc;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(a + i * 1024), 16, 0, 0);
      }
      gcch += BK;
      while (gcch >= p.Cin) {
        gcch -= p.Cin;
        ++gtap;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16* src = (ga_ok[i] && k_ok) ? ga_base[i] + k0 : zsrc;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(a + i * 1024), 16, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16* src = (gw_ok[i] && k_ok) ? gw_base[i] + k0 : zsrc;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(w + i * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read geometry
  const int frow = lane & 15;   // row inside a 16-row MFMA tile
  const int fkc = lane >> 4;    // 16-B chunk inside a 32-wide k-step
  const int a_row0 = wm * 64 + frow;
  const int w_row0 = wn * 64 + frow;
  const int rsw = frow & 7;

  const int nt = (p.K + BK - 1) / BK;
  if (GLDS) {
    issue_tile(0, 0);
  } else {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    if (t + 1 < nt) {
      if (GLDS)
        issue_tile((t + 1) * BK, buf ^ 1);
      else
        load_tile((t + 1) * BK);
    }
    const unsigned char* a = As + buf * (BM * BK * 2);
    const unsigned char* w = Ws + buf * (BN * BK * 2);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = ((ks * 4 + fkc) ^ rsw) << 4;
      bf16x8 fa[4], fw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = *reinterpret_cast<const bf16x8*>(a + (a_row0 + i * 16) * (BK * 2) + coff);
        fw[i] = *reinterpret_cast<const bf16x8*>(w + (w_row0 + i * 16) * (BK * 2) + coff);
      }
#pragma unroll
      for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
          acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[tn], fa[tm], acc[tn][tm], 0, 0, 0);
    }
    if (!GLDS && t + 1 < nt) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds, for row m = .. + (lane&15), channels n = .. + (lane>>4)*4 + {0..3} ----
  const int m_wave = m0 + wm * 64, n_wave = n0 + wn * 64;
  const int nq = (lane >> 4) * 4;
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
    const int m = m_wave + tm * 16 + (lane & 15);
    if (m >= p.M) continue;
    const float* rb = p.rowbias ? p.rowbias + (size_t)(m / p.rows_per_batch) * p.ld_rowbias : nullptr;
    if (p.geglu) {
#pragma unroll
      for (int tp = 0; tp < 2; ++tp) {
        const int n_phys = n_wave + tp * 32 + nq;  // physical (interleaved) column of the value half
        if (n_phys >= p.N) continue;  // N % 32 == 0: the gate half of the pair is inside too
        f32x4 h = acc[2 * tp][tm], g = acc[2 * tp + 1][tm];
        if (p.bias) {
          const f32x4 bh = *reinterpret_cast<const f32x4*>(p.bias + n_phys);
          const f32x4 bg = *reinterpret_cast<const f32x4*>(p.bias + n_phys + 16);
          h += bh;
          g += bg;
        }
        const int n_out = (n_wave >> 1) + tp * 16 + nq;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = h[r] * gelu_erf_f(g[r]);
        u32x2 pk = {pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3])};
        *reinterpret_cast<u32x2*>(reinterpret_cast<bf16*>(p.C) + (size_t)m * p.ldc + n_out) = pk;
      }
    } else {
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        const int n = n_wave + tn * 16 + nq;
        if (n >= p.N) continue;
        f32x4 v = acc[tn][tm];
        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
        if (rb) v += *reinterpret_cast<const f32x4*>(rb + n);
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
        if (p.out_f32) {
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n) = v;
        } else {
          u32x2 pk = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
          *reinterpret_cast<u32x2*>(reinterpret_cast<bf16*>(p.C) + (size_t)m * p.ldc + n) = pk;
        }
      }
    }
  }
}

int launch_gemm(const GemmArgs& a, hipStream_t stream) {
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return SD_ERR_INVALID;
  if ((a.K & 7) || (a.N & 3) || (a.lda & 7) || (a.ldc & 3)) return SD_ERR_UNSUPPORTED;
  if (a.R && (a.ldr & 3)) return SD_ERR_UNSUPPORTED;
  if (a.geglu && ((a.N & 31) || a.out_f32 || a.R || a.rowbias)) return SD_ERR_UNSUPPORTED;
  if (a.conv) {
    if (a.K != 9 * a.Cin || (a.Cin & 7) || (a.stride != 1 && a.stride != 2) || (a.up != 0 && a.up != 1))
      return SD_ERR_UNSUPPORTED;
    if ((long)a.M % ((long)a.Ho * a.Wo) != 0) return SD_ERR_INVALID;
  }
  if (a.rowbias && a.rows_per_batch <= 0) return SD_ERR_INVALID;
  const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
  dim3 grid(ntm * ntn), block(GEMM_THREADS);
  static const bool use_regs = [] {
    const char* e = getenv("MI355X_SD_GEMM");
    return e && strcmp(e, "regs") == 0;
  }();
  if (use_regs) {
    if (a.conv)
      hipLaunchKernelGGL((gemm_bf16_kernel<true, false>), grid, block, 0, stream, a);
    else
      hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), grid, block, 0, stream, a);
  } else {
    if (a.conv)
      hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, block, 0, stream, a);
    else
      hipLaunchKernelGGL((gemm_bf16_kernel<false, true>), grid, block, 0, stream, a);
  }
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

}  // namespace sd
