// 256x256x64 staggered, phase-pipelined bf16 MFMA GEMM / implicit-GEMM conv for gfx950 (the large-problem path of
// launch_gemm; same arguments, layouts and epilogue as gemm.hip).
//
// Why a second structure: at one 8-wave block per CU the plain "issue next tile / multiply / drain / barrier" loop
// leaves the matrix pipe idle while both waves of a SIMD wait on the same barrier and the LDS-DMA queue runs dry
// once per K-tile. Here
//   * the 8 waves form two groups (wave>>2) that run ONE BARRIER APART: while group 0 multiplies, group 1 reads
//     its fragments / issues DMA, and vice versa (waves w and w+4 share a SIMD), so every SIMD always has one
//     wave in an MFMA segment;
//   * a K-tile is four half-tiles (A rows 0-127 / 128-255, W rows 0-127 / 128-255; 16 KB each) and every phase
//     issues exactly one half-tile, seven half-tiles ahead of its consumption -> the DMA queue never drains;
//     the only vmcnt wait is a COUNTED s_waitcnt vmcnt(6) once per K-tile;
//   * per K-tile a wave runs four phases = the four 64x32 quadrants of its 128x64 output (16 MFMAs each).
//
// Schedule (t = K-tile, buffer t&1; half-tile order per tile: B0, B1, A0, A1; I_k = barrier interval):
//   phase q0: ds_read A rows 0-63 + all of B (16 x b128) | issue A1(t+1)          | barrier | MFMA(A0-63 x B0-31)  | barrier
//   phase q1:                                            | issue B0(t+2)          | barrier | MFMA(A0-63 x B32-63) | barrier
//   phase q2: ds_read A rows 64-127 (8 x b128)           | issue B1(t+2)          | barrier | MFMA(A64-127 x B32-63)| barrier
//   phase q3:                                            | issue A0(t+2), vmcnt(6)| barrier | MFMA(A64-127 x B0-31) | barrier
// Hazards. RAW: a wave waits for its own DMA pieces (vmcnt) before the barrier that ends its q3 read segment; group
// 1 does so one interval after group 0, and the first read of tile t+1 (group 0, q0) comes after that barrier.
// WAR: B(t) is last read in q0 (group 1: I_{8t+1}) and B0(t+2) is issued from I_{8t+2} on; A0(t)/A1(t) are last
// read in q2 (I_{8t+4}/I_{8t+5}) and re-staged from I_{8t+6} / I_{8t+8} on; every read segment retires its
// ds_reads (lgkmcnt(0)) before its closing barrier, so "issued after the barrier" implies "after the reads".
#include <stdlib.h>

#include "common.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace sd {

namespace g256 {
constexpr int BM = 256, BN = 256, BK = 64, THREADS = 512;
constexpr int HALF = 128 * BK * 2;        // 16 KB
constexpr int BUF = 4 * HALF;             // A0 A1 B0 B1
constexpr int LDS_BYTES = 2 * BUF;        // 128 KB
}  // namespace g256

#define SD_BARRIER()                      \
  do {                                    \
    __builtin_amdgcn_sched_barrier(0);    \
    __builtin_amdgcn_s_barrier();         \
    __builtin_amdgcn_sched_barrier(0);    \
  } while (0)

// F8 = true: A and W are OCP e4m3 bytes (W8A8), one K-tile = 128 elements = the same 128-byte LDS rows, and the two
// bf16 MFMAs of a (row-tile, column-tile) pair become ONE v_mfma_scale_f32_16x16x128_f8f6f4 (operand layout probed in
// scripts/probes/f8_mfma_probe.hip: lane l supplies row l&15, k = (l>>4)*32 .. +31, i.e. the two adjacent 16-byte LDS
// chunks 2*(l>>4), 2*(l>>4)+1; 2.25x the bf16 issue rate). Same schedule, same register footprint; the per-row (A) and
// per-channel (W) dequantisation scales are applied to the fp32 accumulators in gemm_epilogue_f8.
typedef __attribute__((ext_vector_type(8))) int i32x8;

template <bool CONV, bool LN = false, bool F8 = false, bool WS = false>   // WS: per-channel weight scale in registers (gemm_epilogue.h)
__global__ __launch_bounds__(g256::THREADS, 2) void gemm256_kernel(const GemmArgs p) {
  using namespace g256;
  constexpr int ES = F8 ? 1 : 2;        // bytes per element
  constexpr int KT = F8 ? 128 : 64;     // elements per K-tile (always 128 bytes)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;

  const int ntn = (p.N + BN - 1) / BN;
  const int ntm = (p.M + BM - 1) / BM;
  const int lid = xcd_remap(blockIdx.x, ntm * ntn);
  int tile_m, tile_n;
  tile_coords(lid, ntm, ntn, p.gm, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- LDS-DMA geometry: a half-tile is 16 pieces of 1 KiB (8 rows x 128 B); wave w issues pieces 2w, 2w+1 ----
  const int sub = lane >> 3;
  const int cg = (lane & 7) ^ sub;   // source-side XOR swizzle (LDS destination of a DMA is lane-linear)
  // Buffer (SRD) addressing: 32-bit per-lane byte offsets + a scalar K offset; anything past the last valid
  // byte (rows >= M / N, conv padding -> offset forced out of range) reads as zero in hardware, so there is no
  // zero-page select and no 64-bit address arithmetic in the loop. Requires K % 64 == 0 (no in-row K tail).
  const unsigned a_bytes = CONV ? (unsigned)(((size_t)(p.M / (p.Ho * p.Wo)) * p.Hs * p.Ws - 1) * p.lda + p.Cin) * 2u
                           : F8 ? (unsigned)(p.a_rpb ? (size_t)((p.M - 1) / p.a_rpb) * p.a_bstride + (size_t)(p.a_rpb - 1) * p.lda + p.K
                                                     : (size_t)(p.M - 1) * p.lda + p.K)   // exact: run-out tiles must fall outside
                                : (unsigned)(((size_t)(p.M - 1) * p.lda + p.K) * 2);
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.A), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.W), 0, (unsigned)((size_t)p.N * p.K * ES), 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  unsigned a_off32[4];               // index h*2 + j : half h, piece j (linear: byte offset of the row's chunk)
  int oy[4], ox[4];
  unsigned a_img[4];                 // conv: byte offset of the row's image
  bool a_ok[4];
  unsigned w_off32[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (i >> 1) * 128 + (2 * wave + (i & 1)) * 8 + sub;
    const int m = m0 + r;
    a_ok[i] = m < p.M;
    if (CONV) {
      const int hw = p.Ho * p.Wo;
      const int mm = a_ok[i] ? m : 0;
      const int b = mm / hw;
      const int rem = mm - b * hw;
      oy[i] = rem / p.Wo;
      ox[i] = rem - oy[i] * p.Wo;
      a_img[i] = (unsigned)((size_t)b * p.Hs * p.Ws * p.lda * 2);
      a_off32[i] = 0;
    } else {
      const size_t arow = (F8 && p.a_rpb) ? (size_t)(m / p.a_rpb) * p.a_bstride + (size_t)(m % p.a_rpb) * p.lda : (size_t)m * p.lda;
      a_off32[i] = a_ok[i] ? (unsigned)(arow * ES + cg * 16) : OOB;
      oy[i] = ox[i] = 0;
      a_img[i] = 0;
    }
    const int n = n0 + w_row_of_lds_row<4>(r, p.geglu);   // epilogue-friendly channel order (gemm_epilogue.h)
    w_off32[i] = (n < p.N) ? (unsigned)((size_t)n * p.K * ES + cg * 16) : OOB;
  }
  int kA[2] = {0, 0}, kB[2] = {0, 0};              // next K offset (elements) of each half-tile stream
  int tapA[2] = {0, 0}, chA[2] = {cg * 8, cg * 8}; // conv: running (tap, channel) per A stream
  if (CONV) {
    conv_k_init(p.kb64, 0, cg * 8, p.Cin, tapA[0], chA[0]);
    tapA[1] = tapA[0];
    chA[1] = chA[0];
  }

  auto issue_A = [&](const int h, int buf) {
    unsigned char* dst = smem + buf * BUF + h * HALF + wave * 2048;
    if (CONV) {
      const int ky = tapA[h] / 3, kx = tapA[h] - ky * 3;
      const int Hin = p.Hs << p.up, Win = p.Ws << p.up;
      const bool k_ok = p.kb64 ? chA[h] < p.Cin : tapA[h] < 9;   // run-out tiles past K
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int i = h * 2 + j;
        const int iy = oy[i] * p.stride + ky - p.pad;
        const int ix = ox[i] * p.stride + kx - p.pad;
        const bool ok = a_ok[i] && k_ok && (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win;
        const unsigned off = a_img[i] + (unsigned)(((iy >> p.up) * p.Ws + (ix >> p.up)) * p.lda + chA[h]) * 2u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lptr_t)(dst + j * 1024), 16, ok ? off : OOB, 0, 0, 0);
      }
      conv_k_next(p.kb64, p.Cin, tapA[h], chA[h]);
    } else {
      const int soff = (kA[h] < p.K) ? kA[h] * ES : (int)0x7FFFFFF0;   // run-out tiles: out of range -> zeros
#pragma unroll
      for (int j = 0; j < 2; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lptr_t)(dst + j * 1024), 16, a_off32[h * 2 + j], soff, 0, 0);
    }
    kA[h] += KT;
  };
  auto issue_B = [&](const int h, int buf) {
    unsigned char* dst = smem + buf * BUF + (2 + h) * HALF + wave * 2048;
    const int soff = (kB[h] < p.K) ? kB[h] * ES : (int)0x7FFFFFF0;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lptr_t)(dst + j * 1024), 16, w_off32[h * 2 + j], soff, 0, 0);
    kB[h] += KT;
  };

  f32x4 acc[4][8];   // [n-tile][m-tile]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragment geometry ----
  const int frow = lane & 15, fkc = lane >> 4, rsw = frow & 7;
  // 16-B chunks of a 128-B row this lane reads: bf16 k-steps 0/1 -> chunks fkc, 4+fkc; fp8 -> the adjacent pair 2fkc, 2fkc+1
  const int c0 = ((F8 ? 2 * fkc : fkc) ^ rsw) << 4, c1 = ((F8 ? 2 * fkc + 1 : 4 + fkc) ^ rsw) << 4;
  const int a_off = grp * HALF + frow * 128;                              // + (s*64 + mt*16)*128
  const int b_off = (2 + (wc >> 1)) * HALF + ((wc & 1) * 64 + frow) * 128; // + nt*16*128
  bf16x8 fa[4][2], fb[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) fa[i][0] = fa[i][1] = fb[i][0] = fb[i][1] = bf16x8{};

  auto read_A = [&](const unsigned char* base, const int s) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const unsigned char* r = base + a_off + (s * 64 + mt * 16) * 128;
      fa[mt][0] = *reinterpret_cast<const bf16x8*>(r + c0);
      fa[mt][1] = *reinterpret_cast<const bf16x8*>(r + c1);
    }
  };
  auto read_B = [&](const unsigned char* base) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const unsigned char* r = base + b_off + nt * 16 * 128;
      fb[nt][0] = *reinterpret_cast<const bf16x8*>(r + c0);
      fb[nt][1] = *reinterpret_cast<const bf16x8*>(r + c1);
    }
  };
  auto mma = [&](const int s, const int j) {   // quadrant: A rows s*64.., B cols j*32..
    __builtin_amdgcn_s_setprio(1);
    if constexpr (F8) {
#pragma unroll
      for (int nn = 0; nn < 2; ++nn)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          i32x8 wa, xa;   // 32 operand bytes = the two 16-byte fragments
          __builtin_memcpy(&wa, &fb[2 * j + nn][0], 32);
          __builtin_memcpy(&xa, &fa[mt][0], 32);
          acc[2 * j + nn][s * 4 + mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
              wa, xa, acc[2 * j + nn][s * 4 + mt], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        }
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
            acc[2 * j + nn][s * 4 + mt] =
                mfma_16x16x32(fb[2 * j + nn][ks], fa[mt][ks], acc[2 * j + nn][s * 4 + mt]);
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- prologue: tile 0 (B0 B1 A0 A1) and tile 1 (B0 B1 A0): 7 half-tiles = 14 DMAs per lane ----
  issue_B(0, 0);
  issue_B(1, 0);
  issue_A(0, 0);
  issue_A(1, 0);
  issue_B(0, 1);
  issue_B(1, 1);
  issue_A(0, 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // tile 0 landed (own pieces)
  SD_BARRIER();
  if (grp == 1) SD_BARRIER();                        // stagger: group 1 runs one barrier behind group 0

  const int nt = (p.K + KT - 1) / KT;
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    const unsigned char* base = smem + buf * BUF;
    // q0
    read_A(base, 0);
    read_B(base);
    issue_A(1, buf ^ 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SD_BARRIER();
    mma(0, 0);
    SD_BARRIER();
    // q1
    issue_B(0, buf);
    SD_BARRIER();
    mma(0, 1);
    SD_BARRIER();
    // q2
    read_A(base, 1);
    issue_B(1, buf);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SD_BARRIER();
    mma(1, 1);
    SD_BARRIER();
    // q3
    issue_A(0, buf);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // tile t+1 landed (own pieces); 3 half-tiles stay in flight
    SD_BARRIER();
    mma(1, 0);
    SD_BARRIER();
  }
  if (grp == 0) SD_BARRIER();                         // re-align the two groups
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // run-out DMAs retired before the LDS is released

  if constexpr (F8) gemm_epilogue_f8<8, 4>(p, acc, m0 + grp * 128, n0 + wc * 64, lane);
  else if constexpr (LN) gemm_epilogue_ln<8, 4>(p, acc, m0 + grp * 128, n0 + wc * 64, lane);
  else gemm_epilogue<8, 4, WS>(p, acc, m0 + grp * 128, n0 + wc * 64, lane);
}

int launch_gemm256(const GemmArgs& a, hipStream_t stream) {
  using namespace g256;
  static const bool attr_ok = [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<false, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<false, false, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<false, false, false, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) == hipSuccess;
  }();
  if (!attr_ok) return SD_ERR_HIP;
  const GemmArgs& b = a;
  const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
  if (a.ascale)   // W8A8: fp8 e4m3 operands (launch_gemm_f8 validated the arguments)
    hipLaunchKernelGGL((gemm256_kernel<false, false, true>), dim3(ntm * ntn), dim3(THREADS), LDS_BYTES, stream, b);
  else if (a.rowstat)
    hipLaunchKernelGGL((gemm256_kernel<false, true>), dim3(ntm * ntn), dim3(THREADS), LDS_BYTES, stream, b);
  else if (a.conv)
    hipLaunchKernelGGL(gemm256_kernel<true>, dim3(ntm * ntn), dim3(THREADS), LDS_BYTES, stream, b);
  else if (a.wscale)   // a widened e4m3 matrix (launch_gemm) with its per-channel scale
    hipLaunchKernelGGL((gemm256_kernel<false, false, false, true>), dim3(ntm * ntn), dim3(THREADS), LDS_BYTES, stream, b);
  else
    hipLaunchKernelGGL(gemm256_kernel<false>, dim3(ntm * ntn), dim3(THREADS), LDS_BYTES, stream, b);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

int launch_gemm_f8(const GemmArgs& a_in, hipStream_t stream) {
  GemmArgs a = a_in;
  a.gm = gemm_gm();
  a.c_wide = !a.out_f8 && !(reinterpret_cast<uintptr_t>(a.C) & 15) && !(a.ldc & 7) && !(a.c_bstride & 7);
  if (a.out_f8 && ((a.ldc & 7) || a.c_rpb || a.R || a.gate)) return SD_ERR_UNSUPPORTED;   // 8-byte e4m3 stores
  if (a.M <= 0 || a.N <= 0 || a.K <= 0 || !a.ascale || !a.wscale) return SD_ERR_INVALID;
  // K-tiles of 128 bytes with no in-row tail; 16-byte aligned rows; 32-bit byte offsets
  if ((a.K & 127) || (a.N & 3) || (a.lda & 15) || (a.ldc & 3) || a.conv || a.geglu || a.out_f32 || a.rowbias || a.rowstat ||
      a.silu)
    return SD_ERR_UNSUPPORTED;
  if (a.R && (a.ldr & 3)) return SD_ERR_UNSUPPORTED;
  if (a.gate && a.rows_per_batch <= 0) return SD_ERR_INVALID;
  if ((a.a_rpb && (a.a_bstride & 15)) || (a.c_rpb && (a.c_bstride & 3))) return SD_ERR_UNSUPPORTED;
  const size_t a_ext = a.a_rpb ? (size_t)((a.M - 1) / a.a_rpb) * a.a_bstride + (size_t)(a.a_rpb - 1) * a.lda + a.K
                               : (size_t)(a.M - 1) * a.lda + a.K;
  if (a_ext >= 0xFFFF0000ull || (size_t)a.N * a.K >= 0xFFFF0000ull) return SD_ERR_UNSUPPORTED;
  // (A four-wave form of this tile with a rolling fragment set -- gemm_w4.hip's structure on e4m3 operands -- was built in round 6:
  // bit-identical, 7 % faster on the K = 6144 launch in isolation, and no faster in the SD3 W8A8 step: 55.4 vs 55.5 ms, two interleaved
  // rounds. Removed; source in the history at 745b95f, numbers in profiles/r06_s26_*, r06_s28_switch_check.txt)
  return launch_gemm256(a, stream);
}

}  // namespace sd
