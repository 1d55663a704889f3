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
// Role: tile selection (pick_tile), argument validation, split-K planning / reduction, and the GENERIC loop that serves
// what the specialised loops cannot (K % 64 != 0, fp8 weights, operands >= 4 GiB). The default loops are gemm_pipe.hip
// (128x128, 256x160, 256x320 tiles; software-pipelined) and gemm256.hip (256x256, phased).
//
// Kernel shape (template): WAVES_M x WAVES_N waves, each owning TM x TN v_mfma_f32_16x16x32_bf16 tiles, BK = 64.
//   * 128x128 tile: 2x2 waves of 64x64   (64 KB LDS, 2 blocks / CU)   -- small / ragged problems
//   * 256x256 tile: 2x4 waves of 128x64  (128 KB LDS, 1 block / CU)   -- large problems (see also gemm256.hip)
// Operands go HBM -> LDS directly (global_load_lds_dwordx4: 1 KiB per wave instruction, no staging VGPRs, no
// ds_write pass), double buffered, the next K-tile in flight while the current one is multiplied. LDS rows are
// 128 B with the 16-B chunk index XOR-swizzled by (row & 7) so the ds_read_b128 fragment reads are bank-conflict
// free; since the LDS destination of an LDS-DMA is lane-linear the swizzle is applied to the per-lane SOURCE
// address. Out-of-range chunks (M/N/K tails, the conv zero padding) are sourced from 16 zero bytes.
// The MFMA is issued "swapped" (W as the row operand) so every lane ends up with 4 consecutive output channels
// of one row -> 8-byte stores, and the GEGLU value/gate pair of a channel sits in one lane.
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "gemm_cfg.h"
#include <utility>
#include <vector>

#include "gemm_epilogue.h"
#include "kernels.h"

namespace sd {

// W8 = true: W is fp8 e4m3 (OCP), 64-B LDS rows (16 rows per DMA piece, 16-B chunk index XOR (row>>2)&3 so the 8-byte
// fragment reads are conflict free); fragments are widened to bf16 in registers (every e4m3 value is exact in bf16) and the
// per-channel scale is applied to the fp32 accumulator in the epilogue. Halves the weight bytes a CU has to ingest.
template <bool CONV, class CFG, bool W8 = false, bool LN = false>
__global__ __launch_bounds__(CFG::THREADS, 2) void gemm_bf16_kernel(const GemmArgs p) {
  constexpr int BM = CFG::BM, BN = CFG::BN, TM = CFG::TM, TN = CFG::TN;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;                                 // [STAGES][BM][BK] bf16, swizzled
  unsigned char* Ws = smem + CFG::STAGES * BM * BK * 2;     // [STAGES][BN][BK] bf16 (or fp8: half of it used)
  constexpr int W_ROW = W8 ? BK : BK * 2;         // bytes per LDS row of W
  constexpr int W_PIECES = W8 ? (BN / 16 + CFG::NW - 1) / CFG::NW : CFG::W_PIECES;
  constexpr int W_TOTAL = W8 ? BN / 16 : CFG::W_TOTAL;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / CFG::WAVES_N, wn = wave % CFG::WAVES_N;

  const int ntn = (p.N + BN - 1) / BN;
  const int ntm = (p.M + BM - 1) / BM;
  const int lid = xcd_remap(blockIdx.x, ntm * ntn);
  int tile_m, tile_n;
  tile_coords(lid, ntm, ntn, p.gm, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- LDS-DMA loader geometry: wave w fills pieces w*PIECES .. of A and of W (piece = 8 rows x 128 B) ----
  const int g_sub = lane >> 3;                 // row inside a piece == (row & 7)
  const int g_cg = (lane & 7) ^ g_sub;         // global 16-B chunk this lane fetches (source-side swizzle)
  const bf16* ga_base[CFG::A_PIECES];
  bool ga_ok[CFG::A_PIECES];
  int goy[CFG::A_PIECES], gox[CFG::A_PIECES];
  const unsigned char* gw_base[W_PIECES];   // byte pointers (bf16 or fp8 rows)
  bool gw_ok[W_PIECES];
#pragma unroll
  for (int i = 0; i < CFG::A_PIECES; ++i) {
    const int m = m0 + (wave + i * CFG::NW) * 8 + g_sub;   // piece index = wave + i*NW (interleaved over waves)
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
      const int mm = ga_ok[i] ? m : 0;
      const size_t arow = p.a_rpb ? (size_t)(mm / p.a_rpb) * p.a_bstride + (size_t)(mm % p.a_rpb) * p.lda : (size_t)mm * p.lda;
      ga_base[i] = p.A + arow + g_cg * 8;
      goy[i] = gox[i] = 0;
    }
  }
  const int w_sub = W8 ? (lane >> 2) : g_sub;                                  // row inside a W piece
  const int w_cg = W8 ? ((lane & 3) ^ ((lane >> 4) & 3)) : g_cg;               // source 16-B chunk (swizzled)
#pragma unroll
  for (int i = 0; i < W_PIECES; ++i) {
    const int n = n0 + w_row_of_lds_row<TN>((wave + i * CFG::NW) * (W8 ? 16 : 8) + w_sub, p.geglu);   // epilogue-friendly order
    gw_ok[i] = n < p.N;
    gw_base[i] = reinterpret_cast<const unsigned char*>(p.W) + (size_t)(gw_ok[i] ? n : 0) * p.K * (W8 ? 1 : 2) + w_cg * 16;
  }
  const int nt_all = (p.K + BK - 1) / BK;
  int t0 = 0, t1 = nt_all;   // k-tile range of this block (split-K: blockIdx.y picks the slice)
  if (p.splitk > 1) {
    t0 = blockIdx.y * p.kc;
    t1 = min(nt_all, t0 + p.kc);
  }
  int gtap = 0, gcch = t0 * BK + g_cg * 8;   // conv: running (tap, channel) of this lane's chunk
  if (CONV) conv_k_init(p.kb64, t0, g_cg * 8, p.Cin, gtap, gcch);
  const bf16* zsrc = reinterpret_cast<const bf16*>(g_zero16);

  auto issue_tile = [&](int k0, int buf) {
    const bool k_ok = (k0 + g_cg * 8) < p.K;
    unsigned char* a = As + buf * (BM * BK * 2) + wave * 1024;   // piece (wave + i*NW)
    unsigned char* w = Ws + buf * (BN * W_ROW) + wave * 1024;
    if (CONV) {
      const int ky = gtap / 3, kx = gtap - ky * 3;
      const int Hin = p.Hs << p.up, Win = p.Ws << p.up;
#pragma unroll
      for (int i = 0; i < CFG::A_PIECES; ++i) {
        const int iy = goy[i] * p.stride + ky - p.pad;
        const int ix = gox[i] * p.stride + kx - p.pad;
        const bool ok = ga_ok[i] && k_ok && (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win;
        const size_t off = ((size_t)(iy >> p.up) * p.Ws + (ix >> p.up)) * p.lda + gcch;
        const bf16* src = ok ? ga_base[i] + off : zsrc;
        if (CFG::A_TOTAL % CFG::NW == 0 || wave + i * CFG::NW < CFG::A_TOTAL)
          __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(a + i * (CFG::NW * 1024)), 16, 0, 0);
      }
      conv_k_next(p.kb64, p.Cin, gtap, gcch);
    } else {
#pragma unroll
      for (int i = 0; i < CFG::A_PIECES; ++i) {
        const bf16* src = (ga_ok[i] && k_ok) ? ga_base[i] + k0 : zsrc;
        if (CFG::A_TOTAL % CFG::NW == 0 || wave + i * CFG::NW < CFG::A_TOTAL)
          __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(a + i * (CFG::NW * 1024)), 16, 0, 0);
      }
    }
    const bool kw_ok = W8 ? (k0 + w_cg * 16) < p.K : k_ok;
#pragma unroll
    for (int i = 0; i < W_PIECES; ++i) {
      const void* src = (gw_ok[i] && kw_ok) ? (const void*)(gw_base[i] + (size_t)k0 * (W8 ? 1 : 2)) : (const void*)zsrc;
      if (W_TOTAL % CFG::NW == 0 || wave + i * CFG::NW < W_TOTAL)
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(w + i * (CFG::NW * 1024)), 16, 0, 0);
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read geometry
  const int frow = lane & 15;   // row inside a 16-row MFMA tile
  const int fkc = lane >> 4;    // 16-B chunk inside a 32-wide k-step
  const int a_row0 = wm * (TM * 16) + frow;
  const int w_row0 = wn * (TN * 16) + frow;
  const int rsw = frow & 7;

  auto multiply = [&](const int buf) {
    const unsigned char* a = As + buf * (BM * BK * 2);
    const unsigned char* w = Ws + buf * (BN * W_ROW);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = ((ks * 4 + fkc) ^ rsw) << 4;
      bf16x8 fa[TM], fw[TN];
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        if (W8) {
          // 8 fp8 of row (w_row0 + 16 i): 16-B chunk 2*ks + (fkc >> 1) swizzled by (row >> 2) & 3, half fkc & 1
          const int c16 = ((2 * ks + (fkc >> 1)) ^ ((frow >> 2) & 3)) << 4;
          const u32x2 raw = *reinterpret_cast<const u32x2*>(w + (w_row0 + i * 16) * BK + c16 + (fkc & 1) * 8);
          const auto f0 = __builtin_amdgcn_cvt_pk_f32_fp8(raw[0], false), f1 = __builtin_amdgcn_cvt_pk_f32_fp8(raw[0], true);
          const auto f2 = __builtin_amdgcn_cvt_pk_f32_fp8(raw[1], false), f3 = __builtin_amdgcn_cvt_pk_f32_fp8(raw[1], true);
          fw[i] = bf16x8{(bf16)f0[0], (bf16)f0[1], (bf16)f1[0], (bf16)f1[1], (bf16)f2[0], (bf16)f2[1], (bf16)f3[0], (bf16)f3[1]};
        } else {
          fw[i] = *reinterpret_cast<const bf16x8*>(w + (w_row0 + i * 16) * (BK * 2) + coff);
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(a + (a_row0 + i * 16) * (BK * 2) + coff);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
          acc[tn][tm] = mfma_16x16x32(fw[tn], fa[tm], acc[tn][tm]);
    }
  };

  static_assert(CFG::STAGES == 2, "the generic loop double-buffers; the 3-stage ring lives in gemm_pipe.hip");
  issue_tile(t0 * BK, 0);
  __syncthreads();
  for (int t = t0; t < t1; ++t) {
    const int buf = (t - t0) & 1;
    if (t + 1 < t1) issue_tile((t + 1) * BK, buf ^ 1);
    multiply(buf);
    __syncthreads();
  }

  if (p.splitk > 1) {   // raw partial sums -> ws[split][m][n]; the epilogue runs in splitk_reduce_kernel
    float* ws = p.ws + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int m = m0 + wm * (TM * 16) + tm * 16 + (lane & 15);
      if (m >= p.M) continue;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + wn * (TN * 16) + acc_col<TN>(tn, lane >> 4, p.geglu);
        if (n < p.N) *reinterpret_cast<f32x4*>(ws + (size_t)m * p.N + n) = acc[tn][tm];
      }
    }
    return;
  }
  if constexpr (LN) gemm_epilogue_ln<TM, TN>(p, acc, m0 + wm * (TM * 16), n0 + wn * (TN * 16), lane);
  else gemm_epilogue<TM, TN, W8>(p, acc, m0 + wm * (TM * 16), n0 + wn * (TN * 16), lane);   // (W8 launches always carry the scale)
}

// Deterministic split-K tail: wave = 16 rows x 32 columns in the MFMA accumulator layout, slices summed in index
// order, then the common epilogue (so every fusion -- bias, temb, residual, GEGLU, fp8 scales -- is shared).
// WAVES per block: 4, or 1 where four-wave blocks would leave most CUs without one (64 x 1280 outputs = 40 blocks of four waves: the
// 160 waves then sit on 40 CUs and the 14 MB of slabs of a deep batch-1 convolution are pulled through those)
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void splitk_reduce_kernel(const GemmArgs p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m_wave = blockIdx.x * 16, n_wave = blockIdx.y * (32 * WAVES) + wave * 32;
  const int m = m_wave + (lane & 15);
  f32x4 acc[2][1] = {{f32x4{0.f, 0.f, 0.f, 0.f}}, {f32x4{0.f, 0.f, 0.f, 0.f}}};
  if (m < p.M) {
    const size_t slice = (size_t)p.M * p.N;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int n = n_wave + acc_col<2>(tn, lane >> 4, p.geglu);
      if (n >= p.N) continue;
      const float* src = p.ws + (size_t)m * p.N + n;
#pragma unroll 8
      for (int s = 0; s < p.splitk; ++s) acc[tn][0] += *reinterpret_cast<const f32x4*>(src + s * slice);   // (index order: deterministic)
    }
  }
  if (p.rowstat) gemm_epilogue_ln<1, 2>(p, acc, m_wave, n_wave, lane);
  else gemm_epilogue<1, 2>(p, acc, m_wave, n_wave, lane);
}

void launch_splitk_reduce(const GemmArgs& a, hipStream_t stream) {
  const long blocks4 = (long)((a.M + 15) / 16) * ((a.N + 127) / 128);
  if (blocks4 < 512) hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3((a.M + 15) / 16, (a.N + 31) / 32), dim3(64), 0, stream, a);
  else hipLaunchKernelGGL(splitk_reduce_kernel<4>, dim3((a.M + 15) / 16, (a.N + 127) / 128), dim3(256), 0, stream, a);
}

// Split-K plan for launches that cannot fill the chip (SD-1.5 at batch 1: 64..1024 rows against K up to 23040, i.e.
// weight-streaming problems where 10-40 tiles would otherwise pull the whole weight matrix through 10-40 CUs).
// Up to 160 tiles are sliced, towards ~416 blocks, at least 4 k-tiles per slice, partial sums bounded by the workspace. The two
// constants were 128 / 512 through round 5; scanned inside the batch-1 SD-1.5 step (plain-C step bench, two interleaved rounds each,
// profiles/r06_s32*_splitk_policy.txt): 128:512 5.41 ms, 64:512 5.95, 128:256 5.39, 128:384 5.32, 160:384 5.27, 160:416 5.23,
// 192:384 5.28, 160:512 5.35 -- fewer, longer slices (less slab traffic for the reduce kernel) and slicing a little above 128 tiles.
static void plan_splitk(GemmArgs& a, int bm, int bn) {
  a.splitk = 0;
  static const bool off = sd_switch("MI355X_SD_NO_SPLITK") != nullptr;
  if (!a.ws_base || off || a.w16) return;   // (a widened fp8 matrix occupies the workspace)
  const long tiles = (long)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn);
  const int nt = (a.K + BK - 1) / BK;
  // (max tiles / target blocks / minimum K-tiles per slice; the scans behind them: profiles/r06_s32*_splitk_policy.txt -- the
  // minimum slice length is flat between 4 and 8)
  if (tiles > 160 || nt < 8) return;
  long s = (416 + tiles - 1) / tiles;
  s = std::min<long>(s, nt / 4);
  const size_t slice = (size_t)a.M * a.N * sizeof(float);
  s = std::min<long>(s, (long)(a.ws_bytes / slice));
  if (s < 2) return;
  a.kc = (nt + (int)s - 1) / (int)s;
  a.splitk = (nt + a.kc - 1) / a.kc;
  a.ws = static_cast<float*>(a.ws_base);
  if (a.splitk < 2) a.splitk = 0;
}

template <bool CONV, class CFG, bool W8 = false, bool LN = false>
static int launch_cfg(const GemmArgs& a, hipStream_t stream) {
  static const bool attr_ok = [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<CONV, CFG, W8, LN>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES) == hipSuccess;
  }();
  if (!attr_ok) return SD_ERR_HIP;
  const int ntm = (a.M + CFG::BM - 1) / CFG::BM, ntn = (a.N + CFG::BN - 1) / CFG::BN;
  const int ny = a.splitk > 1 ? a.splitk : 1;
  hipLaunchKernelGGL((gemm_bf16_kernel<CONV, CFG, W8, LN>), dim3(ntm * ntn, ny), dim3(CFG::THREADS), CFG::LDS_BYTES, stream, a);
  if (a.splitk > 1) launch_splitk_reduce(a, stream);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

// Tile selection. These kernels are bound by the per-CU HBM/L2 -> LDS ingest rate (measured ~35-45 GB/s per CU,
// profiles/r01_gemm_tiles.txt), so the cost of a configuration is modelled as the bytes the busiest CU must pull:
//   ceil(tiles / (256 CUs)) * (BM + BN)            (x K, common to all candidates)
// i.e. prefer the largest tile that still gives every CU work and leaves no mostly-empty last round.
// MI355X_SD_GEMM_TILE forces a configuration for A/B measurements: 128 | 256 | 257 (phased 256x256) | 160 | 320.
struct TileChoice { int id, bm, bn; };
static inline bool w_is_f8(const GemmArgs& a) { return a.wscale && !a.w16; }   // the kernel itself reads e4m3 weight bytes
static int pick_tile_model(const GemmArgs& a);
static int pick_tile(const GemmArgs& a) {
  static const int forced = [] {
    const char* e = sd_switch("MI355X_SD_GEMM_TILE");
    return e ? atoi(e) : 0;
  }();
  if (forced) return forced;
  // MI355X_SD_GEMM_TILE_MAP="from:to,from:to" re-maps the model's choice per tile class (A/B measurements inside the step)
  static const std::vector<std::pair<int, int>> remap = [] {
    std::vector<std::pair<int, int>> v;
    const char* e = sd_switch("MI355X_SD_GEMM_TILE_MAP");
    while (e && *e) {
      char* end = nullptr;
      const long from = strtol(e, &end, 10);
      if (!end || *end != ':') break;
      const long to = strtol(end + 1, &end, 10);
      v.emplace_back((int)from, (int)to);
      e = (*end == ',') ? end + 1 : end;
      if (*end != ',') break;
    }
    return v;
  }();
  int chosen = pick_tile_model(a);
  // The wide launches without per-row epilogue operands -- fused QKV, FF1 (GEGLU / tanh-GELU), N >= 1536 with at least three quarters
  // of a chip of 256 x 256 tiles -- take the four-wave tile (gemm_w4.hip, id 258): sustained and interleaved against the tiles above
  // (profiles/r06_s6_w4_probe.txt) 8192 x 3840 x 1280 72.5 vs 82.6 us, 8192 x 10240 x 1280 GEGLU 172 vs 188, 32768 x 5120 x 640 GEGLU
  // 210 vs 228, 32768 x 1920 x 640 101 vs 120. MI355X_SD_GEMM_TILE_MAP="258:160" (debug build) maps it away again.
  static const bool w4_off = sd_switch("MI355X_SD_NO_W4") != nullptr;
  if (!w4_off && a.M >= 2048 && a.N >= 1536 && gemm_w4_applies(a) && (long)((a.M + 255) / 256) * ((a.N + 255) / 256) >= 192) chosen = 258;
  // The four-wave tile at 256 x 160 (id 259: a wave owns 128 x 80) on the launches the model gives the eight-wave 256 x 160 tile
  // (N = 640 / 1280: to_q / to_out / FF2 with their residuals) was built and LOST in the step: 57.10 / 56.71 / 56.96 vs 56.40 / 55.91 /
  // 56.15 ms, three interleaved rounds (profiles/r06_s24_step_ab.txt) -- one tile per CU there: prologue and epilogue dominate, and two
  // waves per SIMD overlap them where one cannot. It stays selectable (MI355X_SD_GEMM_TILE=259 or MI355X_SD_GEMM_TILE_MAP="160:259",
  // debug build) and bit-identical to the generic loop (tests/test_gpu_gemm_variants.py).
  // The small launches of a batch-1 step (<= 128 tiles of 128 x 128, K <= 2560: what used to take split-K slices + a reduce kernel)
  // take the 64 x 64 tile with the six-stage ring (gemm_small.hip, id 64). MI355X_SD_NO_SMALL (debug build): the round-5 path.
  static const bool small_off = sd_switch("MI355X_SD_NO_SMALL") != nullptr;
  if (!small_off && chosen == 128 && gemm_small_applies(a)) chosen = 64;
  for (const auto& m : remap)
    if (m.first == chosen && !(a.geglu && (m.second == 129 || m.second == 160)) && a.M >= 256) return m.second;
  return chosen;
}
static int pick_tile_model(const GemmArgs& a) {
  if (a.M < 256) return 128;
  const bool wide_ok = (size_t)a.M * a.lda * 2 < (1ull << 31) && (size_t)a.N * a.K * 2 < (1ull << 31);
  const TileChoice cand[] = {{128, 128, 128}, {160, 256, 160}, {257, 256, 256}, {320, 256, 320}, {256, 256, 256}};
  int best = 128;
  double best_cost = 1e30;
  for (const TileChoice& c : cand) {
    if (c.id == 257 && ((a.K & 63) || (a.conv && (a.Cin & 63)) || !wide_ok || a.a_rpb || w_is_f8(a))) continue;   // (16-bit operands only; a widened e4m3 matrix is one)
    if (a.geglu && c.id == 160) continue;   // odd number of 16-column tiles per wave
    if (c.id == 320 && a.wscale && a.w16 && !a.geglu) continue;   // widened e4m3 matrix: the tiles whose epilogue holds the scale in registers
    if (c.id == 256 && !w_is_f8(a) && !a.a_rpb) continue;   // the plain 256x256 loop only where the phased kernel cannot run
    const long tiles = (long)((a.M + c.bm - 1) / c.bm) * ((a.N + c.bn - 1) / c.bn);
    const long per_cu = (tiles + 255) / 256;
    double cost = (double)per_cu * (c.bm + c.bn);
    if (c.id == 128) cost *= 0.95;   // two co-resident blocks hide each other's prologue / epilogue
    // Plain GEMMs (no GEGLU, not a conv) prefer the 256x160 tile although the bytes-per-CU model below ranks the larger tiles
    // first: its rounds are exact multiples of the chip on the SDXL / SD3 widths, its residual arrives early
    // (gemm_pipe_pre_kernel), and since round 4 it runs the interleaved loop (gemm_pipe.hip). Measured INSIDE the step with every
    // launch forced onto each family in turn (profiles/r04_s2_step_shapes_ab.txt, ms per step, 160 / 257 / 320): to_out
    // 8192x1280x1280 7.04 / 10.86 / 11.80, fused QKV 5.70 / 5.94 / 6.93, FF2 (K = 5120) 5.29 / 8.28 / 9.10, 32768x640x2560 1.01 /
    // 1.66 / 1.09, 32768x1920x640 1.24 / 1.37 / 1.28, the all-layer cross-attention K/V projection 616x166400x2048 0.48 / 0.50 /
    // 0.51. Round 3 applied the weight for K <= 1536 only: +0.3 ms per step (r04_s3_step_ab.txt). GEGLU launches cannot take the
    // tile (odd number of sub-tiles per wave); the convs keep the model's choice (forced onto 160 they lose 0.35 ms per step).
    if (c.id == 160 && !a.conv && !a.geglu && !w_is_f8(a) && !a.rowstat && !a.a_rpb && !a.c_rpb) cost *= 0.6;   // (measured forms only)
    if (cost < best_cost) {
      best_cost = cost;
      best = c.id;
    }
  }
  return best;
}

// Tile rasterisation (common.h tile_coords), measured inside the SDXL bs-8 step (profiles/r02_o_gemm_gm.txt, one box, ms per
// step): row groups of 2 / 4 / 8 / 16 / 32: 60.98 / 61.48 / 61.19-61.50 / 61.92 / 64.72; column groups of 2 / 4 / 8: 61.11 /
// 60.37 / 60.47. Column groups of 4: an XCD's concurrently running tiles keep the same 4 W column-panels (the small operand)
// hot in its L2 while the A row-panels stream through once.
constexpr int GEMM_GM_DEFAULT = -4;
int gemm_gm() { return GEMM_GM_DEFAULT; }

// Weight-only fp8 on the pipelined loops (round 4; BASELINE config 5). The generic loop widens e4m3 weight bytes to 16 bits in its
// fragment load (8 VALU conversions per fragment next to the MFMAs) and was slower than the same model with 16-bit weights --
// 12.1 vs 13.2 steps/s on SD3-medium bs 8 (profiles/r03_f_bench_sd3_bs8{,_fp8w}.json): the large-M GEMMs are not weight-bound.
// Large launches now widen the matrix ONCE, just in time, into the caller's workspace (exact: every e4m3 value is a bf16 / fp16
// value; the per-channel scale stays in the epilogue) and run the 16-bit kernels on it: N x K bytes read + 2 N K written, ~1 % of
// such a launch, the same products in the same K order as the in-fragment conversion (the bias then starts the accumulators as
// bias / scale instead of being added after the scale: equal to fp32 rounding). HBM keeps holding one byte per weight.
// Small-M launches (the 1232-row context stream at bs 8, anything that takes split-K slices) ARE weight-bound and stay on the
// generic fp8 loop, whose split-K partial sums own the workspace.
__device__ __forceinline__ void widen16(const u32x4 raw, u32x4* __restrict__ dst) {   // 16 e4m3 bytes -> 16 x 16-bit elements
  u32x4 o[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const auto f0 = __builtin_amdgcn_cvt_pk_f32_fp8(raw[j], false), f1 = __builtin_amdgcn_cvt_pk_f32_fp8(raw[j], true);
    o[j >> 1][(j & 1) * 2] = pack_bf16(f0[0], f0[1]);
    o[j >> 1][(j & 1) * 2 + 1] = pack_bf16(f1[0], f1[1]);
  }
  dst[0] = o[0];
  dst[1] = o[1];
}
__global__ __launch_bounds__(256) void widen_fp8_kernel(const u32x4* __restrict__ w8, u32x4* __restrict__ w16, long n16) {
  // 16 weights per lane and load: one 16-byte load, two 16-byte stores; four loads in flight per lane (round 5: 4-byte loads, then
  // one 16-byte load per lane, ran at 1.5-1.8 TB/s -- profiles/r05_s24_sd3_fp8w_kernel_stats.txt: 8.6 us per launch, 0.83 ms per step)
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const u32x4 r0 = __builtin_nontemporal_load(w8 + i), r1 = __builtin_nontemporal_load(w8 + i + stride),
                r2 = __builtin_nontemporal_load(w8 + i + 2 * stride), r3 = __builtin_nontemporal_load(w8 + i + 3 * stride);
    widen16(r0, w16 + 2 * i);
    widen16(r1, w16 + 2 * (i + stride));
    widen16(r2, w16 + 2 * (i + 2 * stride));
    widen16(r3, w16 + 2 * (i + 3 * stride));
  }
  for (; i < n16; i += stride) widen16(__builtin_nontemporal_load(w8 + i), w16 + 2 * i);
}
static bool widen_fp8_applies(const GemmArgs& a) {
  static const bool off = sd_switch("MI355X_SD_NO_WIDEN_F8") != nullptr;   // A/B switch (tests/test_gpu_switches.py: equal to fp32 rounding, not the same bits)
  if (off) return false;
  if (!a.wscale || a.w16 || !a.ws_base || a.conv || a.rowstat || (a.K & 63) || (a.N & 3)) return false;
  if ((size_t)a.N * a.K * 2 > a.ws_bytes || (reinterpret_cast<uintptr_t>(a.W) & 15) || (reinterpret_cast<uintptr_t>(a.ws_base) & 15)) return false;
  const long tiles = (long)((a.M + 255) / 256) * ((a.N + 159) / 160);
  // (launches of <= 128 tiles may take split-K slices: plan_splitk. M >= 4096: below that -- the 1232-row context stream of SD3 at
  // bs 8 -- a launch is about as long as the widening pass itself and reads the matrix once either way)
  static const int min_m = [] {   // A/B switch of the debug build: MI355X_SD_WIDEN_F8_MIN_M
    const char* e = sd_switch("MI355X_SD_WIDEN_F8_MIN_M");
    return e ? atoi(e) : 4096;
  }();
  return tiles > 128 && a.M >= min_m;
}

int launch_gemm(const GemmArgs& a_in, hipStream_t stream) {
  if (widen_fp8_applies(a_in)) {
    const long n16 = (long)a_in.N * a_in.K / 16;   // (K % 64 == 0)
    // (grid: every lane four loads where the matrix is large enough to still give each CU two blocks)
    const long blocks1 = (n16 + 255) / 256;
    hipLaunchKernelGGL(widen_fp8_kernel, dim3((unsigned)(blocks1 >= 2048 ? (blocks1 + 3) / 4 : blocks1)), dim3(256), 0, stream,
                       reinterpret_cast<const u32x4*>(a_in.W), reinterpret_cast<u32x4*>(a_in.ws_base), n16);
    GemmArgs b = a_in;
    b.W = reinterpret_cast<const bf16*>(a_in.ws_base);
    b.w16 = 1;
    return launch_gemm(b, stream);
  }
  GemmArgs a = a_in;
  a.splitk = 0;
  a.gm = gemm_gm();
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return SD_ERR_INVALID;
  if ((a.K & 7) || (a.N & 3) || (a.lda & 7) || (a.ldc & 3)) return SD_ERR_UNSUPPORTED;
  if (a.R && (a.ldr & 3)) return SD_ERR_UNSUPPORTED;
  if (a.geglu && ((a.N & 31) || a.out_f32 || a.R || a.rowbias || a.gate)) return SD_ERR_UNSUPPORTED;
  if ((a.gate || a.rowbias) && a.rows_per_batch <= 0) return SD_ERR_INVALID;
  if ((a.a_rpb && (a.conv || (a.a_bstride & 7))) || (a.c_rpb && (a.c_bstride & 3))) return SD_ERR_UNSUPPORTED;
  if (a.conv) {
    if (a.K != 9 * a.Cin || (a.Cin & 7) || (a.stride != 1 && a.stride != 2) || (a.up != 0 && a.up != 1))
      return SD_ERR_UNSUPPORTED;
    if ((long)a.M % ((long)a.Ho * a.Wo) != 0) return SD_ERR_INVALID;
  }
  if (a.rowbias && a.rows_per_batch <= 0) return SD_ERR_INVALID;
  {   // diagnostics only: a device buffer for per-block time stamps, handed over by scripts/gemm_timeline.py as an address
    static unsigned long long* const ts = [] {
      const char* e = sd_switch("MI355X_SD_GEMM_TSTAMP");
      return e ? reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0)) : nullptr;
    }();
    a.ts = ts;
  }
  a.c_wide = !a.out_f32 && !(reinterpret_cast<uintptr_t>(a.C) & 15) && !(a.ldc & 7) && !(a.c_bstride & 7);
  static const bool epi_batch_off = sd_switch("MI355X_SD_GEMM_NO_EPI_BATCH") != nullptr;   // A/B switch (gemm_epilogue.h)
  a.epi_batch = epi_batch_off ? 0 : 1;
  a.bias_acc = 0;   // launch_gemm_pipe decides
  int tile = pick_tile(a);
  if (tile == 64) {    // 64 x 64 tile, six-stage ring (gemm_small.hip), else the 128 x 128 tiles (with split-K slices)
    const int rc = launch_gemm_small(a, stream);
    if (rc != SD_ERR_UNSUPPORTED) return rc;
    tile = 128;
  }
  if (tile == 258 || tile == 259) {   // four-wave 256 x 256 / 256 x 160 tile (gemm_w4.hip) where it applies, else the model's choice among the others
    const int rc = launch_gemm_w4(a, stream, tile == 258 ? 256 : 160);
    if (rc != SD_ERR_UNSUPPORTED) return rc;
    tile = pick_tile_model(a);
  }
  // Column groups by tile width (round 4). An XCD runs 32 consecutive tile ids per round = 32/g row-tiles x g column-tiles; what
  // its L2 pulls through the fabric per round is (32/g) A row-panels + g W column-panels, minimal at g = sqrt(32 BM / BN): 7.2 for
  // the 256x160 tile, 5.1 for 256x320. Round 2 measured g = 4 and 8 inside the step (one g for every tile family) as equal in time
  // (60.37 / 60.47 ms); the 160-wide launches -- fused QKV walks 24 column tiles -- take 8: their A panels come through the fabric
  // 3 instead of 6 times (the counter traffic of the class, not its time: the loop does not wait for L2, profiles/HISTORY.md section 5).
  if (tile == 160 || tile == 129) a.gm = -8;
  if (tile == 128) plan_splitk(a, 128, 128);
  if (a.rowstat && (a.conv || a.wscale || a.a_rpb || a.c_rpb || a.R || a.rowbias || a.gate || !a.wsum))
    return SD_ERR_UNSUPPORTED;
  {   // software-pipelined loop (gemm_pipe.hip) where it applies (128 / 160 tiles; 256 only as an experiment)
    const int rc = launch_gemm_pipe(a, tile, stream);
    if (rc != SD_ERR_UNSUPPORTED) return rc;
  }
  if (a.rowstat) {   // LayerNorm-folded projection: own kernel instantiations (epilogue in gemm_epilogue_ln)
    if (tile == 257 && !(a.K & 63)) return launch_gemm256(a, stream);
    if (tile == 256 || tile == 257) return launch_cfg<false, Cfg256, false, true>(a, stream);
    if (tile == 160 && !a.geglu) return launch_cfg<false, Cfg256x160, false, true>(a, stream);
    if (tile == 320) return a.geglu ? launch_cfg<false, Cfg256x320g, false, true>(a, stream)
                                    : launch_cfg<false, Cfg256x320, false, true>(a, stream);
    return launch_cfg<false, Cfg128, false, true>(a, stream);
  }
  if (w_is_f8(a)) {   // fp8 weight bytes read by the kernel: generic configurations only
    if (a.conv || (a.K & 15)) return SD_ERR_UNSUPPORTED;
    if (tile == 160 && !a.geglu) return launch_cfg<false, Cfg256x160, true>(a, stream);
    if (tile == 320) return a.geglu ? launch_cfg<false, Cfg256x320g, true>(a, stream) : launch_cfg<false, Cfg256x320, true>(a, stream);
    if (tile == 256 || tile == 257) return launch_cfg<false, Cfg256, true>(a, stream);
    return launch_cfg<false, Cfg128, true>(a, stream);
  }
  if (tile == 257 && !((a.K & 63) || (a.conv && (a.Cin & 63)) || a.a_rpb)) return launch_gemm256(a, stream);
  if (tile == 256 || tile == 257) return a.conv ? launch_cfg<true, Cfg256>(a, stream) : launch_cfg<false, Cfg256>(a, stream);
  if (tile == 160 && !a.geglu)
    return a.conv ? launch_cfg<true, Cfg256x160>(a, stream) : launch_cfg<false, Cfg256x160>(a, stream);
  if (tile == 320 && !a.geglu)
    return a.conv ? launch_cfg<true, Cfg256x320>(a, stream) : launch_cfg<false, Cfg256x320>(a, stream);
  if (tile == 320 && !a.conv) return launch_cfg<false, Cfg256x320g>(a, stream);
  return a.conv ? launch_cfg<true, Cfg128>(a, stream) : launch_cfg<false, Cfg128>(a, stream);
}

}  // namespace sd
