// The four-wave 256 x 256 tile (gemm_w4.hip) on e4m3 operands: W8A8 block GEMMs of the MMDiT (mi355x_sd_linear_f8 / _f8_q, BASELINE
// config 5). Round 6. Same LDS image, DMA pieces, swizzle, persistent blocks, carried prologue and instruction placement; what differs:
//
//   * a K-tile is still 128 bytes per row, i.e. 128 e4m3 elements = ONE k-step of v_mfma_scale_f32_16x16x128_f8f6f4 (32 cycles, twice
//     the work of a 16x16x32 bf16 MFMA): 64 MFMAs per wave and K-tile instead of 128, the same 32 ds_read_b128 and 16 DMA pieces;
//   * an MFMA consumes BOTH 16-byte chunks of its operand rows at once (lane l: row l & 15, k = (l >> 4) * 32 .. + 31 = chunks
//     2 (l >> 4), 2 (l >> 4) + 1), so there are no two half-sets to alternate. One fragment set (8 A + 8 W operands of 8 registers =
//     128 VGPRs) is ROLLED instead: the MFMAs of a tile walk W-major (w0 x a0..a7, w1 x a0..a7, ...); operand w_j is dead after its
//     eight MFMAs and is reloaded with tile t+1's w_j right there; a_i dies in the last column and is reloaded then (a_6, a_7 and w_7
//     in the first steps of the next tile). Every reload is issued >= 6 MFMAs (~190 cycles) ahead of its first use;
//   * one barrier per K-tile at step 5: every wave's reads of tile t have retired (the stage is free for the pieces of tile t+2, one
//     every 3 steps behind it) and its own pieces of tile t+1 have landed (nothing newer in flight: vmcnt(0)), whose reads start at
//     step 9;
//   * the epilogue is gemm_epilogue_f8's arithmetic (acc * ascale[m] * wscale[n] + bias, gate, residual, tanh-GELU, bf16 or e4m3
//     store with the safe row scale) in the lean form of gemm_w4.hip's: per-channel vectors in registers, the residual two row-tiles
//     ahead, tile-uniform batch -- bit-identical to the phased 256 x 256 kernel it replaces (tests/test_gpu_gemm_variants.py).
#include "gemm_w4_common.h"
#include "kernels.h"

namespace sd {

namespace w4 {
typedef __attribute__((ext_vector_type(8))) int i32x8;
// acc += w x a (e4m3, K = 128) with the accumulator tile in a-registers; sc = 0x7F7F7F7F: E8M0 1.0 in every scale byte
__device__ __forceinline__ void mfma8_a(f32x4& acc, const i32x8& w, const i32x8& a, const int sc) {
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+a"(acc) : "v"(w), "v"(a), "v"(sc));
}
// the first MFMA of an accumulator's chain: constant 0 as C
__device__ __forceinline__ void mfma8_a0(f32x4& acc, const i32x8& w, const i32x8& a, const int sc) {
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, 0, %3, %3 op_sel_hi:[0,0,0]" : "=a"(acc) : "v"(w), "v"(a), "v"(sc));
}
}  // namespace w4

// Epilogue (the arithmetic and its order = gemm_epilogue.h gemm_epilogue_f8; layout of the lane's channels = acc_col<8>).
template <int TM, int TN>
__device__ __forceinline__ void w4f8_epilogue(const GemmArgs& p, f32x4 (&acc)[TN][TM], int m_tile, int m_wave, int n_wave, int lane) {
  const int lq = lane >> 4;
  float as[TM], oinv[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int mc = min(m_wave + tm * 16 + (lane & 15), p.M - 1);
    as[tm] = p.ascale[mc];
    oinv[tm] = 0.f;
    if (p.out_f8) {   // safe row scale for the e4m3 output (gemm_epilogue_f8)
      const float bound = 1.1f * (p.a_l2[mc] * p.w_norm_max + p.bias_abs_max);
      oinv[tm] = 448.0f / fmaxf(bound, 1e-12f);
      if (n_wave == 0 && lq == 0 && m_wave + tm * 16 + (lane & 15) < p.M) p.oscale[mc] = fmaxf(bound, 1e-12f) * (1.0f / 448.0f);
    }
  }
  f32x4 wsc[TN], bs[TN], gs[TN];
  const bool has_bias = p.bias != nullptr, has_gate = p.gate != nullptr, has_r = p.R != nullptr;
  const float* gt = has_gate ? p.gate + (size_t)(m_tile / p.rows_per_batch) * p.ld_gate : nullptr;   // (a tile lies inside one batch: launcher)
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int nc = min(n_wave + acc_col<TN>(tn, lq, false), p.N - 4);
    wsc[tn] = *reinterpret_cast<const f32x4*>(p.wscale + nc);
    bs[tn] = has_bias ? *reinterpret_cast<const f32x4*>(p.bias + nc) : f32x4{0.f, 0.f, 0.f, 0.f};
    gs[tn] = has_gate ? *reinterpret_cast<const f32x4*>(gt + nc) : f32x4{1.f, 1.f, 1.f, 1.f};
  }
  size_t c_base = 0;   // row m of C lives at c_base + (m - c_m0) * ldc
  int c_m0 = 0;
  if (p.c_rpb) {
    const int b = m_tile / p.c_rpb;
    c_base = (size_t)b * p.c_bstride;
    c_m0 = b * p.c_rpb;
  }
  u32x4 r_cur[TN / 2], r_nxt[TN / 2];
  auto load_r = [&](const int tm, u32x4 (&r)[TN / 2]) {
    const int m = min(m_wave + tm * 16 + (lane & 15), p.M - 1);
#pragma unroll
    for (int h = 0; h < TN / 2; ++h) r[h] = *reinterpret_cast<const u32x4*>(p.R + (size_t)m * p.ldr + min(n_wave + h * 32 + lq * 8, p.N - 8));
  };
  if (has_r) load_r(0, r_nxt);
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    if (has_r) {
#pragma unroll
      for (int h = 0; h < TN / 2; ++h) r_cur[h] = r_nxt[h];
      if (tm + 1 < TM) load_r(tm + 1, r_nxt);
    }
    const int m = m_wave + tm * 16 + (lane & 15);
    if (m < p.M) {
      const size_t crow = c_base + (size_t)(m - c_m0) * p.ldc;
#pragma unroll
      for (int h = 0; h < TN / 2; ++h) {
        const int n = n_wave + h * 32 + lq * 8;
        if (n >= p.N) continue;   // (N % 8 == 0: a lane's 8 channels are inside or outside together)
        f32x4 lo = acc[2 * h][tm] * as[tm] * wsc[2 * h], hi = acc[2 * h + 1][tm] * as[tm] * wsc[2 * h + 1];
        if (has_bias) {
          lo += bs[2 * h];
          hi += bs[2 * h + 1];
        }
        if (has_gate) {
          lo *= gs[2 * h];
          hi *= gs[2 * h + 1];
        }
        if (has_r) {
          lo = add_r16(lo, u32x2{r_cur[h][0], r_cur[h][1]});
          hi = add_r16(hi, u32x2{r_cur[h][2], r_cur[h][3]});
        }
        lo = act4(p, lo);
        hi = act4(p, hi);
        if (p.out_f8) {   // 8 e4m3 bytes per lane with the row's output scale
          lo *= oinv[tm];
          hi *= oinv[tm];
          int q0 = 0, q1 = 0;
          q0 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[0], lo[1], q0, false);
          q0 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[2], lo[3], q0, true);
          q1 = __builtin_amdgcn_cvt_pk_fp8_f32(hi[0], hi[1], q1, false);
          q1 = __builtin_amdgcn_cvt_pk_fp8_f32(hi[2], hi[3], q1, true);
          *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned char*>(p.C) + crow + n) = u32x2{(unsigned)q0, (unsigned)q1};
        } else {
          const u32x4 pk = {pack_bf16(lo[0], lo[1]), pack_bf16(lo[2], lo[3]), pack_bf16(hi[0], hi[1]), pack_bf16(hi[2], hi[3])};
          *reinterpret_cast<u32x4*>(reinterpret_cast<bf16*>(p.C) + crow + n) = pk;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // one row-tile at a time (gemm_w4.hip)
  }
}

// BS: step of the iteration that carries the barrier (behind the reads of a_6, a_7, w_7 of the tile itself in steps 0 .. 3);
// DSP: steps between two DMA pieces
template <int BS, int DSP, bool CARRY = true>
__global__ __launch_bounds__(256, 1) void gemm_w4f8_kernel(const GemmArgs p) {
  using namespace w4;
  constexpr int BM = 256, BN = 256, TM = 8, TN = 8, NW = 4, AP = 8, WP = 8, KB = 128;   // KB: bytes (= e4m3 elements) of a K-tile row
  constexpr int STAGE_A = BM * KB, STAGE_W = BN * KB;                                    // 32 KiB each
  constexpr int NSTEP = TM * TN;                                                         // one MFMA per step
  static_assert(BS >= 4 && BS < 8 && BS + 1 + (AP + WP - 1) * DSP < NSTEP, "barrier behind the tile's own reads; 16 pieces inside the iteration");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;                 // [2][BM][128 B]
  unsigned char* Ws = smem + 2 * STAGE_A;   // [2][BN][128 B]

  const int ntn = (p.N + BN - 1) / BN;
  const int ntm = (p.M + BM - 1) / BM;
  const int nvb = ntm * ntn;
  const int nt = p.K / KB;   // K % 128 == 0, >= 3 tiles (launcher)
  const unsigned char* A8 = reinterpret_cast<const unsigned char*>(p.A);
  const unsigned char* W8 = reinterpret_cast<const unsigned char*>(p.W);
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(A8), 0, 0xFFFFFFE0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(W8), 0, (unsigned)((size_t)p.N * p.K), 0x00020000);

  constexpr unsigned OOB = 0xFFFFFFF0u;
  unsigned a_off[AP], w_off[WP];
  int m0 = 0, n0 = 0;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  auto place_tile = [&](const int vb, int& tm0, int& tn0) {
    const int lane = lane_id();
    int tile_m, tile_n;
    tile_coords(xcd_remap(vb, nvb), ntm, ntn, p.gm, tile_m, tile_n);
    tm0 = tile_m * BM;
    tn0 = tile_n * BN;
    const int sub = lane >> 3, cg = (lane & 7) ^ sub;
    size_t a_base = 0;   // row m of A lives at a_base + (m - a_m0) * lda (a tile lies inside one batch of remapped rows: launcher)
    int a_m0 = 0;
    if (p.a_rpb) {
      const int b = tm0 / p.a_rpb;
      a_base = (size_t)b * p.a_bstride;
      a_m0 = b * p.a_rpb;
    }
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int m = tm0 + (wave + i * NW) * 8 + sub;
      a_off[i] = m < p.M ? (unsigned)(a_base + (size_t)(m - a_m0) * p.lda + cg * 16) : OOB;
    }
#pragma unroll
    for (int i = 0; i < WP; ++i) {
      const int n = tn0 + w_row_of_lds_row<TN>((wave + i * NW) * 8 + sub, false);
      w_off[i] = (n < p.N) ? (unsigned)((size_t)n * p.K + cg * 16) : OOB;
    }
  };
  auto issue_piece = [&](auto jc, const int stage, const int kb) {
    constexpr int j = decltype(jc)::value;
    if constexpr (j < AP) dma(a_rsrc, As + stage * STAGE_A + (wave + j * NW) * 1024, a_off[j], kb);
    else dma(w_rsrc, Ws + stage * STAGE_W + (wave + (j - AP) * NW) * 1024, w_off[j - AP], kb);
  };
  auto issue_prologue = [&]() {
    static_for<0, AP + WP>([&](auto jc) { issue_piece(jc, 0, 0); });
    static_for<0, AP + WP>([&](auto jc) { issue_piece(jc, 1, KB); });
  };

  int vb = blockIdx.x;
  place_tile(vb, m0, n0);
  issue_prologue();
  bool carried = false;
  while (true) {
    const int lane = lane_id();
    const int wm = wave >> 1, wn = wave & 1;
    f32x4 acc[TN][TM];
    const int frow = lane & 15, fkc = lane >> 4, rsw = frow & 7;
    const int a_row = (wm * (TM * 16) + frow) * 128, w_row = (wn * (TN * 16) + frow) * 128;
    const int c0 = ((2 * fkc) ^ rsw) << 4, c1 = ((2 * fkc + 1) ^ rsw) << 4;
    int sc = 0x7F7F7F7F;
    asm volatile("" : "+v"(sc));
    bf16x8 fa[TM][2], fw[TN][2];   // [.][0], [.][1]: the two 16-byte chunks = the 32 operand bytes of one MFMA (8 consecutive registers)
    auto read_a = [&](auto ic, const int stage) {
      constexpr int i = decltype(ic)::value;
      const unsigned char* r = As + stage * STAGE_A + a_row + i * 16 * 128;
      fa[i][0] = *reinterpret_cast<const bf16x8*>(r + c0);
      fa[i][1] = *reinterpret_cast<const bf16x8*>(r + c1);
    };
    auto read_w = [&](auto ic, const int stage) {
      constexpr int i = decltype(ic)::value;
      const unsigned char* r = Ws + stage * STAGE_W + w_row + i * 16 * 128;
      fw[i][0] = *reinterpret_cast<const bf16x8*>(r + c0);
      fw[i][1] = *reinterpret_cast<const bf16x8*>(r + c1);
    };
    // K-tile 0 landed (first tile of the block: K-tile 1 may still fly; a carried prologue sits in front of the previous tile's
    // stores in the in-order counter: gemm_w4.hip); then the whole fragment set of K-tile 0
    if (carried) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AP + WP) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, TN>([&](auto ic) { read_w(ic, 0); });
    static_for<0, TM>([&](auto ic) { read_a(ic, 0); });

    // iteration of tile t (stage cur = t & 1). DMA: the pieces of tile t+2 go out behind the barrier; NEXT: there is a tile t+1 (its
    // operands replace this tile's as they die); FIRST: t = 0 (the accumulators' chains start; the set was read in one burst above)
    auto iter = [&](auto dmac, auto nextc, auto firstc, const int cur, const int kb2) {
      constexpr int DMA = decltype(dmac)::value, NEXT = decltype(nextc)::value, FIRST = decltype(firstc)::value;
      static_for<0, NSTEP>([&](auto stc) {
        constexpr int s = decltype(stc)::value, j = s / TM, i = s % TM;
        // this tile's last three operands (their registers were busy until the previous tile's last MFMAs)
        if constexpr (!FIRST && s == 0) read_a(ic_t<TM - 2>{}, cur);
        if constexpr (!FIRST && s == 1) read_a(ic_t<TM - 1>{}, cur);
        if constexpr (!FIRST && s == 2) read_w(ic_t<TN - 1>{}, cur);
        if constexpr (s == BS) {
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
        }
        if constexpr (DMA && s > BS && (s - BS - 1) % DSP == 0 && (s - BS - 1) / DSP < AP + WP) issue_piece(ic_t<(s - BS - 1) / DSP>{}, cur, kb2);
        // tile t+1: w_j one step behind its last MFMA (s = 8 j + 7), a_i behind the last column's MFMA on it (s = 56 + i)
        if constexpr (NEXT && j >= 1 && i == 1) read_w(ic_t<j - 1>{}, cur ^ 1);
        if constexpr (NEXT && j == TN - 1 && i >= 2) read_a(ic_t<i - 2>{}, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        {
          i32x8 wv, av;
          __builtin_memcpy(&wv, &fw[j][0], 32);
          __builtin_memcpy(&av, &fa[i][0], 32);
          if constexpr (FIRST) mfma8_a0(acc[j][i], wv, av, sc);
          else mfma8_a(acc[j][i], wv, av, sc);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    iter(ic_t<1>{}, ic_t<1>{}, ic_t<1>{}, 0, 2 * KB);
    int t = 1;
    for (; t < nt - 2; ++t) iter(ic_t<1>{}, ic_t<1>{}, ic_t<0>{}, t & 1, (t + 2) * KB);
    iter(ic_t<0>{}, ic_t<1>{}, ic_t<0>{}, t & 1, 0);
    ++t;
    iter(ic_t<0>{}, ic_t<0>{}, ic_t<0>{}, t & 1, 0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

    const int em0 = m0, en0 = n0;
    const int vb_next = vb + (int)gridDim.x;
    const bool has_next = vb_next < nvb;
    const int lane_e = lane_id();
    const int m_w = em0 + wm * (TM * 16), n_w = en0 + wn * (TN * 16);
    // the next tile's prologue ahead of this tile's stores -- except where the epilogue has ordinary loads to wait for behind it in the
    // in-order counter (a residual: gemm_w4.hip)
    const bool carry_now = CARRY && has_next && !p.R;
    if (carry_now) {
      place_tile(vb_next, m0, n0);
      issue_prologue();
      __builtin_amdgcn_sched_barrier(0);
    }
    w4f8_epilogue<TM, TN>(p, acc, em0, m_w, n_w, lane_e);
    if (!has_next) break;
    vb = vb_next;
    carried = carry_now;
    if (!carry_now) {
      place_tile(vb, m0, n0);
      issue_prologue();
    }
  }
}

// what the e4m3 four-wave tile takes: the launches launch_gemm_f8 admits (16-byte aligned rows, K % 128 == 0, ...) with at least three
// K-tiles, N % 8 == 0, 16-byte bf16 stores or 8-byte e4m3 stores, a 16-bit residual with 16-byte rows, and gate / row remaps whose
// rows-per-batch is a multiple of the tile height (one batch per tile)
bool gemm_w4f8_applies(const GemmArgs& a) {
  if (!a.ascale || !a.wscale || a.conv || a.geglu || a.out_f32 || a.rowbias || a.rowstat || a.silu || a.splitk > 1) return false;
  if ((a.K & 127) || a.K < 384 || (a.N & 7) || (a.lda & 15)) return false;
  if (!a.out_f8 && !a.c_wide) return false;
  if (a.out_f8 && ((a.ldc & 7) || (reinterpret_cast<uintptr_t>(a.C) & 7) || a.c_rpb || a.R || a.gate)) return false;
  if (a.R && (a.r_f32 || (a.ldr & 7) || (reinterpret_cast<uintptr_t>(a.R) & 15))) return false;
  if ((a.a_rpb && ((a.a_rpb & 255) || (a.a_bstride & 15))) || (a.c_rpb && (a.c_rpb & 255)) ||
      (a.gate && (a.rows_per_batch <= 0 || (a.rows_per_batch & 255))))
    return false;
  const size_t lim = 0xFFFF0000ull;
  const size_t a_ext = a.a_rpb ? (size_t)((a.M - 1) / a.a_rpb) * a.a_bstride + (size_t)(a.a_rpb - 1) * a.lda + a.K : (size_t)(a.M - 1) * a.lda + a.K;
  return a_ext < lim && (size_t)a.N * a.K < lim;
}

int launch_gemm_w4f8(const GemmArgs& a, hipStream_t stream) {
  if (!gemm_w4f8_applies(a)) return SD_ERR_UNSUPPORTED;
  constexpr int LDS_BYTES = 2 * (256 + 256) * 128;
  using K = void (*)(const GemmArgs);
  K kern = (K)gemm_w4f8_kernel<5, 3>;
  static const bool attr_ok =
      hipFuncSetAttribute(reinterpret_cast<const void*>((K)gemm_w4f8_kernel<5, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) == hipSuccess;
  if (!attr_ok) return SD_ERR_HIP;
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  const int tiles = ((a.M + 255) / 256) * ((a.N + 255) / 256);
  hipLaunchKernelGGL(kern, dim3(tiles < cus ? tiles : cus), dim3(256), LDS_BYTES, stream, a);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

}  // namespace sd
