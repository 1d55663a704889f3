// 256 x 256 x 64 GEMM tile on FOUR waves -- one wave per SIMD, 128 x 128 outputs per wave, the 64 accumulator tiles of a wave pinned in
// the 256 accumulation registers (a0 .. a255) -- for the wide linear launches of the step (fused QKV 8192 x 3840 x 1280, FF1
// 8192 x 10240 x 1280 and 32768 x 5120 x 640 with the GEGLU epilogue, the MMDiT's 33 k-row projections). Round 6.
//
// Why. The same-box vendor yardstick, sustained and interleaved (profiles/r06_s1_blas_yardstick.txt), put hipBLASLt 35 % ahead of the
// eight-wave tiles of gemm_pipe.hip on exactly these shapes (fused QKV 63.6 vs 85.7 us, 8192 x 10240 x 1280 160.8 vs 219.4 us) and
// level or behind on every N <= 1920 shape. The code object it runs there (disassembled for STRUCTURE only) is a 256 x 256 x 64 tile
// on four waves of 512 registers: 128 MFMAs (16x16x32), 32 ds_read_b128 and 16 LDS-DMA pieces per wave and K-tile, every non-MFMA
// instruction placed between two MFMAs of the same wave, two LDS stages, fragment sets read one 32-deep k-step ahead. What that
// shape buys over 8 waves x (64 x 80): half the LDS fragment bytes per MFMA (a wave re-uses each fragment 8 times instead of 4-5),
// 1.6x the MFMA work per tile prologue / epilogue / barrier, no co-resident wave competing for a SIMD's issue slots. Round 1 had
// tried this geometry twice (profiles/r01_gemm_tiles.txt: compiler-allocated accumulators shuttled through v_accvgpr moves; then
// pinned, but with the K-tile's 16 DMA pieces and 32 reads issued as bursts) and lost to the 8-wave tiles; what it lacked is the
// placement gemm_pipe.hip learnt in round 4: ONE read and at most ONE LDS-DMA piece between every pair of MFMAs.
//
// This kernel (own code, same operand layouts and epilogues as the rest of the library):
//   * LDS image, DMA pieces, swizzle, "swapped" MFMA issue (a lane owns consecutive output channels) and epilogue = gemm_pipe.hip;
//   * accumulators: inline-asm MFMAs with "+a" operands (the only way to keep hipcc from moving them through VGPRs);
//   * a K-tile is 64 steps of [read][DMA piece][2 MFMAs] (sched_barrier-fenced): k-step 0 multiplies fragment set 0 while set 1 of the
//     same tile is read; ONE barrier per K-tile after those reads retire (every wave is then done with the stage of tile t, and its
//     own pieces of tile t+1 have landed: vmcnt(0), nothing newer is in flight); behind it the 16 pieces of tile t+2 go out into the
//     stage just freed, one per DSP steps, and k-step 1 multiplies set 1 while set 0 of tile t+1 is read;
//   * two LDS stages of 64 KiB; the last two iterations of a tile are separate instantiations without DMA / without the reads
//     of a next tile; persistent blocks walk the tiles of a launch (the next tile's prologue needs no extra barrier: after the last
//     barrier of a tile no wave reads LDS again).
#include <type_traits>

#include "common.h"
#include "gemm_cfg.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace sd {

namespace w4 {
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}
template <int V>
using ic_t = std::integral_constant<int, V>;

__device__ __forceinline__ void dma(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds, unsigned voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)lds, 16, voff, soff, 0, 0);
}
// lane index from the execution mask (v_mbcnt), through an opaque asm: recomputed wherever it is needed instead of keeping the
// thread id -- or anything derived from it -- alive across the K loop (kept alive it was spilled, and a scratch reload next to the
// epilogue's stores or the carried prologue is a vmcnt(0))
__device__ __forceinline__ int lane_id() {
  int l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  asm volatile("" : "+v"(l));
  return l;
}
// acc += w x a on the matrix pipe with the accumulator tile in a-registers
__device__ __forceinline__ void mfma_a(f32x4& acc, const bf16x8& w, const bf16x8& a) {
#ifdef MI355X_SD_F16
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
#else
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
#endif
}
// the same with the constant 0 as C operand: the first MFMA of an accumulator's chain (no initialisation pass over 256 registers)
__device__ __forceinline__ void mfma_a0(f32x4& acc, const bf16x8& w, const bf16x8& a) {
#ifdef MI355X_SD_F16
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(acc) : "v"(w), "v"(a));
#else
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(w), "v"(a));
#endif
}
}  // namespace w4

// BS: step of k-step 0 that carries the barrier (>= 16: the reads of set 1 are issued in steps 0 .. 15); DSP: steps between two DMA pieces
// Epilogue of the four-wave tile: bias (from registers) / GEGLU / out_scale / SiLU / tanh-GELU, 16-byte stores of the build's 16-bit type,
// the operands gemm_w4_applies admits, nothing else. Its own function rather than gemm_epilogue<8, 8>:
// the general form (every operand kind of every launch, two output types) came out as 15 k instructions here and made the register
// allocator SPILL accumulator tiles to scratch inside the epilogue -- and every scratch reload is a vmcnt(0) behind the stores issued
// before it, one write-back round trip each: 24 us per 128-KB tile (profiles/r06_s3_w4_ablation.txt: the launch without its
// epilogue ran in 55.8 us, with it in 104.1 us).
// Two halves: the lane's TN bias vectors are fetched AND awaited first (the asm operands make the compiler place its vmcnt(0) here,
// while nothing else is in flight), then the caller may put the next tile's prologue DMA on its way, then the stores.
template <int TN>
__device__ __forceinline__ void w4_load_bias(const GemmArgs& p, f32x4 (&bs)[TN], int n_wave, int lane) {
  const int lq = lane >> 4;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
    bs[tn] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + min(n_wave + acc_col<TN>(tn, lq, p.geglu), p.N - 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) asm volatile("" : "+v"(bs[tn]));
}
// WS instantiations (a just-in-time widened e4m3 matrix: weight-only fp8, BASELINE config 5): the per-output-channel weight scale,
// fetched like the bias; v = acc * scale + bias, the order of gemm_epilogue.h epi4
template <int TN>
__device__ __forceinline__ void w4_load_wscale(const GemmArgs& p, f32x4 (&ws)[TN], int n_wave, int lane) {
  const int lq = lane >> 4;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) ws[tn] = *reinterpret_cast<const f32x4*>(p.wscale + min(n_wave + acc_col<TN>(tn, lq, false), p.N - 4));
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) asm volatile("" : "+v"(ws[tn]));
}
template <int TM, int TN, bool WS>
__device__ __forceinline__ void w4_epilogue(const GemmArgs& p, f32x4 (&acc)[TN][TM], const f32x4 (&bs)[TN], const f32x4 (&ws)[WS ? TN : 1],
                                            int m_wave, int n_wave, int lane) {
  const int lq = lane >> 4;
  bf16* C = reinterpret_cast<bf16*>(p.C);
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int m = m_wave + tm * 16 + (lane & 15);
    if (m >= p.M) continue;
    const size_t crow = (size_t)m * p.ldc;
    if (TN % 4 == 0 && p.geglu) {   // (the launcher admits GEGLU on the 256-wide tile only)
#pragma unroll
      for (int hp = 0; hp < TN / 4; ++hp) {   // (gemm_epilogue.h store_row: two output pairs = 8 consecutive output channels)
        const int n_phys = n_wave + acc_col<TN>(4 * hp, lq, true);
        if (n_phys >= p.N) continue;
        const f32x4 lo = geglu4(acc[4 * hp][tm] + bs[4 * hp], acc[4 * hp + 1][tm] + bs[4 * hp + 1]);
        const f32x4 hi = geglu4(acc[4 * hp + 2][tm] + bs[4 * hp + 2], acc[4 * hp + 3][tm] + bs[4 * hp + 3]);
        const u32x4 pk = {pack_bf16(lo[0], lo[1]), pack_bf16(lo[2], lo[3]), pack_bf16(hi[0], hi[1]), pack_bf16(hi[2], hi[3])};
        *reinterpret_cast<u32x4*>(C + crow + (n_wave >> 1) + hp * 32 + lq * 8) = pk;
      }
    } else {
#pragma unroll
      for (int h = 0; h < TN / 2; ++h) {
        const int n = n_wave + h * 32 + lq * 8;
        if (n >= p.N) continue;   // (N % 8 == 0: a lane's 8 channels are inside or outside together)
        f32x4 a0 = acc[2 * h][tm], a1 = acc[2 * h + 1][tm];
        if constexpr (WS) {
          a0 *= ws[2 * h];
          a1 *= ws[2 * h + 1];
        }
        const f32x4 lo = act4(p, (a0 + bs[2 * h]) * p.out_scale), hi = act4(p, (a1 + bs[2 * h + 1]) * p.out_scale);
        const u32x4 pk = {pack_bf16(lo[0], lo[1]), pack_bf16(lo[2], lo[3]), pack_bf16(hi[0], hi[1]), pack_bf16(hi[2], hi[3])};
        *reinterpret_cast<u32x4*>(C + crow + n) = pk;
      }
      if constexpr ((TN & 1) != 0) {   // (the 160-wide tile: the fifth sub-tile has no partner -- 4 channels per lane, one 8-byte store)
        const int n = n_wave + (TN - 1) * 16 + lq * 4;
        if (n < p.N) {
          f32x4 a0 = acc[TN - 1][tm];
          if constexpr (WS) a0 *= ws[TN - 1];
          const f32x4 lo = act4(p, (a0 + bs[TN - 1]) * p.out_scale);
          *reinterpret_cast<u32x2*>(C + crow + n) = u32x2{pack_bf16(lo[0], lo[1]), pack_bf16(lo[2], lo[3])};
        }
      }
    }
    // one row-tile at a time: left to itself the scheduler reads all 256 accumulator registers up front (256 VGPRs, spills)
    __builtin_amdgcn_sched_barrier(0);
  }
}

// EX: the epilogue of the MMDiT block GEMMs (mi355x_sd_linear_ex): out = R + gate[batch][n] * (acc + bias) (adaLN-Zero gated residuals,
// PPD/models/attention.py:181-196) or (acc + bias + R) * out_scale, C rows remapped into / out of the joint [B, S_img + S_txt, .] buffer.
// The batch of a tile is ONE number (the launcher admits remaps and gates only when their rows-per-batch is a multiple of the 256-row
// tile): no per-row division. The residual rows of the NEXT row-tile are requested before the current one is finished (two row-tiles of
// 16-byte loads in flight); the gate vectors sit in registers next to the bias.
template <int TM, int TN, bool WS>
__device__ __forceinline__ void w4_epilogue_ex(const GemmArgs& p, f32x4 (&acc)[TN][TM], const f32x4 (&bs)[TN], const f32x4 (&ws)[WS ? TN : 1],
                                               int m_tile, int m_wave, int n_wave, int lane) {
  const int lq = lane >> 4;
  bf16* C = reinterpret_cast<bf16*>(p.C);
  f32x4 gs[TN];
  const bool has_gate = p.gate != nullptr;
  if (has_gate) {
    const float* gt = p.gate + (size_t)(m_tile / p.rows_per_batch) * p.ld_gate;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) gs[tn] = *reinterpret_cast<const f32x4*>(gt + min(n_wave + acc_col<TN>(tn, lq, false), p.N - 4));
  } else {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) gs[tn] = f32x4{1.f, 1.f, 1.f, 1.f};
  }
  // row m of C lives at c_base + (m - c_m0) * ldc
  size_t c_base = 0;
  int c_m0 = 0;
  if (p.c_rpb) {
    const int b = m_tile / p.c_rpb;
    c_base = (size_t)b * p.c_bstride;
    c_m0 = b * p.c_rpb;
  }
  const bool has_r = p.R != nullptr;
  u32x4 r_cur[TN / 2], r_nxt[TN / 2];
  u32x2 rl_cur = {0u, 0u}, rl_nxt = {0u, 0u};   // (odd TN: the last sub-tile's 4 channels)
  auto load_r = [&](const int tm, u32x4 (&r)[TN / 2], u32x2& rl) {
    const int m = min(m_wave + tm * 16 + (lane & 15), p.M - 1);
#pragma unroll
    for (int h = 0; h < TN / 2; ++h) r[h] = *reinterpret_cast<const u32x4*>(p.R + (size_t)m * p.ldr + min(n_wave + h * 32 + lq * 8, p.N - 8));
    if constexpr ((TN & 1) != 0) rl = *reinterpret_cast<const u32x2*>(p.R + (size_t)m * p.ldr + min(n_wave + (TN - 1) * 16 + lq * 4, p.N - 4));
  };
  if (has_r) load_r(0, r_nxt, rl_nxt);
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    if (has_r) {
#pragma unroll
      for (int h = 0; h < TN / 2; ++h) r_cur[h] = r_nxt[h];
      rl_cur = rl_nxt;
      if (tm + 1 < TM) load_r(tm + 1, r_nxt, rl_nxt);
    }
    const int m = m_wave + tm * 16 + (lane & 15);
    if (m < p.M) {
      const size_t crow = c_base + (size_t)(m - c_m0) * p.ldc;
#pragma unroll
      for (int h = 0; h < TN / 2; ++h) {
        const int n = n_wave + h * 32 + lq * 8;
        if (n >= p.N) continue;
        f32x4 a0 = acc[2 * h][tm], a1 = acc[2 * h + 1][tm];
        if constexpr (WS) {   // (the order of gemm_epilogue.h epi4: scale, bias, gate -- bit-identical to the other kernels)
          a0 *= ws[2 * h];
          a1 *= ws[2 * h + 1];
        }
        f32x4 lo = (a0 + bs[2 * h]) * gs[2 * h], hi = (a1 + bs[2 * h + 1]) * gs[2 * h + 1];
        if (has_r) {
          lo = add_r16(lo, u32x2{r_cur[h][0], r_cur[h][1]});
          hi = add_r16(hi, u32x2{r_cur[h][2], r_cur[h][3]});
        }
        lo = act4(p, lo * p.out_scale);
        hi = act4(p, hi * p.out_scale);
        const u32x4 pk = {pack_bf16(lo[0], lo[1]), pack_bf16(lo[2], lo[3]), pack_bf16(hi[0], hi[1]), pack_bf16(hi[2], hi[3])};
        *reinterpret_cast<u32x4*>(C + crow + n) = pk;
      }
      if constexpr ((TN & 1) != 0) {
        const int n = n_wave + (TN - 1) * 16 + lq * 4;
        if (n < p.N) {
          f32x4 a0 = acc[TN - 1][tm];
          if constexpr (WS) a0 *= ws[TN - 1];
          f32x4 lo = (a0 + bs[TN - 1]) * gs[TN - 1];
          if (has_r) lo = add_r16(lo, rl_cur);
          lo = act4(p, lo * p.out_scale);
          *reinterpret_cast<u32x2*>(C + crow + n) = u32x2{pack_bf16(lo[0], lo[1]), pack_bf16(lo[2], lo[3])};
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ABL (debug build only, timing ablations with WRONG results): 1 no DMA in the loop, 2 no fragment reads in the loop, 4 no barrier / waits,
// 8 no epilogue
#ifndef W4_CARRY_DEFAULT
#define W4_CARRY_DEFAULT true
#endif
// TNW: 16-column sub-tiles per wave: 8 = the 256 x 256 tile; 5 = 256 x 160 (a wave owns 128 x 80: 40 accumulator tiles, 80 MFMAs, 26 reads
// and 13 DMA pieces per K-tile) for the N = 640 / 1280 launches whose 256 x 256 tiling would leave a third of the chip idle -- measured
// 1.4 % SLOWER in the step than the eight-wave tile of that width (gemm.hip pick_tile), not in the picker, selectable in the debug build
template <int BS, int DSP, int ABL = 0, bool CARRY = W4_CARRY_DEFAULT, bool EX = false, bool WS = false, int TNW = 8>
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(const GemmArgs p) {
  using namespace w4;
  constexpr int BM = 256, TM = 8, TN = TNW, BN = 2 * TN * 16, NW = 4, AP = 8, WP = BN / 32;
  constexpr int STAGE_A = BM * BK * 2, STAGE_W = BN * BK * 2;   // 32 KiB, 32 / 20 KiB
  constexpr int NSTEP = TM * TN / 2;                             // steps of two MFMAs per k-step
  static_assert(BS >= TM + TN && BS + 1 + (AP + WP - 1) * DSP < 2 * NSTEP, "the pieces of a K-tile are issued inside the iteration");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;                 // [2][BM][128 B]
  unsigned char* Ws = smem + 2 * STAGE_A;   // [2][BN][128 B]

  const int ntn = (p.N + BN - 1) / BN;
  const int ntm = (p.M + BM - 1) / BM;
  const int nvb = ntm * ntn;
  const int nt = p.K / BK;   // K % 64 == 0 (launcher)
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.A), 0, 0xFFFFFFE0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.W), 0, (unsigned)((size_t)p.N * p.K * 2), 0x00020000);

  // tile vb: coordinates and the per-lane DMA offsets of this wave's pieces (piece q = wave + 4 i: 8 rows x 128 B; lane -> row
  // q*8 + (lane>>3), 16-byte chunk (lane&7) ^ row -- gemm_pipe.hip's LDS image). The thread id comes through an opaque asm every
  // time: lane constants hoisted out of the tile loop would be kept alive across the K loop (and spilled).
  constexpr unsigned OOB = 0xFFFFFFF0u;
  unsigned a_off[AP], w_off[WP];
  int m0 = 0, n0 = 0;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (an SGPR for the life of the block)
  auto place_tile = [&](const int vb, int& tm0, int& tn0) {
    const int lane = lane_id();
    int tile_m, tile_n;
    tile_coords(xcd_remap(vb, nvb), ntm, ntn, p.gm, tile_m, tile_n);
    tm0 = tile_m * BM;
    tn0 = tile_n * BN;
    const int sub = lane >> 3, cg = (lane & 7) ^ sub;
    size_t a_base = 0;   // row m of A lives at a_base + (m - a_m0) * lda (EX: a tile lies inside one batch of the remapped rows)
    int a_m0 = 0;
    if constexpr (EX) {
      if (p.a_rpb) {
        const int b = tm0 / p.a_rpb;
        a_base = (size_t)b * p.a_bstride;
        a_m0 = b * p.a_rpb;
      }
    }
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int m = tm0 + (wave + i * NW) * 8 + sub;
      a_off[i] = m < p.M ? (unsigned)((a_base + (size_t)(m - a_m0) * p.lda + cg * 8) * 2) : OOB;
    }
#pragma unroll
    for (int i = 0; i < WP; ++i) {
      const int n = tn0 + w_row_of_lds_row<TN>((wave + i * NW) * 8 + sub, p.geglu);
      w_off[i] = (n < p.N) ? (unsigned)(((size_t)n * p.K + cg * 8) * 2) : OOB;
    }
  };
  // piece j of a K-tile: j < 8 an A piece, else a W piece; kb = byte offset of the K-tile
  auto issue_piece = [&](auto jc, const int stage, const int kb) {
    constexpr int j = decltype(jc)::value;
    if constexpr (j < AP) dma(a_rsrc, As + stage * STAGE_A + (wave + j * NW) * 1024, a_off[j], kb);
    else dma(w_rsrc, Ws + stage * STAGE_W + (wave + (j - AP) * NW) * 1024, w_off[j - AP], kb);
  };
  // prologue of a tile: K-tiles 0 and 1 on their way (nothing else: the accumulators need no initialisation -- the first k-step's
  // MFMAs take the constant 0 as their C operand -- and the bias is added in the epilogue)
  auto issue_prologue = [&]() {
    static_for<0, AP + WP>([&](auto jc) { issue_piece(jc, 0, 0); });
    static_for<0, AP + WP>([&](auto jc) { issue_piece(jc, 1, BK * 2); });
  };

  int vb = blockIdx.x;
  place_tile(vb, m0, n0);
  issue_prologue();
  bool carried = false;   // this tile's prologue went out BEFORE the previous tile's stores (block-uniform)
  while (true) {
    const int lane = lane_id();
    const int wm = wave >> 1, wn = wave & 1;
    f32x4 acc[TN][TM];
    const int frow = lane & 15, fkc = lane >> 4, rsw = frow & 7;
    const int a_row = (wm * (TM * 16) + frow) * 128, w_row = (wn * (TN * 16) + frow) * 128;
    const int c0 = ((0 * 4 + fkc) ^ rsw) << 4, c1 = ((1 * 4 + fkc) ^ rsw) << 4;
    bf16x8 fa[2][TM], fw[2][TN];
    // read k of a fragment set, in the order the MFMAs consume them: w0, a0 .. a7, w1 .. w7
    auto read_one = [&](auto setc, auto kc, const int stage) {
      constexpr int set = decltype(setc)::value, k = decltype(kc)::value;
      if constexpr (k == 0 || k > TM) {
        constexpr int i = k == 0 ? 0 : k - TM;
        fw[set][i] = *reinterpret_cast<const bf16x8*>(Ws + stage * STAGE_W + w_row + (set ? c1 : c0) + i * 16 * 128);
      } else {
        fa[set][k - 1] = *reinterpret_cast<const bf16x8*>(As + stage * STAGE_A + a_row + (set ? c1 : c0) + (k - 1) * 16 * 128);
      }
    };
    // K-tile 0 landed. First tile of the block: K-tile 1 may still fly (vmcnt counts it). A carried prologue sits in FRONT of the
    // previous tile's stores in the (in-order) counter: waiting for everything is exact there, and by now cheap -- the pieces
    // landed under the epilogue, what is left is the write-back of its last stores.
    if (carried) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AP + WP) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, TM + TN>([&](auto kc) { read_one(ic_t<0>{}, kc, 0); });
    if constexpr ((ABL & 2) != 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[1][i] = fa[0][i], fw[1][i] = fw[0][i];
    }

    // iteration of tile t (stage cur = t & 1): DMA = 1: the pieces of tile t+2 go out behind the barrier; NEXT = 1: there is a tile t+1;
    // FIRST = 1: t = 0, the accumulators' chains start here. 64 steps of [read][DMA piece][2 MFMAs] (one MFMA per step with the
    // same memory instructions spread over twice as many gaps measured 4 - 8 % SLOWER, profiles/r06_s5_w4_probe.txt).
    auto iter = [&](auto dmac, auto nextc, auto firstc, const int cur, const int kb2) {
      constexpr int DMA = decltype(dmac)::value, NEXT = decltype(nextc)::value, FIRST = decltype(firstc)::value;
      static_for<0, 2 * NSTEP>([&](auto sc) {
        constexpr int s = decltype(sc)::value, ks = s / NSTEP, st = s % NSTEP;
        if constexpr (ks == 0 && st < TM + TN && !(ABL & 2)) read_one(ic_t<1>{}, ic_t<st>{}, cur);                    // set 1 of tile t
        if constexpr (ks == 1 && st < TM + TN && NEXT && !(ABL & 2)) read_one(ic_t<0>{}, ic_t<st>{}, cur ^ 1);        // set 0 of tile t+1
        if constexpr (DMA && !(ABL & 1) && s > BS && (s - BS - 1) % DSP == 0 && (s - BS - 1) / DSP < AP + WP)
          issue_piece(ic_t<(s - BS - 1) / DSP>{}, cur, kb2);
        if constexpr (s == BS && !(ABL & 4)) {
          // every read of tile t retired; own pieces of tile t+1 landed (nothing newer in flight) -> the barrier publishes tile t+1
          // and frees the stage of tile t. The LAST iteration keeps the barrier: behind it no wave reads LDS again, which is what
          // lets any wave start the next tile's prologue DMA without another one.
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 2>([&](auto gc) {
          constexpr int q = st * 2 + decltype(gc)::value;
          if constexpr (FIRST && ks == 0) mfma_a0(acc[q / TM][q % TM], fw[ks][q / TM], fa[ks][q % TM]);
          else mfma_a(acc[q / TM][q % TM], fw[ks][q / TM], fa[ks][q % TM]);
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    // (four instantiations of the K-tile: first, steady state, last but one, last; launches of one or two K-tiles take the generic
    // kernels -- gemm_w4_applies)
    iter(ic_t<1>{}, ic_t<1>{}, ic_t<1>{}, 0, 2 * BK * 2);
    int t = 1;
    for (; t < nt - 2; ++t) iter(ic_t<1>{}, ic_t<1>{}, ic_t<0>{}, t & 1, (t + 2) * BK * 2);
    iter(ic_t<0>{}, ic_t<1>{}, ic_t<0>{}, t & 1, 0);
    ++t;
    iter(ic_t<0>{}, ic_t<0>{}, ic_t<0>{}, t & 1, 0);
    // the matrix pipe's last results are read by VALU code the compiler schedules without knowing the asm above was an MFMA
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

    // the next tile's prologue goes out ahead of this tile's stores (CARRY): its pieces land under the epilogue
    const int em0 = m0, en0 = n0;
    const int vb_next = vb + (int)gridDim.x;
    const bool has_next = vb_next < nvb;
    const int lane_e = lane_id();
    const int m_w = em0 + wm * (TM * 16), n_w = en0 + wn * (TN * 16);
    f32x4 bs[TN], wsv[WS ? TN : 1];
    if constexpr (!(ABL & 8)) w4_load_bias<TN>(p, bs, n_w, lane_e);
    if constexpr (WS) w4_load_wscale<TN>(p, wsv, n_w, lane_e);
    __builtin_amdgcn_sched_barrier(0);
    // (EX launches with a residual: its loads are ordinary loads, and the compiler's waits for them would sit behind the carried DMA
    // in the in-order counter -- there the next prologue goes out after the epilogue)
    const bool carry_now = CARRY && has_next && !(EX && p.R);
    if (carry_now) {
      place_tile(vb_next, m0, n0);
      issue_prologue();
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr ((ABL & 8) != 0) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) asm volatile("" ::"a"(acc[i][j]));
    } else if constexpr (EX) {
      w4_epilogue_ex<TM, TN, WS>(p, acc, bs, wsv, em0, m_w, n_w, lane_e);
    } else {
      w4_epilogue<TM, TN, WS>(p, acc, bs, wsv, m_w, n_w, lane_e);
    }
    if (!has_next) break;
    vb = vb_next;
    carried = carry_now;
    if (!carry_now) {
      place_tile(vb, m0, n0);
      issue_prologue();
    }
  }
}

// shapes the four-wave tile takes: 16-bit operands (a just-in-time widened e4m3 matrix with its per-channel scale is one: WS
// instantiations), K % 64 == 0, no split-K / LN fold / conv / e4m3 bytes read by the kernel, and an epilogue without per-row operands
// (row bias: the general epilogue of 8-sub-tile waves fetches those one group at a time)
// EX launches (own kernel instantiation): gate / 16-bit residual / row remaps whose rows-per-batch is a multiple of the tile height
static bool w4_needs_ex(const GemmArgs& a) { return a.R || a.gate || a.a_rpb || a.c_rpb; }
bool gemm_w4_applies(const GemmArgs& a, int bn) {
  if (a.conv || a.rowstat || a.rowbias || a.out_f32 || a.splitk > 1 || !a.c_wide) return false;
  if (bn != 256 && (bn != 160 || a.geglu)) return false;   // (GEGLU pairs sub-tiles: the 160-wide tile has an odd number per wave)
  if (a.wscale && (!a.w16 || a.geglu || (a.N & 3))) return false;   // (the scale of a WIDENED matrix, in the epilogue; no GEGLU form of it)
  if (w4_needs_ex(a)) {
    // (a tile must lie inside one batch: the per-row division by a run-time rows-per-batch, hoisted out of the tile loop as VGPR
    // constants, did not survive the K loop's 256 + 256 registers -- spilled, and reloaded between the epilogue's stores)
    if (a.geglu || (a.R && (a.r_f32 || (a.ldr & 7))) || (a.a_rpb && (a.a_rpb & 255)) || (a.c_rpb && (a.c_rpb & 255)) ||
        (a.gate && (a.rows_per_batch <= 0 || (a.rows_per_batch & 255))))
      return false;
  }
  if ((a.K & 63) || a.K < 192 || (a.N & 7)) return false;   // (at least three K-tiles: first / last-but-one / last iterations)
  if (a.geglu && (a.N & 31)) return false;
  const size_t lim = 0xFFFF0000ull;
  const size_t a_ext = a.a_rpb ? ((size_t)((a.M - 1) / a.a_rpb) * a.a_bstride + (size_t)(a.a_rpb - 1) * a.lda + a.K) * 2
                               : ((size_t)(a.M - 1) * a.lda + a.K) * 2;
  return a_ext < lim && (size_t)a.N * a.K * 2 < lim;
}

int launch_gemm_w4(const GemmArgs& a_in, hipStream_t stream, int bn) {
  if (!gemm_w4_applies(a_in, bn)) return SD_ERR_UNSUPPORTED;
  GemmArgs a = a_in;
  a.bias_acc = 0;   // (this kernel adds the bias in its epilogue, from registers: the order of the other kernels' generic form)
  const int LDS_BYTES = 2 * (256 + bn) * BK * 2;
  const bool n160 = bn == 160;
  if (n160) a.gm = -8;   // (column groups of 8 for the 160-wide tiles, as launch_gemm does for the eight-wave ones)
  // (the 256-wide tile keeps the library's column groups of 4: scanned inside the SDXL step, ms per step for groups of columns 2 / 4 / 5 /
  // 8 / 16: 59.4 / 58.6 / 58.5 / 58.6 / 58.8, of rows 2 / 4 / 8: 59.1 / 59.8 / 59.8 -- profiles/r06_s34_w4_gm.txt)
  using K = void (*)(const GemmArgs);
  // (BS, DSP) of the shipped schedule; the debug build can pick another instantiation for A/B runs (MI355X_SD_W4_SCHED=0..3)
  static const int sched = [] {
    const char* e = sd_switch("MI355X_SD_W4_SCHED");
    return e ? atoi(e) : 0;
  }();
  const bool ex = w4_needs_ex(a), wsc = a.wscale != nullptr;
  K kern = ex ? (wsc ? (K)gemm_w4_kernel<18, 2, 0, W4_CARRY_DEFAULT, true, true> : (K)gemm_w4_kernel<18, 2, 0, W4_CARRY_DEFAULT, true>)
              : (wsc ? (K)gemm_w4_kernel<18, 2, 0, W4_CARRY_DEFAULT, false, true> : (K)gemm_w4_kernel<18, 2>);
  if (n160)   // (barrier at step 13 = behind the 13 reads of set 1; a piece every 2 steps: 13 pieces inside the 40 steps)
    kern = ex ? (wsc ? (K)gemm_w4_kernel<13, 2, 0, W4_CARRY_DEFAULT, true, true, 5> : (K)gemm_w4_kernel<13, 2, 0, W4_CARRY_DEFAULT, true, false, 5>)
              : (wsc ? (K)gemm_w4_kernel<13, 2, 0, W4_CARRY_DEFAULT, false, true, 5> : (K)gemm_w4_kernel<13, 2, 0, W4_CARRY_DEFAULT, false, false, 5>);
#ifdef MI355X_SD_DEBUG_SWITCHES
  if (!ex && !wsc && !n160) {
  if (sched == 1) kern = (K)gemm_w4_kernel<18, 2, 0, false>;   // the next tile's prologue behind the stores (A/B of the carried form)
  if (sched == 10) kern = (K)gemm_w4_kernel<18, 2, 1>;    // timing ablations (wrong results)
  if (sched == 11) kern = (K)gemm_w4_kernel<18, 2, 2>;
  if (sched == 12) kern = (K)gemm_w4_kernel<18, 2, 4>;
  if (sched == 13) kern = (K)gemm_w4_kernel<18, 2, 7>;
  if (sched == 14) kern = (K)gemm_w4_kernel<18, 2, 8>;
  if (sched == 15) kern = (K)gemm_w4_kernel<18, 2, 15>;
  if (sched == 16) kern = (K)gemm_w4_kernel<18, 2, 3>;
  if (sched == 17) kern = (K)gemm_w4_kernel<18, 2, 5>;
  }
#endif
  static bool attr_done[160] = {};
  const int si = ((sched >= 0 && sched < 20) ? sched : 0) + (ex ? 20 : 0) + (wsc ? 40 : 0) + (n160 ? 80 : 0);
  if (!attr_done[si]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
      return SD_ERR_HIP;
    attr_done[si] = true;
  }
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  const int tiles = ((a.M + 255) / 256) * ((a.N + bn - 1) / bn);
  hipLaunchKernelGGL(kern, dim3(tiles < cus ? tiles : cus), dim3(256), LDS_BYTES, stream, a);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

}  // namespace sd
