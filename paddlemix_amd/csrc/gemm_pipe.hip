// Software-pipelined bf16 MFMA GEMM / implicit-GEMM conv3x3 for gfx950: the loops of the 128x128, 128x160, 256x160 tiles
// (register-pipelined, interleaved) and of the 256x256 / 256x320 tiles (streaming).
//
// Same tiles, LDS image (1-KiB LDS-DMA pieces, 128-B rows, 16-B chunk index XOR (row & 7)), MFMA issue order and epilogues as
// gemm.hip (the generic fall-back loop). Common to both loops:
//   * buffer (SRD) addressing: 32-bit per-lane offsets computed once + a scalar K offset; rows >= M / N and the conv padding are
//     an out-of-range offset that the hardware zero-fills (no zero page, no selects). Needs K % 64 == 0 and operands below 4 GiB;
//     launch_gemm falls back to gemm.hip otherwise;
//   * one barrier per K-tile, counted vmcnt waits (the newest staged tile stays in flight), persistent blocks for launches of more
//     tiles than the chip holds blocks, bias as the accumulators' initial value, optional early fetch of the residual rows.
// History of the loop (profiles/): round 1 -- two fragment register sets, three LDS stages, compiler-visible s_waitcnt; round 2 --
// loader waves (gone in round 4: the interleaved loop beats them everywhere, r04_s2_step_shapes_ab.txt); round 3 -- persistent
// blocks, early residual, bias in the accumulators; round 4 -- the INTERLEAVED register-pipelined loop below.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "gemm_cfg.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace sd {

#define SD_PIPE_BARRIER()                 \
  do {                                    \
    __builtin_amdgcn_sched_barrier(0);    \
    __builtin_amdgcn_s_barrier();         \
    __builtin_amdgcn_sched_barrier(0);    \
  } while (0)

// compile-time loop: f(std::integral_constant<int, I>{}) for I = B .. E-1 (every index a constant expression: register arrays stay
// in registers, and `if constexpr` inside f places an instruction at exactly one step of an unrolled schedule)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}
template <int V>
using ic_t = std::integral_constant<int, V>;
// step (of nstep) at which the i-th of np LDS-DMA pieces of a half-iteration is issued: spread, none at step 0. (Round 4 measured
// three other placements -- reads front-loaded / pieces last, steps of four MFMAs, pieces first: all within +-0.6 % of this one
// inside the step, profiles/r04_s3_step_ab.txt; gone.)
constexpr int il_piece_step(int i, int np, int nstep) { return 1 + (i * (nstep - 1)) / np; }

// 16-byte-per-lane LDS-DMA from a buffer resource. A plain (non-template) function on purpose: called with
// type-dependent arguments straight from the kernel template, the builtin makes the host pass of hipcc drop the kernel's
// instantiation without a diagnostic (the device pass is fine), leaving the stub symbol undefined at load time.
__device__ __forceinline__ void dma(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds, unsigned voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)lds, 16, voff, soff, 0, 0);
}

// PRE: the epilogue's residual / bias operands are fetched during the last K iterations (gemm_epilogue.h EpiPre). Its own
// instantiation, not a run-time branch: two alternative consumers of the accumulators make the register allocator split their
// live ranges and spill inside the K loop (header of gemm_epilogue.h).
template <bool CONV, class CFG, bool LN, bool PRE, bool WS = false>
__device__ __forceinline__ void gemm_pipe_body(const GemmArgs& p) {
  static_assert(!PRE || (!LN && (CFG::TM + CFG::TN) <= 10), "early epilogue operands: register-pipelined tiles");
  constexpr int BM = CFG::BM, BN = CFG::BN, TM = CFG::TM, TN = CFG::TN, ST = CFG::STAGES, NW = CFG::NW;
  constexpr int PW = NW;                                 // every wave owns LDS-DMA pieces
  constexpr int AP = (CFG::A_TOTAL + PW - 1) / PW, WP = (CFG::W_TOTAL + PW - 1) / PW;
  constexpr int STAGE_A = BM * BK * 2, STAGE_W = BN * BK * 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;                 // [ST][BM][128 B]
  unsigned char* Ws = smem + ST * STAGE_A;  // [ST][BN][128 B]

  const int ntn = (p.N + BN - 1) / BN;
  const int ntm = (p.M + BM - 1) / BM;
  // Persistent blocks (round 3): launches of more tiles than the chip holds blocks start only that many blocks, block b walks the
  // tiles b, b + gridDim.x, ... -- the tiles the dispatcher would have handed it anyway (same XCD, same L2 neighbourhood), without
  // the ~3.4 us it takes to re-dispatch 256 blocks x 8 waves at every round boundary (scripts/gemm_timeline.py: FF1's second round
  // starts 45 us after the first although a tile takes 41). The only cost: one barrier between the K loop and the epilogue, so
  // that no wave's next prologue DMA overwrites a stage another wave still reads; a wave that is done with its stores goes
  // straight on to the next tile's prologue while slower waves are still storing.
  const int nvb = ntm * ntn;
  const bool persist = (int)gridDim.x < nvb;             // launch-uniform
  for (int vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
  // (the thread id is re-read through an opaque asm per tile: otherwise every per-lane constant of the prologue is hoisted out of
  // this loop and kept alive across the K loop for the next tile -- 256 VGPRs and spills at the 256x320 tiles)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // timeline diagnostics (p.ts != NULL only under scripts/gemm_timeline.py): thread 0 of each block stamps the 100-MHz wall clock
  auto stamp = [&](const int slot) {
    if (p.ts && tid == 0) p.ts[(size_t)(blockIdx.y * nvb + vb) * 6 + slot] = wall_clock64();
  };
  const int pw = wave;                                   // piece-owner index of this wave
  const int wm = wave / CFG::WAVES_N, wn = wave % CFG::WAVES_N;
  stamp(0);
  const int lid = xcd_remap(vb, nvb);
  int tile_m, tile_n;
  tile_coords(lid, ntm, ntn, p.gm, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int nt_all = p.K / BK;   // K % 64 == 0 on this path
  int t0 = 0, t1 = nt_all;       // k-tile range of this block (split-K: blockIdx.y picks the slice)
  if (p.splitk > 1) {
    t0 = blockIdx.y * p.kc;
    t1 = min(nt_all, t0 + p.kc);
  }

  // ---- LDS-DMA geometry: piece q = wave + i*NW (8 rows x 128 B); lane -> row q*8 + (lane>>3), 16-B chunk (lane&7)^row ----
  const int sub = lane >> 3;
  const int cg = (lane & 7) ^ sub;
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.A), 0, 0xFFFFFFE0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.W), 0, (unsigned)((size_t)p.N * p.K * 2), 0x00020000);
  unsigned a_off[AP], w_off[WP];
  int oy[AP], ox[AP];
  bool a_ok[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int m = m0 + (pw + i * PW) * 8 + sub;
    a_ok[i] = m < p.M;
    const int mm = a_ok[i] ? m : 0;
    if (CONV) {
      const int hw = p.Ho * p.Wo;
      const int b = mm / hw;
      const int rem = mm - b * hw;
      oy[i] = rem / p.Wo;
      ox[i] = rem - oy[i] * p.Wo;
      a_off[i] = (unsigned)((size_t)b * p.Hs * p.Ws * p.lda * 2);   // byte offset of the row's image
    } else {
      const size_t arow = p.a_rpb ? (size_t)(mm / p.a_rpb) * p.a_bstride + (size_t)(mm % p.a_rpb) * p.lda : (size_t)mm * p.lda;
      a_off[i] = a_ok[i] ? (unsigned)((arow + cg * 8) * 2) : OOB;
      oy[i] = ox[i] = 0;
    }
  }
#pragma unroll
  for (int i = 0; i < WP; ++i) {
    const int n = n0 + w_row_of_lds_row<TN>((pw + i * PW) * 8 + sub, p.geglu);   // epilogue-friendly channel order
    w_off[i] = (n < p.N) ? (unsigned)(((size_t)n * p.K + cg * 8) * 2) : OOB;
  }
  int gtap = 0, gcch = t0 * BK + cg * 8;   // conv: running (tap, channel) of this lane's chunk
  if (CONV) conv_k_init(p.kb64, t0, cg * 8, p.Cin, gtap, gcch);
  int kiss = t0 * BK;   // K offset of the next tile to stage

  auto issue_tile = [&](const int stage) {
    unsigned char* a = As + stage * STAGE_A + pw * 1024;
    unsigned char* w = Ws + stage * STAGE_W + pw * 1024;
    if (CONV) {
      const int ky = gtap / 3, kx = gtap - ky * 3;
      const int Hin = p.Hs << p.up, Win = p.Ws << p.up;
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const int iy = oy[i] * p.stride + ky - p.pad;
        const int ix = ox[i] * p.stride + kx - p.pad;
        const bool ok = a_ok[i] && (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win;
        const unsigned off = a_off[i] + (unsigned)(((iy >> p.up) * p.Ws + (ix >> p.up)) * p.lda + gcch) * 2u;
        if (CFG::A_TOTAL % PW == 0 || pw + i * PW < CFG::A_TOTAL) dma(a_rsrc, a + i * (PW * 1024), ok ? off : OOB, 0);
      }
      conv_k_next(p.kb64, p.Cin, gtap, gcch);
    } else {
#pragma unroll
      for (int i = 0; i < AP; ++i)
        if (CFG::A_TOTAL % PW == 0 || pw + i * PW < CFG::A_TOTAL) dma(a_rsrc, a + i * (PW * 1024), a_off[i], kiss * 2);
    }
#pragma unroll
    for (int i = 0; i < WP; ++i)
      if (CFG::W_TOTAL % PW == 0 || pw + i * PW < CFG::W_TOTAL) dma(w_rsrc, w + i * (PW * 1024), w_off[i], kiss * 2);
    kiss += BK;
  };
  // ---- piece-level issue (interleaved loop): A piece i / W piece i of this wave; kA / kW = K offset of the A / W tile being staged ----
  [[maybe_unused]] int kA = t0 * BK, kW = t0 * BK;
  [[maybe_unused]] auto issue_a = [&](auto ic, const int stage) {
    constexpr int i = decltype(ic)::value;
    if ((i + 1) * PW <= CFG::A_TOTAL || pw + i * PW < CFG::A_TOTAL) {   // (only a ragged last piece is a run-time question)
      unsigned char* a = As + stage * STAGE_A + (pw + i * PW) * 1024;
      if (CONV) {
        const int ky = gtap / 3, kx = gtap - ky * 3;
        const int Hin = p.Hs << p.up, Win = p.Ws << p.up;
        const int iy = oy[i] * p.stride + ky - p.pad;
        const int ix = ox[i] * p.stride + kx - p.pad;
        const bool ok = a_ok[i] && (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win;
        const unsigned off = a_off[i] + (unsigned)(((iy >> p.up) * p.Ws + (ix >> p.up)) * p.lda + gcch) * 2u;
        dma(a_rsrc, a, ok ? off : OOB, 0);
      } else {
        dma(a_rsrc, a, a_off[i], kA * 2);
      }
    }
  };
  [[maybe_unused]] auto next_a = [&]() {
    kA += BK;
    if (CONV) conv_k_next(p.kb64, p.Cin, gtap, gcch);
  };
  [[maybe_unused]] auto issue_w = [&](auto ic, const int stage) {
    constexpr int i = decltype(ic)::value;
    if ((i + 1) * PW <= CFG::W_TOTAL || pw + i * PW < CFG::W_TOTAL) dma(w_rsrc, Ws + stage * STAGE_W + (pw + i * PW) * 1024, w_off[i], kW * 2);
  };
  [[maybe_unused]] auto next_w = [&]() { kW += BK; };
  // s_waitcnt vmcnt(BASE [+ the pieces of one tile]). Where the pieces do not divide evenly over the waves (256 x 160: 20 W pieces on
  // 8 waves) the count is the SMALLER one for every wave: exact for the waves that own fewer pieces; the others also wait for
  // the oldest piece of the newer tile, issued most of an iteration earlier -- cheaper than the branch per K-tile that picks the
  // immediate by wave (s_waitcnt takes no register operand).
  [[maybe_unused]] auto wait_newer = [&](auto basec, auto newerc) {
    constexpr int BASE = decltype(basec)::value;
    constexpr int PMIN = CFG::A_TOTAL / PW + CFG::W_TOTAL / PW;
    wait_vmcnt_imm<BASE + (decltype(newerc)::value ? PMIN : 0)>();
  };

  constexpr bool STREAM = (TM + TN) > 10;   // two full fragment sets would not fit next to the accumulators
  // The accumulators start at the bias of their channel (GemmArgs::bias_acc: plain bf16-weight launches without split-K): the
  // loads go out here, ahead of the first LDS-DMA, and are consumed after the prologue's wait -- the epilogue then has no bias
  // load (one dependent L2 round trip per 4-channel group before, gemm_epilogue.h). Channels past N read a clamped address; they
  // are never stored.
  f32x4 acc[TN][TM];
  f32x4 bias4[TN];
  if (!LN && p.bias_acc) {
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int nb = min(n0 + wn * (TN * 16) + acc_col<TN>(i, lane >> 4, p.geglu), p.N - 4);
      bias4[i] = *reinterpret_cast<const f32x4*>(p.bias + nb);
      // (weight-only fp8 on a widened matrix: the epilogue multiplies the accumulators by the channel's weight scale, so the
      // initial value is bias / scale -- fp32, the scale is absmax / 448 > 0)
      if (p.wscale) bias4[i] /= *reinterpret_cast<const f32x4*>(p.wscale + nb);
    }
  } else {
#pragma unroll
    for (int i = 0; i < TN; ++i) bias4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  auto init_acc = [&]() {
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = bias4[i];
  };

  // early fetch of the epilogue's residual (gemm_epilogue.h EpiPre): plain bf16 residual launches of at least TM + 2 K-tiles
  // (launch_pipe decides). The loads land in registers the compiler does not manage (gemm_epilogue.h tells why); it neither
  // counts nor waits for them: the counted waits of the loop include them and gemm_epilogue_pre starts with its own wait.
  const int m_pre = m0 + wm * (TM * 16), n_pre = n0 + wn * (TN * 16);
  [[maybe_unused]] const u32x4 r_srd = PRE ? epi_r_srd(p) : u32x4{0u, 0u, 0u, 0u};

  const int frow = lane & 15, fkc = lane >> 4, rsw = frow & 7;
  const int a_row = (wm * (TM * 16) + frow) * 128, w_row = (wn * (TN * 16) + frow) * 128;
  const int c0 = ((0 * 4 + fkc) ^ rsw) << 4, c1 = ((1 * 4 + fkc) ^ rsw) << 4;

  if constexpr (!STREAM) {
  // ---- interleaved loop, register-pipelined tiles (round 4) ----
  // The round-3 loop issued a K-tile's 6-7 LDS-DMA pieces as one burst and its 9 fragment reads as two bursts; both waves of a SIMD do
  // so at the same moment (one barrier per K-tile keeps them in step), and an LDS-DMA piece costs an in-order wave 60-180 cycles of
  // issue (MI355X_MICROARCH.md constants) in which it feeds the matrix pipe nothing: in-loop 0.47-0.55 of the MFMA rate on the
  // 256x160 tile (scripts/gemm_timeline.py, 1.15 us per K-tile for 0.53 us of MFMA work; 0.92 us now). A half-iteration is NSTEP steps of
  //     [one ds_read_b128 of the OTHER fragment set] [at most one LDS-DMA piece] [G MFMAs]
  // fenced with sched_barrier(0) so that this is the emitted order: the non-MFMA issue slots sit in the shadow of MFMAs of the same
  // wave or of its SIMD partner, and the two waves of a SIMD drift apart by themselves instead of colliding on the VMEM port.
  // The DMA work of a tile is split over TWO half-iterations: the W pieces of tile t+AHEAD go out in half 0 of iteration t, the A
  // pieces of tile t+1+AHEAD in half 1 of iteration t -- into the stage of tile t itself, which the mid barrier of iteration t has
  // just proven fully read (every wave retires its reads of tile t, lgkmcnt(0), before that barrier; half 1 multiplies fragments
  // that are in registers already). RAW as before: a wave waits for its own pieces of tile t+1 (counted vmcnt: the A and W pieces
  // of tile t+2 -- and the early residual loads of iterations t-1 and t -- may stay in flight) before the mid barrier of
  // iteration t; every read of tile t+1 comes after that barrier.
  // The last iterations are UNROLLED (template R = tiles after this one): what is issued and the wait counts are compile-time
  // facts there, and the main loop has neither the "is there a next tile" conditionals nor the run-time vmcnt switch
  // (wait_vmcnt_dyn: ~40 of the 155 scalar instructions per K-tile of the round-3 PRE loop).
  constexpr int AHEAD = ST - 1, NM = TM * TN, NR = TM + TN, G = 2, NSTEP = (NM + G - 1) / G;
  static_assert(NR <= NSTEP && AP < NSTEP && WP < NSTEP, "one read and at most one DMA piece per step");
  constexpr int LPR = EpiPre<(PRE ? TM : 1), (PRE ? TN : 1)>::LOADS_PER_ROW;
  constexpr int NTAIL = (PRE && TM > AHEAD ? TM : AHEAD) + 1;
  bf16x8 fa[2][TM], fw[2][TN];
  // read k of a fragment set, in the order the MFMAs consume them: w0, a0 .. a(TM-1), w1 .. w(TN-1)
  auto read_one = [&](auto setc, auto kc, const int stage) {
    constexpr int set = decltype(setc)::value, k = decltype(kc)::value;
    if constexpr (k == 0 || k > TM) {
      constexpr int i = k == 0 ? 0 : k - TM;
      fw[set][i] = *reinterpret_cast<const bf16x8*>(Ws + stage * STAGE_W + w_row + (set ? c1 : c0) + i * 16 * 128);
    } else {
      fa[set][k - 1] = *reinterpret_cast<const bf16x8*>(As + stage * STAGE_A + a_row + (set ? c1 : c0) + (k - 1) * 16 * 128);
    }
  };
  // one half-iteration. SET: the fragment set the MFMAs consume (the reads fill the other one from rstage); DK: 0 no DMA, 1 this
  // wave's W pieces, 2 its A pieces (-> dstage), +2: only if dma_on (run-time, wave-uniform); PR: residual row-tile requested at
  // the last step (-1: none, -2: row pr_rt, run-time)
  auto half = [&](auto setc, auto rdc, auto dkc, auto prc, const int rstage, const int dstage, const bool dma_on, const int pr_rt) {
    constexpr int SET = decltype(setc)::value, RD = decltype(rdc)::value, DK = decltype(dkc)::value, PR = decltype(prc)::value;
    static_for<0, NSTEP>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if constexpr (RD && k < NR) read_one(ic_t<SET ^ 1>{}, kc, rstage);
      if constexpr (DK == 1 || DK == 3) static_for<0, WP>([&](auto ic) {
        if constexpr (il_piece_step(decltype(ic)::value, WP, NSTEP) == k) {
          if (DK == 1 || dma_on) issue_w(ic, dstage);
        }
      });
      if constexpr (DK == 2 || DK == 4) static_for<0, AP>([&](auto ic) {
        if constexpr (il_piece_step(decltype(ic)::value, AP, NSTEP) == k) {
          if (DK == 2 || dma_on) issue_a(ic, dstage);
        }
      });
      if constexpr (PRE && k == NSTEP - 1) {
        if constexpr (PR >= 0) {
          epi_prefetch_row<(PR >= 0 ? PR : 0), TM, TN>(p, r_srd, m_pre, n_pre, lane);
        } else if constexpr (PR == -2) {
          static_assert(!PRE || TM <= 4, "four row-tiles enumerated");
          if (pr_rt == 0) epi_prefetch_row<0, TM, TN>(p, r_srd, m_pre, n_pre, lane);
          if constexpr (TM > 1) if (pr_rt == 1) epi_prefetch_row<(TM > 1 ? 1 : 0), TM, TN>(p, r_srd, m_pre, n_pre, lane);
          if constexpr (TM > 2) if (pr_rt == 2) epi_prefetch_row<(TM > 2 ? 2 : 0), TM, TN>(p, r_srd, m_pre, n_pre, lane);
          if constexpr (TM > 3) if (pr_rt == 3) epi_prefetch_row<(TM > 3 ? 3 : 0), TM, TN>(p, r_srd, m_pre, n_pre, lane);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, G>([&](auto gc) {
        constexpr int q = k * G + decltype(gc)::value;
        if constexpr (q < NM) acc[q / TM][q % TM] = mfma_16x16x32(fw[SET][q / TM], fa[SET][q % TM], acc[q / TM][q % TM]);
      });
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  int stage = 0, s1 = 1, sw = ST - 1;   // stages of tiles t, t+1, t+AHEAD (rotated after every iteration)
  auto rotate = [&]() {
    const int s0 = stage;
    stage = s1;
    if constexpr (ST == 3) {
      s1 = sw;
      sw = s0;
    } else {
      s1 = sw = s0;
    }
  };
  // iteration of tile t; R = tiles after it when that is a compile-time fact (unrolled tail), -1: at least NTAIL
  auto iter = [&](auto rc) {
    constexpr int R = decltype(rc)::value;
    constexpr bool has1 = R != 0, hasW = R < 0 || R >= AHEAD, hasA = R < 0 || R >= AHEAD + 1;
    constexpr int PR = (PRE && R >= 1 && R <= TM) ? TM - R : -1;
    half(ic_t<0>{}, ic_t<1>{}, ic_t<(hasW ? 1 : 0)>{}, ic_t<PR>{}, stage, sw, true, 0);
    if constexpr (hasW) next_w();
    __builtin_amdgcn_s_waitcnt(0xC07F);          // every read of tile t retired
    if constexpr (has1) {
      // tile t+1 must have landed; issued after it: (three stages) the residual loads of iteration t-1, the A and W pieces of tile
      // t+2, this iteration's residual loads; (two stages) only this iteration's residual loads
      constexpr int rcur = PR >= 0 ? LPR : 0, rprev = (PRE && R >= 0 && R + 1 <= TM) ? LPR : 0;
      wait_newer(ic_t<(AHEAD == 2 ? rprev + rcur : rcur)>{}, ic_t<(AHEAD == 2 && (R < 0 || R >= 2) ? 1 : 0)>{});
      SD_PIPE_BARRIER();                         // publishes tile t+1; every wave is done reading tile t
    }
    half(ic_t<1>{}, ic_t<(has1 ? 1 : 0)>{}, ic_t<(hasA ? 2 : 0)>{}, ic_t<-1>{}, s1, stage, true, 0);
    if constexpr (hasA) next_a();
    rotate();
  };
  // the same with R = TM .. 1 at run time: the early-residual kernels walk their last iterations in a LOOP. Unrolled, the register
  // allocator renames accumulators between the copies (out-of-place MFMA destinations) and took registers above v215 -- the
  // landing zone of the residual rows, which amdgpu_num_vgpr(216) budgets but does not reserve: the rows were overwritten (NaN on
  // the first hardware run; scripts/check_landing_zone.py now checks every build). In a loop the accumulators are loop-carried
  // and stay where they are. Cost: three small wave-uniform selections per K-tile in these TM iterations only.
  [[maybe_unused]] auto iter_rt = [&](const int R) {
    const bool hasW = R >= AHEAD, hasA = R >= AHEAD + 1;
    half(ic_t<0>{}, ic_t<1>{}, ic_t<3>{}, ic_t<-2>{}, stage, sw, hasW, TM - R);
    if (hasW) next_w();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if constexpr (AHEAD == 2) {   // (the counts of iter<R> above)
      if (R == TM) wait_newer(ic_t<LPR>{}, ic_t<1>{});
      else if (R >= 2) wait_newer(ic_t<2 * LPR>{}, ic_t<1>{});
      else wait_newer(ic_t<2 * LPR>{}, ic_t<0>{});
    } else {
      wait_newer(ic_t<LPR>{}, ic_t<0>{});
    }
    SD_PIPE_BARRIER();
    half(ic_t<1>{}, ic_t<1>{}, ic_t<4>{}, ic_t<-1>{}, s1, stage, hasA, 0);
    if (hasA) next_a();
    rotate();
  };

  const int nt = t1 - t0;
  // prologue: tile t0 (and t0+1 with three stages), then what "half 1 of iteration t0-1" would have issued: the A pieces of t0+AHEAD
  static_for<0, AP>([&](auto ic) { issue_a(ic, 0); });
  static_for<0, WP>([&](auto ic) { issue_w(ic, 0); });
  next_a();
  next_w();
  constexpr int PA = CFG::A_TOTAL / PW, PT = CFG::A_TOTAL / PW + CFG::W_TOTAL / PW;   // pieces every wave owns (wait_newer)
  if (AHEAD == 2 && nt > 1) {
    static_for<0, AP>([&](auto ic) { issue_a(ic, 1); });
    static_for<0, WP>([&](auto ic) { issue_w(ic, 1); });
    next_a();
    next_w();
  }
  if (nt > AHEAD) {
    static_for<0, AP>([&](auto ic) { issue_a(ic, AHEAD); });
    next_a();
  }
  // tile t0 landed; what was issued after it may stay in flight
  if (nt > AHEAD) wait_vmcnt_imm<(AHEAD == 2 ? PT + PA : PA)>();
  else if (AHEAD == 2 && nt > 1) wait_vmcnt_imm<PT>();
  else wait_vmcnt_imm<0>();
  SD_PIPE_BARRIER();
  stamp(1);
  init_acc();
  static_for<0, NR>([&](auto kc) { read_one(ic_t<0>{}, kc, 0); });
  int nrem = nt;
  for (; nrem > NTAIL; --nrem) iter(ic_t<-1>{});
  if constexpr (PRE) {   // (these launches have at least TM + 2 > NTAIL K-tiles: launch_pipe)
    static_assert(!PRE || (TM >= AHEAD && NTAIL == TM + 1), "tail of the early-residual kernels");
#pragma nounroll
    for (int R = TM; R >= 1; --R) iter_rt(R);
    iter(ic_t<0>{});
  } else {
    static_for<0, NTAIL>([&](auto jc) {
      constexpr int R = NTAIL - 1 - decltype(jc)::value;
      if (nrem > R) iter(ic_t<R>{});
    });
  }
  } else {
  // ---- streaming variant (256x320 tiles: 160 accumulator registers) ----
  // The smaller operand of a k-step is HELD (two sets, k-step 0 / 1), the larger one STREAMS through a (Q+1)-slot
  // register queue: step j of the 2*SN steps of a K-tile multiplies one streamed fragment with the HN held ones
  // (HN MFMAs) and issues the ds_read of the fragment of step j+Q, so an LDS read has Q*HN MFMAs (>= 192 cycles) to land.
  // Two LDS stages: the DMA of tile t+1 is issued at the top of iteration t and awaited Q steps before its end (85 % of
  // an iteration of lead); there one barrier publishes tile t+1 and proves tile t fully read, after which the queue and
  // the held set roll over into tile t+1 without a bubble. (Round 4 built the interleaved form of this loop too -- one LDS-DMA
  // piece per step instead of the burst at the top: no gain on FF1, 2 % slower convs, profiles/r04_s2_step_shapes_ab.txt; with
  // 80 MFMAs per wave and K-tile the burst is a small share here. Gone.)
  static_assert(ST == 2, "streaming loop is written for two LDS stages");
  constexpr bool HOLD_A = TM <= TN;
  constexpr int HN = HOLD_A ? TM : TN, SN = HOLD_A ? TN : TM, Q = 3, QN = Q + 1, STEPS = 2 * SN;
  const int h_row = HOLD_A ? a_row : w_row, s_row = HOLD_A ? w_row : a_row;
  bf16x8 hold[2][HN], qf[QN];
  auto read_hold = [&](const int set, const int stage) {
    const unsigned char* b = (HOLD_A ? As + stage * STAGE_A : Ws + stage * STAGE_W) + h_row + (set ? c1 : c0);
#pragma unroll
    for (int i = 0; i < HN; ++i) hold[set][i] = *reinterpret_cast<const bf16x8*>(b + i * 16 * 128);
  };
  auto read_stream = [&](const int slot, const int stage, const int ks, const int s) {
    const unsigned char* b = (HOLD_A ? Ws + stage * STAGE_W : As + stage * STAGE_A) + s_row + (ks ? c1 : c0);
    qf[slot] = *reinterpret_cast<const bf16x8*>(b + s * 16 * 128);
  };

  issue_tile(0);
  wait_vmcnt_imm<0>();
  SD_PIPE_BARRIER();
  stamp(1);
  init_acc();
  read_hold(0, 0);
#pragma unroll
  for (int j = 0; j < Q; ++j) read_stream(j % QN, 0, j / SN, j % SN);
  for (int t = t0; t < t1; ++t) {
    const int cur = (t - t0) & 1, nxt = cur ^ 1;
    const bool more = t + 1 < t1;
    if (more) issue_tile(nxt);   // stage of tile t-1: every wave passed the roll-over barrier of iteration t-1
#pragma unroll
    for (int j = 0; j < STEPS; ++j) {
      const int ks = j / SN, s = j % SN;
      if (j == 0) read_hold(1, cur);
      if (j == STEPS - Q && more) {
        wait_vmcnt_imm<0>();                     // own pieces of tile t+1 (issued STEPS - Q steps ago)
        __builtin_amdgcn_s_waitcnt(0xC07F);     // every read of tile t retired
        SD_PIPE_BARRIER();
        read_hold(0, nxt);
      }
      const int jr = j + Q;
      if (jr < STEPS) read_stream(jr % QN, cur, jr / SN, jr % SN);
      else if (more) read_stream(jr % QN, nxt, (jr - STEPS) / SN, (jr - STEPS) % SN);
      __builtin_amdgcn_sched_barrier(0);   // keep the read Q steps ahead of its use (the scheduler would sink it)
#pragma unroll
      for (int h = 0; h < HN; ++h) {
        if (HOLD_A) acc[s][h] = mfma_16x16x32(qf[j % QN], hold[ks][h], acc[s][h]);
        else acc[h][s] = mfma_16x16x32(hold[ks][h], qf[j % QN], acc[h][s]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  }

  if (persist) SD_PIPE_BARRIER();   // every wave has read its last fragments: the next tile's prologue may overwrite the stages
  const int m_w = m0 + wm * (TM * 16), n_w = n0 + wn * (TN * 16);
  if (p.ts) {   // (the stamp must not be taken before the accumulators are final: touch one)
    asm volatile("" ::"v"(acc[TN - 1][TM - 1][0]));
    stamp(2);
  }
  if (p.splitk > 1) {   // raw partial sums -> ws[split][m][n]; the epilogue runs in splitk_reduce_kernel (gemm.hip)
    float* ws = p.ws + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int m = m_w + tm * 16 + (lane & 15);
      if (m >= p.M) continue;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int n = n_w + acc_col<TN>(tn, lane >> 4, p.geglu);
        if (n < p.N) *reinterpret_cast<f32x4*>(ws + (size_t)m * p.N + n) = acc[tn][tm];
      }
    }
    return;
  }
  if constexpr (LN) {
    gemm_epilogue_ln<TM, TN>(p, acc, m_w, n_w, lane);
  } else if constexpr (PRE) {
    gemm_epilogue_pre<TM, TN>(p, acc, m_w, n_w, lane);
  } else {
    gemm_epilogue<TM, TN, WS>(p, acc, m_w, n_w, lane);
  }
  if (p.ts) {
    stamp(3);                                   // stores issued (not yet drained)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(4);                                   // this wave's stores written back
    if (tid == 0) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      p.ts[(size_t)(blockIdx.y * nvb + vb) * 6 + 5] = ((unsigned long long)(xcc & 15) << 32) | (unsigned)vb;
    }
  }
  }   // tiles of this block
}

template <bool CONV, class CFG, bool LN>
__global__ __launch_bounds__(CFG::THREADS, CFG::MIN_WAVES) void gemm_pipe_kernel(const GemmArgs p) {
  gemm_pipe_body<CONV, CFG, LN, false>(p);
}
// launches with a per-channel weight scale (a just-in-time widened e4m3 matrix): the scale vectors live in registers in the
// epilogue (gemm_epilogue.h WS) -- own instantiations so that no other kernel's register allocation sees them
template <class CFG>
__global__ __launch_bounds__(CFG::THREADS, CFG::MIN_WAVES) void gemm_pipe_ws_kernel(const GemmArgs p) {
  gemm_pipe_body<false, CFG, false, false, true>(p);
}
// the early-residual form: the compiler is budgeted v0 .. v215, v216 .. v255 are the landing zone of the asm loads
// (gemm_epilogue.h EpiPre; the attribute takes no template-dependent argument, hence the second entry point). The budget is not
// a reservation: scripts/check_landing_zone.py verifies on every build that no compiler-generated instruction names v216+.
template <bool CONV, class CFG>
__global__ __launch_bounds__(CFG::THREADS, CFG::MIN_WAVES) __attribute__((amdgpu_num_vgpr(216))) void gemm_pipe_pre_kernel(const GemmArgs p) {
  gemm_pipe_body<CONV, CFG, false, true>(p);
}

// Grid of a launch: one block per tile, capped at what the chip holds at once (persistent blocks, see the kernel) when there are
// more tiles than that. MI355X_SD_GEMM_PERSIST=0: one block per tile always (A/B switch, tests/test_gpu_gemm_variants.py).
template <class CFG>
static int pipe_grid_x(int tiles, int ny) {
  static const bool off = [] {
    const char* e = sd_switch("MI355X_SD_GEMM_PERSIST");
    return e && atoi(e) == 0;
  }();
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  // blocks per CU by LDS and registers: the eight-wave tiles own a CU, the four-wave ones share it
  constexpr int bpc = CFG::NW == 8 ? 1 : 2;
  const int cap = cus * bpc;
  return (off || ny > 1 || tiles <= cap) ? tiles : cap;
}

// early residual fetch (gemm_pipe_pre_kernel) applies: plain bf16 residual launches of the register-pipelined tiles
template <bool CONV, class CFG, bool LN>
static bool pre_applies(const GemmArgs& a) {
  if constexpr (!CONV && !LN && (CFG::TM + CFG::TN) <= 10 && CFG::TM <= 4) {
    static const bool pre_off = sd_switch("MI355X_SD_GEMM_NO_PRE") != nullptr;   // A/B switch
    return !pre_off && a.R && !a.r_f32 && !a.geglu && !a.gate && !a.wscale && a.splitk <= 1 && (!a.bias || a.bias_acc) &&
           a.K / BK >= CFG::TM + 2 && !(a.N & 7) && ((size_t)(a.M - 1) * a.ldr + a.N) * 2 < 0xFFFF0000ull;
  } else {
    return false;
  }
}

template <bool CONV, class CFG, bool LN>
static int launch_pipe(const GemmArgs& a, hipStream_t stream) {
  const int ntm = (a.M + CFG::BM - 1) / CFG::BM, ntn = (a.N + CFG::BN - 1) / CFG::BN;
  const int ny = a.splitk > 1 ? a.splitk : 1;
  if constexpr (!CONV && !LN && (CFG::TM + CFG::TN) <= 10 && CFG::TM <= 4) {
    // residual launches (to_out, proj_out, FF2): the residual is fetched during the last TM K iterations. bf16 residual rows
    // addressed with 32-bit offsets, N % 8 == 0 (a 16-byte pair load never straddles the row's end), bias in the accumulators, no
    // GEGLU / gate / fp8 scale / split-K (those epilogues live elsewhere); not the implicit-GEMM convs (their gather state leaves no
    // room for the 40 landing registers)
    if (pre_applies<CONV, CFG, LN>(a)) {
      static const bool pre_ok = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pipe_pre_kernel<CONV, CFG>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES) == hipSuccess;
      }();
      if (!pre_ok) return SD_ERR_HIP;
      hipLaunchKernelGGL((gemm_pipe_pre_kernel<CONV, CFG>), dim3(pipe_grid_x<CFG>(ntm * ntn, ny), ny), dim3(CFG::THREADS), CFG::LDS_BYTES, stream, a);
      return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
    }
  }
  if constexpr (!CONV && !LN) {
    if (a.wscale && a.splitk <= 1) {   // (a.w16: launch_gemm_pipe turned e4m3 weight bytes away)
      static const bool ws_ok = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pipe_ws_kernel<CFG>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES) == hipSuccess;
      }();
      if (!ws_ok) return SD_ERR_HIP;
      hipLaunchKernelGGL((gemm_pipe_ws_kernel<CFG>), dim3(pipe_grid_x<CFG>(ntm * ntn, ny), ny), dim3(CFG::THREADS), CFG::LDS_BYTES, stream, a);
      return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
    }
  }
  static const bool attr_ok = [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pipe_kernel<CONV, CFG, LN>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES) == hipSuccess;
  }();
  if (!attr_ok) return SD_ERR_HIP;
  hipLaunchKernelGGL((gemm_pipe_kernel<CONV, CFG, LN>), dim3(pipe_grid_x<CFG>(ntm * ntn, ny), ny), dim3(CFG::THREADS), CFG::LDS_BYTES, stream, a);
  if (a.splitk > 1) launch_splitk_reduce(a, stream);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

// tile: 128 | 160 (pick_tile ids). Returns SD_ERR_UNSUPPORTED when the fast path does not apply (caller falls back).
int launch_gemm_pipe(const GemmArgs& a_in, int tile, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GemmArgs a = a_in;
  static const bool bias_acc_off = sd_switch("MI355X_SD_GEMM_NO_BIAS_ACC") != nullptr;   // A/B switch
  a.bias_acc = (a.bias && (!a.wscale || a.w16) && a.splitk <= 1 && !a.rowstat && !bias_acc_off) ? 1 : 0;
  static const bool off = sd_switch("MI355X_SD_NO_PIPE") != nullptr;     // every launch on the generic loop of gemm.hip (the variant test's reference)
  if (off || (a.wscale && !a.w16) || (a.K & 63) || (tile == 160 && a.geglu)) return SD_ERR_UNSUPPORTED;   // (e4m3 weight bytes: gemm.hip)
  if (tile == 320 && a.conv && a.geglu) return SD_ERR_UNSUPPORTED;
  // 256x256: the phased kernel (gemm256.hip) stays the default where it can run (id 257); id 256 is what pick_tile
  // returns when it cannot (A row remap of the MMDiT output projections) and takes the pipelined loop here
  if (tile != 128 && tile != 160 && tile != 320 && tile != 256 && tile != 129) return SD_ERR_UNSUPPORTED;
  if (a.geglu && tile == 129) return SD_ERR_UNSUPPORTED;   // odd number of 16-column sub-tiles per wave
  if (a.conv && (a.Cin & 7)) return SD_ERR_UNSUPPORTED;
  // 32-bit buffer offsets: every addressed byte of A and W must sit below 4 GiB - 64 KiB
  const size_t lim = 0xFFFF0000ull;
  size_t a_ext;
  if (a.conv) a_ext = (size_t)(a.M / ((size_t)a.Ho * a.Wo)) * a.Hs * a.Ws * a.lda * 2;
  else if (a.a_rpb) a_ext = ((size_t)((a.M - 1) / a.a_rpb) * a.a_bstride + (size_t)(a.a_rpb - 1) * a.lda + a.K) * 2;
  else a_ext = ((size_t)(a.M - 1) * a.lda + a.K) * 2;
  if (a_ext >= lim || (size_t)a.N * a.K * 2 >= lim) return SD_ERR_UNSUPPORTED;
  const bool ln = a.rowstat != nullptr;
  if (tile == 256) {
    if (ln) return launch_pipe<false, Cfg256, true>(a, stream);
    return a.conv ? launch_pipe<true, Cfg256, false>(a, stream) : launch_pipe<false, Cfg256, false>(a, stream);
  }
  if (tile == 320) {
    if (a.geglu) return ln ? launch_pipe<false, Cfg256x320g, true>(a, stream) : launch_pipe<false, Cfg256x320g, false>(a, stream);
    if (ln) return launch_pipe<false, Cfg256x320, true>(a, stream);
    return a.conv ? launch_pipe<true, Cfg256x320, false>(a, stream) : launch_pipe<false, Cfg256x320, false>(a, stream);
  }
  if (tile == 129 && !ln) return a.conv ? launch_pipe<true, Cfg128x160, false>(a, stream) : launch_pipe<false, Cfg128x160, false>(a, stream);
  if (tile == 129) return SD_ERR_UNSUPPORTED;
  if (tile == 160) {
    if (ln) return launch_pipe<false, Cfg256x160s3, true>(a, stream);
    return a.conv ? launch_pipe<true, Cfg256x160s3, false>(a, stream) : launch_pipe<false, Cfg256x160s3, false>(a, stream);
  }
  if (ln) return launch_pipe<false, Cfg128, true>(a, stream);
  return a.conv ? launch_pipe<true, Cfg128, false>(a, stream) : launch_pipe<false, Cfg128, false>(a, stream);
}

}  // namespace sd
