// Software-pipelined bf16 MFMA GEMM / implicit-GEMM conv3x3 for gfx950: the loop the 128x128 and 256x160 tiles run.
//
// Same tiles, LDS image (1-KiB LDS-DMA pieces, 128-B rows, 16-B chunk index XOR (row & 7)), MFMA issue order and
// epilogues as gemm.hip. What changes is the K loop, after reading the ISA of the generic loop: there the compiler
// places every batch of ds_read_b128 directly in front of the MFMAs that consume it and drains lgkmcnt(0) five times
// per K-tile (MFMA pipe ~36 % busy), and the DMA issue is ~150 instructions of 64-bit address selects and branches.
//   * two fragment register sets: the ds_reads of k-step 1 are issued before the MFMAs of k-step 0, and -- with
//     three LDS stages -- the k-step-0 fragments of tile t+1 before the MFMAs of k-step 1 of tile t, so every LDS
//     read has >= 16..20 MFMAs (256..320 cycles) to land and the waits become counted lgkmcnt(N);
//   * three LDS stages where they fit (256x160: 3 x 52 KiB): tile t+2 is in flight while t is multiplied, the only
//     VMEM wait is a COUNTED vmcnt (this wave's pieces of one tile), one barrier per K-tile;
//   * buffer (SRD) addressing: 32-bit per-lane offsets computed once + a scalar K offset; rows >= M / N and the conv
//     padding are an out-of-range offset that the hardware zero-fills (no zero page, no selects). Needs K % 64 == 0 and
//     operands below 4 GiB; launch_gemm falls back to gemm.hip otherwise.
// Hazards. RAW: a wave waits for its own DMA pieces of tile t+1 (vmcnt) before the mid-iteration barrier of iteration
// t; every read of tile t+1 comes after that barrier. WAR: the stage of tile t is re-staged at the top of iteration
// t+1 (as tile t+3 with three stages, t+2 with two), i.e. after the mid barrier of iteration t, in front of which every
// wave has retired its reads of tile t (lgkmcnt(0) before the barrier; the k-step-0 reads of t+1 are issued after it).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "gemm_cfg.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace sd {

// loader waves chosen by shape (K >= 4096): -0.45 ms per SDXL bs-8 step in two A/B pairs (profiles/r03_s1_step_ab.txt), after the
// isolated -10 % / -16 % on FF2 / the 11520-deep convs of round 2 (profiles/r02_gemm_loaders.txt)
constexpr int GEMM_LOADERS_DEFAULT = 0;   // round 4: none -- the interleaved loop beats them on the long-K convs too (profiles/r04_s2_*)

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
// Schedules of a half-iteration of the interleaved loop (template IL = 1 + schedule id; all give the same bits):
//   0  steps of 2 MFMAs; read k at step k; the np DMA pieces spread over the steps, none at step 0           (the default)
//   1  steps of 2 MFMAs; reads two per step (front-loaded: retired well before the barrier), the pieces in the later steps
//   2  steps of 4 MFMAs; reads two per step, pieces spread
//   3  steps of 2 MFMAs; read k at step k; the pieces in the FIRST steps (longest time to land)
constexpr int il_group(int v) { return v == 2 ? 4 : 2; }
constexpr int il_reads_per_step(int v) { return (v == 1 || v == 2) ? 2 : 1; }
constexpr int il_piece_step(int v, int i, int np, int nstep) {
  return v == 1 ? nstep - np + i : v == 3 ? i : v == 2 ? (i < nstep - 1 ? 1 + i : nstep - 1) : 1 + (i * (nstep - 1)) / np;
}

// 16-byte-per-lane LDS-DMA from a buffer resource. A plain (non-template) function on purpose: called with
// type-dependent arguments straight from the kernel template, the builtin makes the host pass of hipcc drop the kernel's
// instantiation without a diagnostic (the device pass is fine), leaving the stub symbol undefined at load time.
__device__ __forceinline__ void dma(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds, unsigned voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)lds, 16, voff, soff, 0, 0);
}

// LW > 0: LW extra LOADER waves per block issue every LDS-DMA piece; the NW compute waves never touch VMEM inside the K loop.
// Why: an in-order wave pays ~60 cycles of issue per LDS-DMA piece among bare MFMAs and 100-185 in a phase that also carries
// ds_read_b128s (MI355X_MICROARCH.md constants; measured here: 9 pieces per wave per K-tile = 1600-3300 cycles per SIMD next to
// ~2600 cycles of MFMA) -- cycles in which it cannot issue MFMAs. A loader wave has nothing else to do; one per SIMD (LW = 4)
// moves a K-tile's pieces in about the time the two compute waves of that SIMD need for its MFMAs. Same LDS image, same barriers
// (loaders take part in them), same hazards argument: the loader issues tile t+AHEAD after the barrier at which every compute wave
// retired its reads of the stage it overwrites, and waits for its pieces of tile t+1 before the barrier that publishes them.
// SG (with LW > 0): the compute waves' fragment reads are INTERLEAVED with the MFMAs of the other fragment set
// (sched_group_barrier: one ds_read_b128 per two MFMAs) instead of issued as a burst in front of them -- the read's issue slot
// then sits in the shadow of a 16-cycle MFMA, and with the DMA gone from these waves a half-iteration is straight-line code.
template <int NREAD, int NMFMA>
__device__ __forceinline__ void sgb_reads_under_mfmas() {
  constexpr int PER = NMFMA / NREAD;
#pragma unroll
  for (int i = 0; i < NREAD; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // 1 DS read
    __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);   // PER MFMAs
  }
  if constexpr (NMFMA - PER * NREAD > 0) __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - PER * NREAD, 0);
}

// PRE: the epilogue's residual / bias operands are fetched during the last K iterations (gemm_epilogue.h EpiPre). Its own
// instantiation, not a run-time branch: two alternative consumers of the accumulators make the register allocator split their
// live ranges and spill inside the K loop (header of gemm_epilogue.h).
// IL (round 4): the INTERLEAVED K loop -- every LDS-DMA piece and every fragment read sits between two small MFMA groups instead of
// in a burst at the top of the iteration; see the IL sections below. LW == 0 only.
template <bool CONV, class CFG, bool LN, int LW, int SG, bool PRE, int IL = 0>
__device__ __forceinline__ void gemm_pipe_body(const GemmArgs& p) {
  static_assert(!PRE || (LW == 0 && !LN && (CFG::TM + CFG::TN) <= 10), "early epilogue operands: register-pipelined tiles without loader waves");
  static_assert(!IL || LW == 0, "the interleaved loop has no loader waves");
  constexpr int BM = CFG::BM, BN = CFG::BN, TM = CFG::TM, TN = CFG::TN, ST = CFG::STAGES, NW = CFG::NW;
  constexpr int PW = LW ? LW : NW;                       // waves that own LDS-DMA pieces
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
  // (not in the loader-wave kernels: they live on exactly the 168 registers of three waves per SIMD and the stamp costs one)
  auto stamp = [&](const int slot) {
    if constexpr (LW == 0) {
      if (p.ts && tid == 0) p.ts[(size_t)(blockIdx.y * nvb + vb) * 6 + slot] = wall_clock64();
    }
  };
  const bool loader = LW > 0 && wave >= NW;              // wave-uniform
  const int pw = LW ? (loader ? wave - NW : 0) : wave;   // piece-owner index of this wave
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
  int mine = 0;   // LDS-DMA instructions this wave issues per K-tile (the last waves may own one piece fewer)
#pragma unroll
  for (int i = 0; i < AP; ++i) mine += (pw + i * PW < CFG::A_TOTAL) ? 1 : 0;
#pragma unroll
  for (int i = 0; i < WP; ++i) mine += (pw + i * PW < CFG::W_TOTAL) ? 1 : 0;

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
  // this wave's pieces of the newest staged tile may stay in flight, everything older has landed
  auto wait_all_but_newest = [&]() {
    constexpr int PMAX = AP + WP;
    if (mine == PMAX) wait_vmcnt_imm<PMAX>();
    else if (mine == PMAX - 1) wait_vmcnt_imm<(PMAX > 1 ? PMAX - 1 : 0)>();
    else wait_vmcnt_imm<(PMAX > 2 ? PMAX - 2 : 0)>();
  };

  // ---- piece-level issue (IL loops): A piece i / W piece i of this wave; kA / kW = K offset of the A / W tile being staged ----
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
  if (LW > 0 && loader) {
    // ---- loader waves: the K loop's DMA side, barrier for barrier what the compute waves below execute ----
    if constexpr (!STREAM) {
      constexpr int AHEAD = ST - 1;
      issue_tile(0);
      if (AHEAD == 2 && t0 + 1 < t1) {
        issue_tile(1);
        wait_all_but_newest();
      } else {
        wait_vmcnt_imm<0>();
      }
      SD_PIPE_BARRIER();
      int stage = 0;
      for (int t = t0; t < t1; ++t) {
        const int s1 = stage == ST - 1 ? 0 : stage + 1;
        const int s_new = ST == 3 ? (stage == 0 ? 2 : stage - 1) : s1;
        if (t + AHEAD < t1) issue_tile(s_new);
        if (t + 1 < t1) {
          if (AHEAD == 2 && t + 2 < t1) wait_all_but_newest();
          else wait_vmcnt_imm<0>();
          SD_PIPE_BARRIER();
        }
        stage = s1;
      }
    } else {
      issue_tile(0);
      wait_vmcnt_imm<0>();
      SD_PIPE_BARRIER();
      for (int t = t0; t < t1; ++t) {
        if (t + 1 < t1) {
          issue_tile(((t - t0) & 1) ^ 1);
          wait_vmcnt_imm<0>();
          SD_PIPE_BARRIER();
        }
      }
    }
    if (persist) SD_PIPE_BARRIER();   // (the compute waves' barrier between K loop and epilogue)
    continue;
  }

  // The accumulators start at the bias of their channel (GemmArgs::bias_acc: plain bf16-weight launches without split-K): the
  // loads go out here, ahead of the first LDS-DMA, and are consumed after the prologue's wait -- the epilogue then has no bias
  // load (one dependent L2 round trip per 4-channel group before, gemm_epilogue.h). Channels past N read a clamped address; they
  // are never stored.
  f32x4 acc[TN][TM];
  f32x4 bias4[TN];
  if (!LN && p.bias_acc) {
#pragma unroll
    for (int i = 0; i < TN; ++i)
      bias4[i] = *reinterpret_cast<const f32x4*>(p.bias + min(n0 + wn * (TN * 16) + acc_col<TN>(i, lane >> 4, p.geglu), p.N - 4));
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

  // early fetch of the epilogue's residual (gemm_epilogue.h EpiPre): the register-pipelined tiles without loader waves (those
  // are capped at 168 registers), plain bf16 residual launches of at least TM + 2 K-tiles only (launch_pipe decides).
  // Row-tile j is requested in iteration t1 - 1 - TM + j, right after that iteration's LDS-DMA. At the wait of iteration t (which
  // must see tile t+1, issued one iteration earlier in FRONT of that iteration's residual loads) the residual loads of iterations
  // t-1 and t may stay in flight next to the DMA of tile t+2: two K iterations (~2 us) for each batch to land, all of it under
  // MFMA work. The loads land in registers the compiler does not manage (gemm_epilogue.h tells why); it neither counts nor
  // waits for them: the counted waits below include them and gemm_epilogue_pre starts with its own wait.
  const int m_pre = m0 + wm * (TM * 16), n_pre = n0 + wn * (TN * 16);
  [[maybe_unused]] const u32x4 r_srd = PRE ? epi_r_srd(p) : u32x4{0u, 0u, 0u, 0u};
  [[maybe_unused]] const int t_pre = t1 - 1 - TM;   // >= t0 + 1 (launch_pipe)
  [[maybe_unused]] int rprev = 0;
  auto prefetch_rows = [&](const int t) -> int {
    if constexpr (PRE) {
      const int j = t - t_pre;
      if (j < 0 || j >= TM) return 0;
      if (j == 0) epi_prefetch_row<0, TM, TN>(p, r_srd, m_pre, n_pre, lane);
      if constexpr (TM > 1) if (j == 1) epi_prefetch_row<(TM > 1 ? 1 : 0), TM, TN>(p, r_srd, m_pre, n_pre, lane);
      if constexpr (TM > 2) if (j == 2) epi_prefetch_row<(TM > 2 ? 2 : 0), TM, TN>(p, r_srd, m_pre, n_pre, lane);
      if constexpr (TM > 3) if (j == 3) epi_prefetch_row<(TM > 3 ? 3 : 0), TM, TN>(p, r_srd, m_pre, n_pre, lane);
      static_assert(!PRE || TM <= 4, "prefetch_rows enumerates four row-tiles");
      return EpiPre<(PRE ? TM : 1), (PRE ? TN : 1)>::LOADS_PER_ROW;
    } else {
      return 0;
    }
  };

  const int frow = lane & 15, fkc = lane >> 4, rsw = frow & 7;
  const int a_row = (wm * (TM * 16) + frow) * 128, w_row = (wn * (TN * 16) + frow) * 128;
  const int c0 = ((0 * 4 + fkc) ^ rsw) << 4, c1 = ((1 * 4 + fkc) ^ rsw) << 4;

  if constexpr (IL && !STREAM) {
  // ---- interleaved loop, register-pipelined tiles (round 4) ----
  // The loop above issues a K-tile's 6-7 LDS-DMA pieces as one burst and its 9 fragment reads as two bursts; both waves of a SIMD do
  // so at the same moment (one barrier per K-tile keeps them in step), and an LDS-DMA piece costs an in-order wave 60-180 cycles of
  // issue (MI355X_MICROARCH.md constants) in which it feeds the matrix pipe nothing: in-loop 0.47-0.55 of the MFMA rate on the
  // 256x160 tile (scripts/gemm_timeline.py, 1.15 us per K-tile for 0.53 us of MFMA work). Here a half-iteration is NSTEP steps of
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
  constexpr int SV = IL - 1;   // schedule id
  constexpr int AHEAD = ST - 1, NM = TM * TN, NR = TM + TN, G = il_group(SV), RPS = il_reads_per_step(SV), NSTEP = (NM + G - 1) / G;
  static_assert(NR <= NSTEP * RPS && AP < NSTEP && WP < NSTEP, "reads and DMA pieces fit the steps");
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
      if constexpr (RD) static_for<0, RPS>([&](auto rc_) {
        constexpr int r = k * RPS + decltype(rc_)::value;
        if constexpr (r < NR) read_one(ic_t<SET ^ 1>{}, ic_t<r>{}, rstage);
      });
      if constexpr (DK == 1 || DK == 3) static_for<0, WP>([&](auto ic) {
        if constexpr (il_piece_step(SV, decltype(ic)::value, WP, NSTEP) == k) {
          if (DK == 1 || dma_on) issue_w(ic, dstage);
        }
      });
      if constexpr (DK == 2 || DK == 4) static_for<0, AP>([&](auto ic) {
        if constexpr (il_piece_step(SV, decltype(ic)::value, AP, NSTEP) == k) {
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
  } else if constexpr (!STREAM) {
  // ---- fragments: two register sets (k-step 0 / 1 of a K-tile) ----
  bf16x8 fa[2][TM], fw[2][TN];
  auto read_frag = [&](const int set, const int stage) {
    const unsigned char* a = As + stage * STAGE_A + a_row + (set ? c1 : c0);
    const unsigned char* w = Ws + stage * STAGE_W + w_row + (set ? c1 : c0);
#pragma unroll
    for (int i = 0; i < TN; ++i) fw[set][i] = *reinterpret_cast<const bf16x8*>(w + i * 16 * 128);
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[set][i] = *reinterpret_cast<const bf16x8*>(a + i * 16 * 128);
  };
  auto mma = [&](const int set) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
        acc[tn][tm] = mfma_16x16x32(fw[set][tn], fa[set][tm], acc[tn][tm]);
  };

  // ---- K loop: AHEAD = ST - 1 tiles staged beyond the one being multiplied ----
  // Explicit waits use the s_waitcnt BUILTIN (0xC07F = lgkmcnt(0) only), not inline asm: the compiler's own waitcnt
  // insertion sees them and therefore emits no conservative lgkmcnt(0) in front of the MFMA batches (checked in the ISA:
  // barrier, 9 ds_read, 20 MFMA, lgkmcnt(0), 7 DMA, 9 ds_read, 20 MFMA, waits, barrier).
  constexpr int AHEAD = ST - 1;
  if constexpr (LW == 0) {
    issue_tile(0);
    if (AHEAD == 2 && t0 + 1 < t1) {
      issue_tile(1);
      wait_all_but_newest();
    } else {
      wait_vmcnt_imm<0>();
    }
  }
  SD_PIPE_BARRIER();
  stamp(1);
  init_acc();
  read_frag(0, 0);
  int stage = 0;
  if constexpr (LW > 0 && SG) {
    // compute waves with loader waves beside them: no VMEM, no conditionals inside a half-iteration -> reads interleaved
    for (int t = t0; t < t1 - 1; ++t) {
      const int s1 = stage == ST - 1 ? 0 : stage + 1;
      __builtin_amdgcn_s_waitcnt(0xC07F);      // k-step-0 fragments of tile t
      read_frag(1, stage);
      mma(0);
      sgb_reads_under_mfmas<TM + TN, TM * TN>();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xC07F);      // k-step-1 fragments: every read of tile t retired
      SD_PIPE_BARRIER();                       // the loaders published tile t+1
      read_frag(0, s1);
      mma(1);
      sgb_reads_under_mfmas<TM + TN, TM * TN>();
      __builtin_amdgcn_sched_barrier(0);
      stage = s1;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    read_frag(1, stage);
    mma(0);
    sgb_reads_under_mfmas<TM + TN, TM * TN>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    mma(1);
  } else
  for (int t = t0; t < t1; ++t) {
    const int s1 = stage == ST - 1 ? 0 : stage + 1;          // stage of tile t+1
    const int s_new = ST == 3 ? (stage == 0 ? 2 : stage - 1) : s1;   // stage the tile t+AHEAD goes to
    __builtin_amdgcn_s_waitcnt(0xC07F);        // k-step-0 fragments of tile t (issued under the previous MFMA batch)
    // re-staged stage: last read as tile t-1 (ST = 3) / t-1 (ST = 2), retired before the mid barrier of iteration t-1
    if constexpr (LW == 0) {
      if (t + AHEAD < t1) issue_tile(s_new);
    }
    [[maybe_unused]] int rcur = 0;
    if constexpr (PRE) {
      __builtin_amdgcn_sched_barrier(0);       // the counted waits below rely on: DMA of this iteration, THEN these loads
      rcur = prefetch_rows(t);
      __builtin_amdgcn_sched_barrier(0);
    }
    read_frag(1, stage);
    __builtin_amdgcn_sched_barrier(0);
    mma(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);        // k-step-1 fragments (issued TM*TN MFMAs ago): every read of tile t retired
    if (t + 1 < t1) {
      if constexpr (LW == 0 && PRE) {
        // tile t+1 must have landed. Issued after it: (three stages) the residual loads of iteration t-1, the DMA of tile t+2
        // and this iteration's residual loads; (two stages) only this iteration's residual loads
        if (AHEAD == 2) wait_vmcnt_dyn((t + 2 < t1 ? mine : 0) + rprev + rcur);
        else wait_vmcnt_dyn(rcur);
        rprev = rcur;
      } else if constexpr (LW == 0) {
        if (AHEAD == 2 && t + 2 < t1) wait_all_but_newest();   // own pieces of tile t+1 landed, t+2 may stay in flight
        else wait_vmcnt_imm<0>();
      }
      SD_PIPE_BARRIER();                       // publishes tile t+1; every wave is done reading tile t
      read_frag(0, s1);
    }
    __builtin_amdgcn_sched_barrier(0);
    mma(1);
    stage = s1;
  }
  } else if constexpr (IL) {
  // ---- interleaved streaming loop (round 4): the streaming variant below with the LDS-DMA pieces of a tile issued ONE PER STEP ----
  // instead of as a burst of 8-9 at the top of the iteration. Two LDS stages: the stage of tile t is free once every wave has
  // passed the roll-over barrier of iteration t (step STEPS - Q), so the first Q pieces of tile t+2 go out in the Q steps after
  // that barrier and its remaining pieces in the first steps of iteration t+1; the roll-over of iteration t+1 waits for all of
  // them (vmcnt(0): nothing newer is in flight), at least STEPS - Q - (pieces - Q) steps (> 700 cycles) after the last one was
  // issued. W pieces first: a conv's gather state advances once per tile, with the A pieces, which then all sit in one iteration.
  static_assert(ST == 2, "streaming loop is written for two LDS stages");
  constexpr bool HOLD_A = TM <= TN;
  constexpr int HN = HOLD_A ? TM : TN, SN = HOLD_A ? TN : TM, Q = 3, QN = Q + 1, STEPS = 2 * SN, NP = AP + WP;
  static_assert(WP >= Q && NP - Q <= STEPS - Q - 4, "piece schedule of the interleaved streaming loop");
  const int h_row = HOLD_A ? a_row : w_row, s_row = HOLD_A ? w_row : a_row;
  bf16x8 hold[2][HN], qf[QN];
  auto read_hold = [&](auto setc, const int stage) {
    constexpr int set = decltype(setc)::value;
    const unsigned char* b = (HOLD_A ? As + stage * STAGE_A : Ws + stage * STAGE_W) + h_row + (set ? c1 : c0);
    static_for<0, HN>([&](auto ic) { hold[set][decltype(ic)::value] = *reinterpret_cast<const bf16x8*>(b + decltype(ic)::value * 16 * 128); });
  };
  auto read_stream = [&](auto slotc, const int stage, auto jc) {   // fragment of step j (0 .. STEPS-1) of the tile in `stage`
    constexpr int j = decltype(jc)::value, ks = j / SN, sidx = j % SN;
    const unsigned char* b = (HOLD_A ? Ws + stage * STAGE_W : As + stage * STAGE_A) + s_row + (ks ? c1 : c0);
    qf[decltype(slotc)::value] = *reinterpret_cast<const bf16x8*>(b + sidx * 16 * 128);
  };
  auto issue_piece = [&](auto ic, const int stage) {   // W pieces 0 .. WP-1, then A pieces
    constexpr int i = decltype(ic)::value;
    if constexpr (i < WP) {
      issue_w(ic, stage);
      if constexpr (i == WP - 1) next_w();
    } else {
      issue_a(ic_t<i - WP>{}, stage);
      if constexpr (i == NP - 1) next_a();
    }
  };
  // ONE copy of the body; "is there a tile t+1 / t+2" are run-time flags (a dozen never-taken wave-uniform branches per K-tile).
  // Separate copies for the last iterations (as the register-pipelined loop has them) made the register allocator rename the
  // 160 accumulators between the copies: 110 spilled registers in the conv instantiation, 27-70 in the others.
  auto iter = [&](auto rc, const int cur, const bool more2, const bool more) {
    const int nxt = cur ^ 1;
    static_for<0, STEPS>([&](auto jc) {
      constexpr int j = decltype(jc)::value, ks = j / SN, sidx = j % SN;
      // (the k-step-1 held set is read two steps before its first use, not at step 0: its registers are free while the pieces
      // of the first steps -- a conv's gather arithmetic -- are issued; at step 0 the same code spilled 110 registers)
      if constexpr (j == SN - 2) read_hold(ic_t<1>{}, cur);
      if (j == STEPS - Q && more) {
        wait_vmcnt_imm<0>();                      // own pieces of tile t+1 (the last one issued STEPS - NP steps ago)
        __builtin_amdgcn_s_waitcnt(0xC07F);       // every read of tile t retired
        SD_PIPE_BARRIER();
        read_hold(ic_t<0>{}, nxt);
      }
      constexpr int jr = j + Q;
      if constexpr (jr < STEPS) read_stream(ic_t<jr % QN>{}, cur, ic_t<jr>{});
      else if (more) read_stream(ic_t<jr % QN>{}, nxt, ic_t<(jr >= STEPS ? jr - STEPS : 0)>{});
      if constexpr (j < NP - Q) { if (more) issue_piece(ic_t<(j < NP - Q ? Q + j : 0)>{}, nxt); }
      if constexpr (j >= STEPS - Q) {
        if (more2) issue_piece(ic_t<(j >= STEPS - Q ? j - (STEPS - Q) : 0)>{}, cur);
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the read Q steps ahead of its use (the scheduler would sink it)
      static_for<0, HN>([&](auto hc) {
        constexpr int h = decltype(hc)::value;
        if constexpr (HOLD_A) acc[sidx][h] = mfma_16x16x32(qf[j % QN], hold[ks][h], acc[sidx][h]);
        else acc[h][sidx] = mfma_16x16x32(hold[ks][h], qf[j % QN], acc[h][sidx]);
      });
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  const int nt = t1 - t0;
  static_for<0, NP>([&](auto ic) { issue_piece(ic, 0); });
  wait_vmcnt_imm<0>();
  SD_PIPE_BARRIER();
  if (nt > 1) static_for<0, Q>([&](auto ic) { issue_piece(ic, 1); });
  stamp(1);
  init_acc();
  read_hold(ic_t<0>{}, 0);
  static_for<0, Q>([&](auto jc) { read_stream(ic_t<decltype(jc)::value % QN>{}, 0, jc); });
  int cur = 0, nrem = nt;
  for (; nrem > 0; --nrem) {
    iter(ic_t<1>{}, cur, nrem > 2, nrem > 1);
    cur ^= 1;
  }
  } else {
  // ---- streaming variant (256x320 tiles: 160 accumulator registers) ----
  // The smaller operand of a k-step is HELD (two sets, k-step 0 / 1), the larger one STREAMS through a (Q+1)-slot
  // register queue: step j of the 2*SN steps of a K-tile multiplies one streamed fragment with the HN held ones
  // (HN MFMAs) and issues the ds_read of the fragment of step j+Q, so an LDS read has Q*HN MFMAs (>= 192 cycles) to land.
  // Two LDS stages: the DMA of tile t+1 is issued at the top of iteration t and awaited Q steps before its end (85 % of
  // an iteration of lead); there one barrier publishes tile t+1 and proves tile t fully read, after which the queue and
  // the held set roll over into tile t+1 without a bubble.
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

  if constexpr (LW == 0) {
    issue_tile(0);
    wait_vmcnt_imm<0>();
  }
  SD_PIPE_BARRIER();
  stamp(1);
  init_acc();
  read_hold(0, 0);
#pragma unroll
  for (int j = 0; j < Q; ++j) read_stream(j % QN, 0, j / SN, j % SN);
  for (int t = t0; t < t1; ++t) {
    const int cur = (t - t0) & 1, nxt = cur ^ 1;
    const bool more = t + 1 < t1;
    if constexpr (LW == 0) {
      if (more) issue_tile(nxt);   // stage of tile t-1: every wave passed the roll-over barrier of iteration t-1
    }
#pragma unroll
    for (int j = 0; j < STEPS; ++j) {
      const int ks = j / SN, s = j % SN;
      if (j == 0) read_hold(1, cur);
      if (j == STEPS - Q && more) {
        if constexpr (LW == 0) wait_vmcnt_imm<0>();   // own pieces of tile t+1 (issued STEPS - Q steps ago)
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
  if constexpr (LW == 0) {
    if (p.ts) {   // (the stamp must not be taken before the accumulators are final: touch one)
      asm volatile("" ::"v"(acc[TN - 1][TM - 1][0]));
      stamp(2);
    }
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
    gemm_epilogue<TM, TN>(p, acc, m_w, n_w, lane);
  }
  if constexpr (LW == 0) if (p.ts) {
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

template <bool CONV, class CFG, bool LN, int LW = 0, int SG = 0, int IL = 0>
__global__ __launch_bounds__(CFG::THREADS + LW * 64, (LW ? 3 : CFG::MIN_WAVES)) void gemm_pipe_kernel(const GemmArgs p) {
  gemm_pipe_body<CONV, CFG, LN, LW, SG, false, IL>(p);
}
// the early-residual form: the compiler allocates v0 .. v215 only, v216 .. v255 are the landing zone of the asm loads
// (gemm_epilogue.h EpiPre; the attribute takes no template-dependent argument, hence the second entry point)
template <bool CONV, class CFG, int IL = 0>
__global__ __launch_bounds__(CFG::THREADS, CFG::MIN_WAVES) __attribute__((amdgpu_num_vgpr(216))) void gemm_pipe_pre_kernel(const GemmArgs p) {
  gemm_pipe_body<CONV, CFG, false, 0, 0, true, IL>(p);
}

// Grid of a launch: one block per tile, capped at what the chip holds at once (persistent blocks, see the kernel) when there are
// more tiles than that. MI355X_SD_GEMM_PERSIST=0: one block per tile always (A/B switch).
template <class CFG, int LW>
static int pipe_grid_x(int tiles, int ny) {
  static const bool off = [] {
    const char* e = getenv("MI355X_SD_GEMM_PERSIST");
    return e && atoi(e) == 0;
  }();
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  // blocks per CU by LDS and registers: the eight-wave tiles (and anything with loader waves) own a CU, the four-wave ones share it
  constexpr int bpc = (CFG::NW == 8 || LW > 0) ? 1 : 2;
  const int cap = cus * bpc;
  return (off || ny > 1 || tiles <= cap) ? tiles : cap;
}

// loader waves (template LW): MI355X_SD_GEMM_LOADERS=0 | 4 | 5 (5 = 4 loaders + interleaved fragment reads, template SG) for every
// 256x160 launch; -1 = by shape: 4 loaders + interleaved reads where they measured a gain in isolation (K >= 4096: FF2 -10 %, the
// 11520-deep convs -16 %, profiles/r02_gemm_loaders.txt), none for the K = 1280 launches (no gain there)
static int gemm_loaders() {
  static const int v = [] {
    const char* e = getenv("MI355X_SD_GEMM_LOADERS");
    return e ? atoi(e) : GEMM_LOADERS_DEFAULT;
  }();
  return v;
}

template <bool CONV, class CFG, bool LN, int LW, int SG = 0>
static int launch_pipe_lw(const GemmArgs& a, hipStream_t stream) {
  static const bool attr_ok = [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pipe_kernel<CONV, CFG, LN, LW, SG>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES) == hipSuccess;
  }();
  if (!attr_ok) return SD_ERR_HIP;
  const int ntm = (a.M + CFG::BM - 1) / CFG::BM, ntn = (a.N + CFG::BN - 1) / CFG::BN;
  const int ny = a.splitk > 1 ? a.splitk : 1;
  hipLaunchKernelGGL((gemm_pipe_kernel<CONV, CFG, LN, LW, SG>), dim3(pipe_grid_x<CFG, LW>(ntm * ntn, ny), ny), dim3(CFG::THREADS + LW * 64), CFG::LDS_BYTES, stream, a);
  if (a.splitk > 1) launch_splitk_reduce(a, stream);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

// early residual fetch (gemm_pipe_pre_kernel) applies: plain bf16 residual launches of the register-pipelined tiles
template <bool CONV, class CFG, bool LN>
static bool pre_applies(const GemmArgs& a) {
  if constexpr (!CONV && !LN && (CFG::TM + CFG::TN) <= 10 && CFG::TM <= 4) {
    static const bool pre_off = getenv("MI355X_SD_GEMM_NO_PRE") != nullptr;   // A/B switch
    return !pre_off && a.R && !a.r_f32 && !a.geglu && !a.gate && !a.wscale && a.splitk <= 1 && (!a.bias || a.bias_acc) &&
           a.K / BK >= CFG::TM + 2 && !(a.N & 7) && ((size_t)(a.M - 1) * a.ldr + a.N) * 2 < 0xFFFF0000ull;
  } else {
    return false;
  }
}

// interleaved K loop (template IL): MI355X_SD_GEMM_IL=0 selects the round-3 burst loop (A/B switch; same results bit for bit)
static int gemm_il() {
  static const int v = [] {
    const char* e = getenv("MI355X_SD_GEMM_IL");
    return e ? atoi(e) : 1;
  }();
  return v;
}

template <bool CONV, class CFG, bool LN, int IL>
static int launch_pipe_il(const GemmArgs& a, hipStream_t stream) {
  const int ntm = (a.M + CFG::BM - 1) / CFG::BM, ntn = (a.N + CFG::BN - 1) / CFG::BN;
  const int ny = a.splitk > 1 ? a.splitk : 1;
  if constexpr (!CONV && !LN && (CFG::TM + CFG::TN) <= 10 && CFG::TM <= 4) {
    // residual launches (to_out, proj_out, FF2): the residual is fetched during the last TM + 1 K iterations. bf16 residual rows
    // addressed with 32-bit offsets, N % 8 == 0 (a 16-byte pair load never straddles the row's end), bias in the accumulators, no
    // GEGLU / gate / fp8 scale / split-K (those epilogues live elsewhere); not the implicit-GEMM convs (their gather state leaves no
    // room for the 40 landing registers)
    if (pre_applies<CONV, CFG, LN>(a)) {
      static const bool pre_ok = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pipe_pre_kernel<CONV, CFG, IL>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES) == hipSuccess;
      }();
      if (!pre_ok) return SD_ERR_HIP;
      hipLaunchKernelGGL((gemm_pipe_pre_kernel<CONV, CFG, IL>), dim3(pipe_grid_x<CFG, 0>(ntm * ntn, ny), ny), dim3(CFG::THREADS), CFG::LDS_BYTES, stream, a);
      return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
    }
  }
  static const bool attr_ok = [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pipe_kernel<CONV, CFG, LN, 0, 0, IL>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES) == hipSuccess;
  }();
  if (!attr_ok) return SD_ERR_HIP;
  hipLaunchKernelGGL((gemm_pipe_kernel<CONV, CFG, LN, 0, 0, IL>), dim3(pipe_grid_x<CFG, 0>(ntm * ntn, ny), ny), dim3(CFG::THREADS), CFG::LDS_BYTES, stream, a);
  if (a.splitk > 1) launch_splitk_reduce(a, stream);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

template <bool CONV, class CFG, bool LN>
static int launch_pipe(const GemmArgs& a, hipStream_t stream) {
  // only the 256x160 three-stage tile leaves room for 12 waves per CU (166 VGPRs <= the 168 of three waves per SIMD); the larger
  // register tiles (196-246 VGPRs) would spill into their K loops. By shape (-1): the long-K launches that cannot take the early
  // residual fetch -- inside the step the K = 5120 FF2 runs better on that than on loader waves (GEMM class 39.2 vs 40.2 ms,
  // profiles/r03_s5_step_ab.txt), the 11520-deep convs the other way round (conv class 10.4 vs 10.7 ms)
  if constexpr (CFG::NW == 8 && !LN && CFG::BN == 160 && CFG::STAGES == 3) {
    if (gemm_loaders() == 4) return launch_pipe_lw<CONV, CFG, LN, 4>(a, stream);
    if (gemm_loaders() == 5 || (gemm_loaders() == -1 && a.K >= 4096 && a.splitk <= 1 && !pre_applies<CONV, CFG, LN>(a)))
      return launch_pipe_lw<CONV, CFG, LN, 4, 1>(a, stream);   // + interleaved fragment reads
  }
  if constexpr (!CONV && !LN && CFG::NW == 8 && CFG::BN == 160 && CFG::STAGES == 3) {   // schedule experiments: this tile only
    if (gemm_il() == 2) return launch_pipe_il<CONV, CFG, LN, 2>(a, stream);
    if (gemm_il() == 3) return launch_pipe_il<CONV, CFG, LN, 3>(a, stream);
    if (gemm_il() == 4) return launch_pipe_il<CONV, CFG, LN, 4>(a, stream);
  }
  return gemm_il() ? launch_pipe_il<CONV, CFG, LN, 1>(a, stream) : launch_pipe_il<CONV, CFG, LN, 0>(a, stream);
}

// tile: 128 | 160 (pick_tile ids). Returns SD_ERR_UNSUPPORTED when the fast path does not apply (caller falls back).
int launch_gemm_pipe(const GemmArgs& a_in, int tile, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  GemmArgs a = a_in;
  static const bool bias_acc_off = getenv("MI355X_SD_GEMM_NO_BIAS_ACC") != nullptr;   // A/B switch
  a.bias_acc = (a.bias && !a.wscale && a.splitk <= 1 && !a.rowstat && !bias_acc_off) ? 1 : 0;
  static const bool off = getenv("MI355X_SD_NO_PIPE") != nullptr;
  static const bool off128 = getenv("MI355X_SD_NO_PIPE128") != nullptr;   // A/B switch for the 128x128 variant
  if (tile == 128 && off128) return SD_ERR_UNSUPPORTED;
  static const bool on256 = getenv("MI355X_SD_PIPE256") != nullptr;   // experiment: pipelined 256x256 instead of the phased kernel
  if (off || a.wscale || (a.K & 63) || (tile == 160 && a.geglu)) return SD_ERR_UNSUPPORTED;
  static const bool off320 = getenv("MI355X_SD_NO_PIPE320") != nullptr;   // A/B switch for the streaming 256x320 variants
  if (tile == 320 && (off320 || (a.conv && a.geglu))) return SD_ERR_UNSUPPORTED;
  // 256x256: the phased kernel (gemm256.hip) stays the default where it can run (id 257); id 256 is what pick_tile
  // returns when it cannot (A row remap of the MMDiT output projections) and takes the pipelined loop here
  if (tile != 128 && tile != 160 && tile != 320 && tile != 256 && tile != 129 && !(on256 && tile == 257)) return SD_ERR_UNSUPPORTED;
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
  if (tile == 256 || tile == 257) {
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
