// EXPERIMENT (off by default, MI355X_SD_GEMM_PERSIST=1; written at the end of round 2 without GPU time left -- not yet run on
// hardware): persistent-block form of the streaming 256x320 loop of gemm_pipe.hip.
//
// Why: a K = 640..2880 launch of the 256x320 tiles spends 22-30 % of its time outside the K loop (launch
// ramp, pipeline prologue, epilogue, block turnover), and one 8-wave block owns a CU (147 KiB of LDS), so nothing overlaps it.
// The launches that use these tiles put 2-4 tiles on every CU one after the other (FF1 GEGLU 8192x10240x1280: 1024 tiles; the
// 640-channel level; the 131072-row convs). Here ONE block per CU walks its tiles: the LDS-DMA of the next tile's first K-tile
// is issued at the top of the current tile's last K iteration (into the stage that iteration no longer reads), so it is in flight
// under that iteration's MFMAs and under the epilogue, and the block never leaves the CU (no block turnover, no SRD / geometry
// prologue in front of an idle matrix pipe).
//
// Same tiles, LDS image, fragment streaming (held operand + 4-slot queue), hazards argument and epilogue as the streaming branch
// of gemm_pipe_kernel; the additions are marked "persist:". Work item w = blockIdx.x + i * gridDim.x keeps w % 8 == blockIdx.x % 8,
// so xcd_remap() still hands every XCD one contiguous range of logical tiles, 32 consecutive ones per round.
#include <stdlib.h>

#include "common.h"
#include "gemm_cfg.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace sd {

#define SD_PERSIST_BARRIER()              \
  do {                                    \
    __builtin_amdgcn_sched_barrier(0);    \
    __builtin_amdgcn_s_barrier();         \
    __builtin_amdgcn_sched_barrier(0);    \
  } while (0)

// (a plain function: see the note at dma() in gemm_pipe.hip)
__device__ __forceinline__ void dma_persist(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds, unsigned voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)lds, 16, voff, soff, 0, 0);
}

template <bool CONV, class CFG>
__global__ __launch_bounds__(CFG::THREADS, CFG::MIN_WAVES) void gemm_stream_persist_kernel(const GemmArgs p) {
  constexpr int BM = CFG::BM, BN = CFG::BN, TM = CFG::TM, TN = CFG::TN, NW = CFG::NW;
  static_assert(CFG::STAGES == 2 && (TM + TN) > 10, "the persistent form exists for the two-stage streaming tiles only");
  constexpr int AP = (CFG::A_TOTAL + NW - 1) / NW, WP = (CFG::W_TOTAL + NW - 1) / NW;
  constexpr int STAGE_A = BM * BK * 2, STAGE_W = BN * BK * 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;                 // [2][BM][128 B]
  unsigned char* Ws = smem + 2 * STAGE_A;   // [2][BN][128 B]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / CFG::WAVES_N, wn = wave % CFG::WAVES_N;

  const int ntn = (p.N + BN - 1) / BN;
  const int ntm = (p.M + BM - 1) / BM;
  const int nwork = ntm * ntn;
  const int nt_all = p.K / BK;   // K % 64 == 0 on this path; no split-K here

  const int sub = lane >> 3;
  const int cg = (lane & 7) ^ sub;
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.A), 0, 0xFFFFFFE0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.W), 0, (unsigned)((size_t)p.N * p.K * 2), 0x00020000);

  // ---- per-tile LDS-DMA geometry (persist: recomputed in place when the issue stream moves on to the next tile) ----
  unsigned a_off[AP], w_off[WP];
  int oy[AP], ox[AP];
  bool a_ok[AP];
  int m0 = 0, n0 = 0;
  int gtap = 0, gcch = 0, kiss = 0;
  auto setup_tile = [&](const int work) {
    const int lid = xcd_remap(work, nwork);
    int tile_m, tile_n;
    tile_coords(lid, ntm, ntn, p.gm, tile_m, tile_n);
    m0 = tile_m * BM;
    n0 = tile_n * BN;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int m = m0 + (wave + i * NW) * 8 + sub;
      a_ok[i] = m < p.M;
      const int mm = a_ok[i] ? m : 0;
      if (CONV) {
        const int hw = p.Ho * p.Wo;
        const int b = mm / hw;
        const int rem = mm - b * hw;
        oy[i] = rem / p.Wo;
        ox[i] = rem - oy[i] * p.Wo;
        a_off[i] = (unsigned)((size_t)b * p.Hs * p.Ws * p.lda * 2);
      } else {
        const size_t arow = p.a_rpb ? (size_t)(mm / p.a_rpb) * p.a_bstride + (size_t)(mm % p.a_rpb) * p.lda : (size_t)mm * p.lda;
        a_off[i] = a_ok[i] ? (unsigned)((arow + cg * 8) * 2) : OOB;
        oy[i] = ox[i] = 0;
      }
    }
#pragma unroll
    for (int i = 0; i < WP; ++i) {
      const int n = n0 + w_row_of_lds_row<TN>((wave + i * NW) * 8 + sub, p.geglu);
      w_off[i] = (n < p.N) ? (unsigned)(((size_t)n * p.K + cg * 8) * 2) : OOB;
    }
    gtap = 0;
    gcch = cg * 8;
    if (CONV) conv_k_init(p.kb64, 0, cg * 8, p.Cin, gtap, gcch);
    kiss = 0;
  };

  auto issue_tile = [&](const int stage_) {
    // (persist: stage and K offset are wave-uniform by construction but loop-carried through the tile switch; say so, or the
    // compiler wraps every LDS-DMA in a waterfall loop over its scalar operands)
    const int stage = __builtin_amdgcn_readfirstlane(stage_);
    const int koff = __builtin_amdgcn_readfirstlane(kiss * 2);
    unsigned char* a = As + stage * STAGE_A + wave * 1024;
    unsigned char* w = Ws + stage * STAGE_W + wave * 1024;
    if (CONV) {
      const int ky = gtap / 3, kx = gtap - ky * 3;
      const int Hin = p.Hs << p.up, Win = p.Ws << p.up;
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const int iy = oy[i] * p.stride + ky - p.pad;
        const int ix = ox[i] * p.stride + kx - p.pad;
        const bool ok = a_ok[i] && (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win;
        const unsigned off = a_off[i] + (unsigned)(((iy >> p.up) * p.Ws + (ix >> p.up)) * p.lda + gcch) * 2u;
        if (CFG::A_TOTAL % NW == 0 || wave + i * NW < CFG::A_TOTAL) dma_persist(a_rsrc, a + i * (NW * 1024), ok ? off : OOB, 0);
      }
      conv_k_next(p.kb64, p.Cin, gtap, gcch);
    } else {
#pragma unroll
      for (int i = 0; i < AP; ++i)
        if (CFG::A_TOTAL % NW == 0 || wave + i * NW < CFG::A_TOTAL) dma_persist(a_rsrc, a + i * (NW * 1024), a_off[i], koff);
    }
#pragma unroll
    for (int i = 0; i < WP; ++i)
      if (CFG::W_TOTAL % NW == 0 || wave + i * NW < CFG::W_TOTAL) dma_persist(w_rsrc, w + i * (NW * 1024), w_off[i], koff);
    kiss += BK;
  };

  f32x4 acc[TN][TM];
  const int frow = lane & 15, fkc = lane >> 4, rsw = frow & 7;
  const int a_row = (wm * (TM * 16) + frow) * 128, w_row = (wn * (TN * 16) + frow) * 128;
  const int c0 = ((0 * 4 + fkc) ^ rsw) << 4, c1 = ((1 * 4 + fkc) ^ rsw) << 4;

  // ---- fragment streaming, exactly as in gemm_pipe_kernel's STREAM branch ----
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

  int work = blockIdx.x;   // persist: grid = min(tiles, CUs); every block walks work, work + gridDim.x, ...
  setup_tile(work);
  issue_tile(0);
  wait_vmcnt_imm<0>();
  SD_PERSIST_BARRIER();
  int par = 0;             // persist: LDS stage of the K-tile the multiply stream is at (runs on across output tiles)
  for (;;) {
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the tile being multiplied (setup_tile() for the NEXT tile overwrites m0 / n0 during the last K iteration)
    const int m_w = m0 + wm * (TM * 16), n_w = n0 + wn * (TN * 16);
    const int next_work = work + (int)gridDim.x;
    const bool have_next = next_work < nwork;

    read_hold(0, par);
#pragma unroll
    for (int j = 0; j < Q; ++j) read_stream(j % QN, par, j / SN, j % SN);
    for (int t = 0; t < nt_all; ++t) {
      const int cur = par, nxt = par ^ 1;
      const bool more = t + 1 < nt_all;
      if (more) {
        issue_tile(nxt);   // stage of K-tile t-1: every wave passed the roll-over barrier of iteration t-1
      } else if (have_next) {
        // persist: the issue stream moves on -- first K-tile of the next output tile, into the stage this (last) iteration does
        // not read. WAR: that stage held K-tile t-1, retired before the roll-over barrier of iteration t-1 (for a one-K-tile
        // GEMM: before the barrier that ended the previous output tile).
        setup_tile(next_work);
        issue_tile(nxt);
      }
#pragma unroll
      for (int j = 0; j < STEPS; ++j) {
        const int ks = j / SN, s = j % SN;
        if (j == 0) read_hold(1, cur);
        if (j == STEPS - Q && more) {
          wait_vmcnt_imm<0>();                  // own pieces of K-tile t+1 (issued STEPS - Q steps ago)
          __builtin_amdgcn_s_waitcnt(0xC07F);   // every read of K-tile t retired
          SD_PERSIST_BARRIER();
          read_hold(0, nxt);
        }
        const int jr = j + Q;
        if (jr < STEPS) read_stream(jr % QN, cur, jr / SN, jr % SN);
        else if (more) read_stream(jr % QN, nxt, (jr - STEPS) / SN, (jr - STEPS) % SN);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < HN; ++h) {
          if (HOLD_A) acc[s][h] = mfma_16x16x32(qf[j % QN], hold[ks][h], acc[s][h]);
          else acc[h][s] = mfma_16x16x32(hold[ks][h], qf[j % QN], acc[h][s]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      par = nxt;
    }
    // persist: this wave's pieces of the next tile's first K-tile have had a whole K iteration to land; wait for them BEFORE the
    // epilogue's stores join the same counter (vmcnt counts stores on gfx9: waiting after them would expose the store drain)
    if (have_next) wait_vmcnt_imm<0>();
    gemm_epilogue<TM, TN>(p, acc, m_w, n_w, lane);
    if (!have_next) break;
    work = next_work;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    SD_PERSIST_BARRIER();   // publishes K-tile 0 of the next tile (stage `par`); every wave is done reading this tile's last K-tile
  }
}

static bool persist_on() {
  static const bool v = [] {
    const char* e = getenv("MI355X_SD_GEMM_PERSIST");
    return e && atoi(e) != 0;
  }();
  return v;
}

template <bool CONV, class CFG>
static int launch_persist_cfg(const GemmArgs& a, hipStream_t stream) {
  static const bool attr_ok = [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_stream_persist_kernel<CONV, CFG>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES) == hipSuccess;
  }();
  if (!attr_ok) return SD_ERR_HIP;
  static const int num_cu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return n & ~7;   // a multiple of the XCD count, so that work % 8 == blockIdx.x % 8 holds for every work item
  }();
  const int ntm = (a.M + CFG::BM - 1) / CFG::BM, ntn = (a.N + CFG::BN - 1) / CFG::BN;
  const int tiles = ntm * ntn;
  hipLaunchKernelGGL((gemm_stream_persist_kernel<CONV, CFG>), dim3(tiles < num_cu ? tiles : num_cu), dim3(CFG::THREADS),
                     CFG::LDS_BYTES, stream, a);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

// Called by launch_gemm_pipe for the 256x320 tiles. SD_ERR_UNSUPPORTED: not switched on / not applicable -> the caller continues
// with the one-tile-per-block kernel.
int launch_gemm_persist(const GemmArgs& a, int tile, void* stream_) {
  if (!persist_on() || tile != 320 || a.rowstat || a.splitk > 1 || a.wscale || (a.K & 63)) return SD_ERR_UNSUPPORTED;
  const int tiles = ((a.M + 255) / 256) * ((a.N + 319) / 320);
  if (tiles <= 256) return SD_ERR_UNSUPPORTED;   // one tile per CU: nothing to overlap
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (a.geglu) return a.conv ? SD_ERR_UNSUPPORTED : launch_persist_cfg<false, Cfg256x320g>(a, stream);
  return a.conv ? launch_persist_cfg<true, Cfg256x320>(a, stream) : launch_persist_cfg<false, Cfg256x320>(a, stream);
}

}  // namespace sd
