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
template <int BS, int DSP>
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(const GemmArgs p) {
  using namespace w4;
  constexpr int BM = 256, BN = 256, TM = 8, TN = 8, NW = 4, AP = 8, WP = 8;
  constexpr int STAGE_A = BM * BK * 2, STAGE_W = BN * BK * 2;   // 32 KiB each
  constexpr int NSTEP = 32;                                      // steps of two MFMAs per k-step
  static_assert(BS >= 16 && BS + 1 + 15 * DSP < 2 * NSTEP, "the 16 pieces of a K-tile are issued inside the iteration");
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

  for (int vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));   // (per-tile lane constants must not be hoisted out of the tile loop and kept alive across the K loop)
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lid = xcd_remap(vb, nvb);
    int tile_m, tile_n;
    tile_coords(lid, ntm, ntn, p.gm, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- LDS-DMA geometry (gemm_pipe.hip): piece q = wave + 4 i (8 rows x 128 B); lane -> row q*8 + (lane>>3), chunk (lane&7) ^ row ----
    const int sub = lane >> 3;
    const int cg = (lane & 7) ^ sub;
    constexpr unsigned OOB = 0xFFFFFFF0u;
    unsigned a_off[AP], w_off[WP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int m = m0 + (wave + i * NW) * 8 + sub;
      const size_t arow = p.a_rpb ? (size_t)(m / p.a_rpb) * p.a_bstride + (size_t)(m % p.a_rpb) * p.lda : (size_t)m * p.lda;
      a_off[i] = m < p.M ? (unsigned)((arow + cg * 8) * 2) : OOB;
    }
#pragma unroll
    for (int i = 0; i < WP; ++i) {
      const int n = n0 + w_row_of_lds_row<TN>((wave + i * NW) * 8 + sub, p.geglu);
      w_off[i] = (n < p.N) ? (unsigned)(((size_t)n * p.K + cg * 8) * 2) : OOB;
    }
    // piece j of a K-tile: j < 8 an A piece, else a W piece; kb = byte offset of the K-tile
    auto issue_piece = [&](auto jc, const int stage, const int kb) {
      constexpr int j = decltype(jc)::value;
      if constexpr (j < AP) dma(a_rsrc, As + stage * STAGE_A + (wave + j * NW) * 1024, a_off[j], kb);
      else dma(w_rsrc, Ws + stage * STAGE_W + (wave + (j - AP) * NW) * 1024, w_off[j - AP], kb);
    };

    // ---- prologue: tiles 0 and 1 on their way (nothing else: the accumulators need no initialisation, the first k-step's MFMAs
    // take the constant 0 as their C operand; the bias is added in the epilogue from registers, gemm_epilogue.h BR) ----
    static_for<0, AP + WP>([&](auto jc) { issue_piece(jc, 0, 0); });
    static_for<0, AP + WP>([&](auto jc) { issue_piece(jc, 1, min(1, nt - 1) * BK * 2); });
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
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AP + WP) : "memory");   // tile 0 landed (tile 1 may still fly)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, TM + TN>([&](auto kc) { read_one(ic_t<0>{}, kc, 0); });

    // iteration of tile t (stage cur = t & 1): DMA = 1: the pieces of tile t+2 go out behind the barrier; NEXT = 1: there is a tile t+1;
    // FIRST = 1: t = 0, the accumulators' chains start here
    auto iter = [&](auto dmac, auto nextc, auto firstc, const int cur, const int kb2) {
      constexpr int DMA = decltype(dmac)::value, NEXT = decltype(nextc)::value, FIRST = decltype(firstc)::value;
      static_for<0, 2 * NSTEP>([&](auto sc) {
        constexpr int s = decltype(sc)::value, ks = s / NSTEP, st = s % NSTEP;
        if constexpr (ks == 0 && st < TM + TN) read_one(ic_t<1>{}, ic_t<st>{}, cur);                    // set 1 of tile t
        if constexpr (ks == 1 && st < TM + TN && NEXT) read_one(ic_t<0>{}, ic_t<st>{}, cur ^ 1);        // set 0 of tile t+1
        if constexpr (DMA && s > BS && (s - BS - 1) % DSP == 0 && (s - BS - 1) / DSP < AP + WP)
          issue_piece(ic_t<(s - BS - 1) / DSP>{}, cur, kb2);
        if constexpr (s == BS && NEXT) {
          // every read of tile t retired; own pieces of tile t+1 landed (nothing newer in flight) -> the barrier publishes tile t+1
          // and frees the stage of tile t
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

    const int m_w = m0 + wm * (TM * 16), n_w = n0 + wn * (TN * 16);
    gemm_epilogue<TM, TN, false, true>(p, acc, m_w, n_w, lane);
  }
}

// shapes the four-wave tile takes: 16-bit operands, K % 64 == 0, no split-K / LN fold / conv / fp8 scale, and an epilogue without
// per-row operands (residual, gate, row bias: the general epilogue of 8-sub-tile waves fetches those one group at a time)
bool gemm_w4_applies(const GemmArgs& a) {
  if (a.conv || a.rowstat || a.wscale || a.R || a.gate || a.rowbias || a.out_f32 || a.splitk > 1) return false;
  if ((a.K & 63) || a.K < 192 || (a.N & 7)) return false;   // (at least three K-tiles: first / last-but-one / last iterations)
  if (a.geglu && (a.N & 31)) return false;
  const size_t lim = 0xFFFF0000ull;
  const size_t a_ext = a.a_rpb ? ((size_t)((a.M - 1) / a.a_rpb) * a.a_bstride + (size_t)(a.a_rpb - 1) * a.lda + a.K) * 2
                               : ((size_t)(a.M - 1) * a.lda + a.K) * 2;
  return a_ext < lim && (size_t)a.N * a.K * 2 < lim;
}

int launch_gemm_w4(const GemmArgs& a_in, hipStream_t stream) {
  if (!gemm_w4_applies(a_in)) return SD_ERR_UNSUPPORTED;
  GemmArgs a = a_in;
  a.bias_acc = 0;   // (this kernel adds the bias in its epilogue, from registers: the order of the other kernels' generic form)
  constexpr int LDS_BYTES = 2 * (256 + 256) * BK * 2;
  using K = void (*)(const GemmArgs);
  // (BS, DSP) of the shipped schedule; the debug build can pick another instantiation for A/B runs (MI355X_SD_W4_SCHED=0..3)
  static const int sched = [] {
    const char* e = sd_switch("MI355X_SD_W4_SCHED");
    return e ? atoi(e) : 0;
  }();
  K kern = sched == 1 ? (K)gemm_w4_kernel<20, 1> : sched == 2 ? (K)gemm_w4_kernel<16, 2> : sched == 3 ? (K)gemm_w4_kernel<24, 2>
                                                                                                     : (K)gemm_w4_kernel<18, 2>;
  static bool attr_done[4] = {false, false, false, false};
  const int si = (sched >= 1 && sched <= 3) ? sched : 0;
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
  const int tiles = ((a.M + 255) / 256) * ((a.N + 255) / 256);
  hipLaunchKernelGGL(kern, dim3(tiles < cus ? tiles : cus), dim3(256), LDS_BYTES, stream, a);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

}  // namespace sd
