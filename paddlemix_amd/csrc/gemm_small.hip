// 64 x 64 x 64 GEMM tile with a deep LDS ring for the small launches of a batch-1 step (SD-1.5 at 512^2: 64 .. 4096 rows against
// N, K = 320 .. 1280) -- problems of ~1 GFLOP whose time is a chain of latencies, not work. Round 6.
//
// Why. These launches took split-K slices of 128 x 128 tiles plus a reduce kernel: 13.5 + 7 us per call on the step's three most
// frequent shapes (256 x 1280 x 1280, 1024 x 640 x 640, 4096 x 320 x 320: 75 of its 146 linear launches), where the vendor library
// answers with ONE kernel of 64 x 64 (or smaller) tiles in 7.4 - 9.9 us (profiles/r06_s11_blas_kernels_sd15.txt, kernel times of a
// rocprofv3 trace; through Python both sides are host-bound at 15 - 18 us per call). What such a tile needs is not MFMA scheduling
// -- a block multiplies for ~0.4 us in all -- but a short serial chain: K-tiles in flight instead of K-tiles in sequence.
//   * four waves, 32 x 32 outputs each (2 x 2 MFMA tiles), BK = 64; LDS image / DMA pieces / swizzle / "swapped" issue / epilogue as
//     in gemm_pipe.hip (one 1-KiB piece = 8 rows x 128 B; a 64-row operand tile = 8 pieces = 2 per wave);
//   * a ring of NS = 6 stages of 16 KiB: the prologue puts up to five K-tiles in flight, iteration t waits for its own pieces of
//     tile t only (counted vmcnt: the newer tiles stay in flight), ONE barrier, then re-issues the stage tile t-1 just left
//     (every wave retired its reads of it before the barrier) and multiplies tile t;
//   * no split-K, no reduce kernel, no fp32 slabs; 96 KiB of LDS: one block per CU, which is what these grids (20 .. 320 blocks) are.
#include <type_traits>

#include "common.h"
#include "gemm_cfg.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace sd {

namespace small {
__device__ __forceinline__ void dma(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds, unsigned voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)lds, 16, voff, soff, 0, 0);
}
}  // namespace small

template <bool LN>
__global__ __launch_bounds__(256, 1) void gemm_small_kernel(const GemmArgs p) {
  constexpr int BM = 64, BN = 64, TM = 2, TN = 2, NS = 6, PPT = 4;   // PPT: pieces per K-tile and wave (2 of A, 2 of W)
  constexpr int STAGE = (BM + BN) * BK * 2;                             // 16 KiB: [A 64 rows][W 64 rows] x 128 B
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  int tile_m, tile_n;
  tile_coords(xcd_remap(blockIdx.x, ntm * ntn), ntm, ntn, p.gm, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nt = p.K / BK;   // K % 64 == 0 (launcher)

  const int sub = lane >> 3, cg = (lane & 7) ^ sub;
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.A), 0, 0xFFFFFFE0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.W), 0, (unsigned)((size_t)p.N * p.K * 2), 0x00020000);
  unsigned a_off[2], w_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + (wave + i * 4) * 8 + sub;
    const size_t arow = p.a_rpb ? (size_t)(m / p.a_rpb) * p.a_bstride + (size_t)(m % p.a_rpb) * p.lda : (size_t)m * p.lda;
    a_off[i] = m < p.M ? (unsigned)((arow + cg * 8) * 2) : OOB;
    const int n = n0 + w_row_of_lds_row<TN>((wave + i * 4) * 8 + sub, p.geglu);
    w_off[i] = n < p.N ? (unsigned)(((size_t)n * p.K + cg * 8) * 2) : OOB;
  }
  auto issue_tile = [&](const int t) {
    unsigned char* st = smem + (t % NS) * STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) small::dma(a_rsrc, st + (wave + i * 4) * 1024, a_off[i], t * BK * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i) small::dma(w_rsrc, st + BM * 128 + (wave + i * 4) * 1024, w_off[i], t * BK * 2);
  };

  // the ring first, then the bias the accumulators start at (the compiler waits for an ordinary load with vmcnt(0): behind the DMA
  // that costs the landing of the whole prologue once, in front of it a full load latency before the first piece is even requested)
  const int npre = min(nt, NS - 1);
  for (int t = 0; t < npre; ++t) issue_tile(t);
  f32x4 acc[TN][TM], bias4[TN];
  const bool bias_acc = !LN && p.bias_acc;
#pragma unroll
  for (int i = 0; i < TN; ++i)
    bias4[i] = bias_acc ? *reinterpret_cast<const f32x4*>(p.bias + min(n0 + wn * 32 + acc_col<TN>(i, lane >> 4, p.geglu), p.N - 4))
                        : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = bias4[i];

  const int frow = lane & 15, fkc = lane >> 4, rsw = frow & 7;
  const int a_row = (wm * 32 + frow) * 128, w_row = BM * 128 + (wn * 32 + frow) * 128;
  const int c0 = ((0 * 4 + fkc) ^ rsw) << 4, c1 = ((1 * 4 + fkc) ^ rsw) << 4;
  for (int t = 0; t < nt; ++t) {
    // own pieces of tile t landed: the tiles issued after it (at most NS - 2 of them while the ring is full) may stay in flight.
    // s_waitcnt takes an immediate: one instruction per possible count, selected by a wave-uniform branch.
    const int newer = min(nt - 1 - t, NS - 2);
    switch (newer) {
      case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PPT) : "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PPT) : "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPT) : "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * PPT) : "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the fragment reads of tile t-1: its stage is re-issued below)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (t + NS - 1 < nt) issue_tile(t + NS - 1);   // into the stage of tile t-1
    const unsigned char* st = smem + (t % NS) * STAGE;
    bf16x8 fa[2][TM], fw[2][TN];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < TN; ++i) fw[ks][i] = *reinterpret_cast<const bf16x8*>(st + w_row + (ks ? c1 : c0) + i * 16 * 128);
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[ks][i] = *reinterpret_cast<const bf16x8*>(st + a_row + (ks ? c1 : c0) + i * 16 * 128);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = mfma_16x16x32(fw[ks][i], fa[ks][j], acc[i][j]);
  }
  const int m_w = m0 + wm * 32, n_w = n0 + wn * 32;
  if constexpr (LN) gemm_epilogue_ln<TM, TN>(p, acc, m_w, n_w, lane);
  else gemm_epilogue<TM, TN>(p, acc, m_w, n_w, lane);
}

// what the small tile takes: 16-bit operands of a linear launch, K % 64 == 0, few enough K-tiles that one block walks them all
// (long-K weight streaming -- FF2 at K = 5120 -- needs the bandwidth of more blocks: split-K keeps those). NOT the small 3x3 convs:
// built and measured (profiles/r06_s13_sd15_small_tile_conv_ab.txt), 45 - 90 K-tiles per block lose to 9 - 18 per slice.
bool gemm_small_applies(const GemmArgs& a) {
  if (a.conv || a.wscale || a.splitk > 1 || (a.K & 63)) return false;
  if (a.M > 4096 || (long)((a.M + 127) / 128) * ((a.N + 127) / 128) > 128) return false;   // (launches that would take split-K slices)
  // one block walks all K-tiles at ~0.2 us each: past ~48 of them per block the chain is longer than the sliced form's, unless there
  // are enough blocks that the slices' partial sums would cost more than the chain (rows x columns >= 160 tiles of 64 x 64)
  const long tiles = (long)((a.M + 63) / 64) * ((a.N + 63) / 64);
  const int nt = a.K / BK;
  // (the three constants scanned inside the batch-1 SD-1.5 step, 24 .. 96 / 64 .. 320 / 40 .. 128: 5.22 .. 5.27 ms, flat --
  // profiles/r06_s35_small_policy.txt)
  if (nt > 48 && !(tiles >= 160 && nt <= 96)) return false;
  const size_t lim = 0xFFFF0000ull;
  size_t a_ext;
  if (a.a_rpb) a_ext = ((size_t)((a.M - 1) / a.a_rpb) * a.a_bstride + (size_t)(a.a_rpb - 1) * a.lda + a.K) * 2;
  else a_ext = ((size_t)(a.M - 1) * a.lda + a.K) * 2;
  return a_ext < lim && (size_t)a.N * a.K * 2 < lim;
}

int launch_gemm_small(const GemmArgs& a_in, hipStream_t stream) {
  if (!gemm_small_applies(a_in)) return SD_ERR_UNSUPPORTED;
  GemmArgs a = a_in;
  const bool ln = a.rowstat != nullptr;
  static const bool bias_acc_off = sd_switch("MI355X_SD_GEMM_NO_BIAS_ACC") != nullptr;   // A/B switch (as launch_gemm_pipe)
  a.bias_acc = (a.bias && !ln && !bias_acc_off) ? 1 : 0;
  constexpr int LDS_BYTES = 6 * (64 + 64) * BK * 2;
  static const bool attr_ok = [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_small_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_small_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) == hipSuccess;
  }();
  if (!attr_ok) return SD_ERR_HIP;
  const int tiles = ((a.M + 63) / 64) * ((a.N + 63) / 64);
  if (ln) hipLaunchKernelGGL(gemm_small_kernel<true>, dim3(tiles), dim3(256), LDS_BYTES, stream, a);
  else hipLaunchKernelGGL(gemm_small_kernel<false>, dim3(tiles), dim3(256), LDS_BYTES, stream, a);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

}  // namespace sd
