// Tile configurations shared by the GEMM kernels (gemm.hip: generic loop; gemm_pipe.hip: software-pipelined loops).
#pragma once
#include "common.h"

namespace sd {

constexpr int BK = 64;

template <int WAVES_M_, int WAVES_N_, int TM_, int TN_, int STAGES_ = 2, int MIN_WAVES_ = 2>
struct GemmCfg {
  static constexpr int WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, TM = TM_, TN = TN_, STAGES = STAGES_;
  static constexpr int MIN_WAVES = MIN_WAVES_;   // __launch_bounds__ occupancy hint (waves per SIMD)
  static constexpr int NW = WAVES_M * WAVES_N;
  static constexpr int THREADS = NW * 64;
  static constexpr int BM = WAVES_M * TM * 16;
  static constexpr int BN = WAVES_N * TN * 16;
  static constexpr int A_TOTAL = BM / 8, W_TOTAL = BN / 8;   // 1-KiB (8 rows x 128 B) LDS-DMA pieces per K-tile
  static constexpr int A_PIECES = (A_TOTAL + NW - 1) / NW;    // per wave (the last waves may own one fewer)
  static constexpr int W_PIECES = (W_TOTAL + NW - 1) / NW;
  static constexpr int LDS_BYTES = STAGES * (BM + BN) * BK * 2;
};
using Cfg128 = GemmCfg<2, 2, 4, 4>;       // 128 x 128, 2 blocks / CU
using Cfg256 = GemmCfg<2, 4, 8, 4>;       // 256 x 256
using Cfg256x160 = GemmCfg<4, 2, 4, 5>;   // 256 x 160: N = 1280 -> 8 column tiles (8192 x 1280 = exactly 256 tiles)
using Cfg256x160s3 = GemmCfg<4, 2, 4, 5, 3>;   // the same with three LDS stages (3 x 52 KiB of the 160 KiB), gemm_pipe.hip
using Cfg256x320 = GemmCfg<2, 4, 8, 5>;   // 256 x 320: N = 640 -> 2 column tiles
using Cfg256x320g = GemmCfg<4, 2, 4, 10>; // 256 x 320 with an even tile count per wave (GEGLU value/gate pairs)
// a four-wave tile sized for TWO co-resident blocks per CU (<= 80 KiB of LDS, <= 256 VGPRs): while one block sits in its prologue,
// barrier or epilogue the other one's waves feed the matrix pipe from the same SIMDs. Round 3 measured three of them per shape
// (profiles/r03_s3_gemm_variants_tiles.txt): 160 x 160 (80 x 80 per wave) and 192 x 128 (96 x 64) lose everywhere (FF1 921 vs 1069 TFLOP/s,
// to_out 580 vs 756) and are gone; 128 x 160 ties the eight-wave 256 x 160 tile on the K = 1280 projections and wins a few
// 640-wide launches
using Cfg128x160 = GemmCfg<2, 2, 4, 5>;   // 128 x 160: the 8-wave 256 x 160 tile cut in two, 2 x 36 KiB


template <int N>
__device__ __forceinline__ void wait_vmcnt_imm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Candidate tile ids of pick_tile (gemm.hip)
int launch_gemm_pipe(const struct GemmArgs& a, int tile, void* stream);   // SD_ERR_UNSUPPORTED: caller falls back

}  // namespace sd
