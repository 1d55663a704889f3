// Epilogue store-pattern probe: a grid of 256x256 output tiles (8 waves, 128x64 per wave, like gemm256_kernel) writes an
// M x N bf16 matrix with three lane->address mappings; no loads, no math. Measures what the GEMM epilogue's write
// pattern costs by itself.  hipcc --offload-arch=gfx950 -O3 store_probe.hip -o store_probe && ./store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void store_kernel(unsigned short* C, int M, int N, int ldc) {
  const int tiles_n = N / 256;
  const int m0 = (blockIdx.x / tiles_n) * 256, n0 = (blockIdx.x % tiles_n) * 256;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m_wave = m0 + (wave >> 2) * 128, n_wave = n0 + (wave & 3) * 64;
  const int lr = lane & 15, lq = lane >> 4;
  unsigned v = threadIdx.x * 0x10001u;
  if (MODE == 0) {   // today's mapping: lane -> row lr, 4 consecutive columns per 16-wide n sub-tile (8-byte stores)
#pragma unroll
    for (int tm = 0; tm < 8; ++tm)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        u32x2 pk = {v + tm, v + tn};
        *reinterpret_cast<u32x2*>(C + (size_t)(m_wave + tm * 16 + lr) * ldc + n_wave + tn * 16 + lq * 4) = pk;
      }
  } else if (MODE == 1) {   // permuted n: lane owns 8 consecutive columns per half (16-byte stores, 64 B per row per instruction)
#pragma unroll
    for (int tm = 0; tm < 8; ++tm)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        u32x4 pk = {v + tm, v + h, v, v + 1};
        *reinterpret_cast<u32x4*>(C + (size_t)(m_wave + tm * 16 + lr) * ldc + n_wave + h * 32 + lq * 8) = pk;
      }
  } else if (MODE == 2) {   // row-contiguous: 8 lanes x 16 B = one 128-byte row segment, 8 rows per instruction
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      u32x4 pk = {v + i, v, v, v + 1};
      *reinterpret_cast<u32x4*>(C + (size_t)(m_wave + i * 8 + (lane >> 3)) * ldc + n_wave + (lane & 7) * 8) = pk;
    }
  } else {   // block-wide row-contiguous: a 512-byte row of the tile per 32 lanes (what an LDS-staged epilogue could do)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      u32x4 pk = {v + i, v, v, v + 1};
      const int r = i * 16 + (threadIdx.x >> 5);
      *reinterpret_cast<u32x4*>(C + (size_t)(m0 + r) * ldc + n0 + (threadIdx.x & 31) * 8) = pk;
    }
  }
}

template <int MODE>
static void run(unsigned short* C, int M, int N) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int grid = (M / 256) * (N / 256);
  float best = 1e9f;
  for (int it = 0; it < 6; ++it) {
    hipEventRecord(a);
    store_kernel<MODE><<<grid, 512>>>(C, M, N, N);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (it && ms < best) best = ms;
  }
  printf("M=%d N=%d mode %d: %.1f us  %.2f TB/s\n", M, N, MODE, best * 1e3f, 2.0 * M * N / (best * 1e-3) / 1e12);
}

int main() {
  unsigned short* C;
  hipMalloc(&C, (size_t)32768 * 10240 * 2);
  const int shapes[][2] = {{8192, 10240}, {8192, 3840}, {8192, 1280}, {32768, 10240}};
  for (auto& s : shapes) {
    run<0>(C, s[0], s[1]); run<1>(C, s[0], s[1]); run<2>(C, s[0], s[1]); run<3>(C, s[0], s[1]);
  }
  return 0;
}
