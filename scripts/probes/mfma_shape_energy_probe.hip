// Which MFMA shape does more work per joule on a power-capped MI355X? Dense MFMA streams on RANDOM bf16 operands (operand toggling is
// most of the matrix pipe's power: MI355X_MICROARCH.md "DVFS give-back"), no memory traffic in the loop, for
//   v_mfma_f32_32x32x16_bf16 (16 MACs per operand element, the attention kernels' shape) and
//   v_mfma_f32_16x16x32_bf16 ( 8 MACs per operand element, the GEMM kernels' shape),
// each wave cycling through four operand pairs and four / eight accumulators. Every kernel stamps s_memtime against the 100-MHz
// wall clock: the board settles each stream at the clock its power allows, so TFLOP/s = pipe utilisation x that clock; the shape
// that sustains more TFLOP/s under the same cap is the cheaper one per flop.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_shape_energy_probe.hip -o /tmp/shp && /tmp/shp
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE>   // 32: 32x32x16, 16: 16x16x32
__global__ __launch_bounds__(256, 2) void mfma_kernel(const bf16x8* ops, float* out, unsigned long long* stamp, int iters) {
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = ops[(i * 2 + 0) * 256 + threadIdx.x];
    b[i] = ops[(i * 2 + 1) * 256 + threadIdx.x];
  }
  float r = 0.f;
  if constexpr (SHAPE == 32) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
      for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int c = 0; c < 16; ++c) acc[c & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(c >> 2) & 3], b[c & 3], acc[c & 3], 0, 0, 0);
    }
    for (int j = 0; j < 4; ++j)
      for (int q = 0; q < 16; ++q) r += acc[j][q];
  } else {
    f32x4 acc[8];
    for (int j = 0; j < 8; ++j)
      for (int q = 0; q < 4; ++q) acc[j][q] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int c = 0; c < 32; ++c) acc[c & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(c >> 3) & 3], b[c & 3], acc[c & 7], 0, 0, 0);
    }
    for (int j = 0; j < 8; ++j)
      for (int q = 0; q < 4; ++q) r += acc[j][q];
  }
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0) {
    stamp[2 * blockIdx.x] = __builtin_readcyclecounter() - c0;
    stamp[2 * blockIdx.x + 1] = wall_clock64() - w0;
  }
}

template <int SHAPE>
static void run(const char* name, const bf16x8* ops, float* out, unsigned long long* stamp, int waves_per_simd, float seconds) {
  const int grid = 256 * waves_per_simd, iters = 20000;
  const double flop_per_launch = 2.0 * (SHAPE == 32 ? 16.0 * 32 * 32 * 16 : 32.0 * 16 * 16 * 32) * iters * grid * 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  mfma_kernel<SHAPE><<<grid, 256>>>(ops, out, stamp, 200);
  hipDeviceSynchronize();
  // run back to back for `seconds` so that the clock settles, report the last launch
  float ms = 0.f, total = 0.f;
  int n = 0;
  while (total < seconds * 1e3f) {
    hipEventRecord(e0);
    mfma_kernel<SHAPE><<<grid, 256>>>(ops, out, stamp, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    total += ms;
    ++n;
  }
  unsigned long long h[2];
  hipMemcpy(h, stamp + 2 * (grid / 2), sizeof(h), hipMemcpyDeviceToHost);
  const double cyc_per_mfma = (double)h[0] / ((double)iters * (SHAPE == 32 ? 16 : 32)) / waves_per_simd;
  printf("%-28s %d wave(s)/SIMD: %8.1f TFLOP/s, shader clock %6.0f MHz, %5.2f cycles per MFMA per SIMD (%d launches, last %.2f ms)\n", name,
         waves_per_simd, flop_per_launch / ms * 1e-9, 100.0 * h[0] / h[1], cyc_per_mfma, n, ms);
}

int main(int argc, char** argv) {
  const float seconds = argc > 1 ? atof(argv[1]) : 1.0f;
  const size_t n = 8 * 256;
  bf16x8* h = (bf16x8*)malloc(n * sizeof(bf16x8));
  unsigned long long s = 88172645463325252ULL;
  for (size_t i = 0; i < n; ++i)
    for (int j = 0; j < 8; ++j) {
      s ^= s >> 12, s ^= s << 25, s ^= s >> 27;
      const float u = (float)((s * 2685821657736338717ULL) >> 40) * (1.0f / 8388608.0f) - 1.0f;
      h[i][j] = (__bf16)(0.25f * u);
    }
  bf16x8* ops;
  float* out;
  unsigned long long* stamp;
  hipMalloc(&ops, n * sizeof(bf16x8));
  hipMalloc(&out, 256 * 4 * 256 * 4);
  hipMalloc(&stamp, 256 * 4 * 16);
  hipMemcpy(ops, h, n * sizeof(bf16x8), hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep)
    for (int k = 1; k <= 2; ++k) {
      run<32>("32x32x16 bf16, random", ops, out, stamp, k, seconds);
      run<16>("16x16x32 bf16, random", ops, out, stamp, k, seconds);
    }
  hipMemset(ops, 0, n * sizeof(bf16x8));
  run<32>("32x32x16 bf16, zeros", ops, out, stamp, 2, seconds);
  run<16>("16x16x32 bf16, zeros", ops, out, stamp, 2, seconds);
  return 0;
}
