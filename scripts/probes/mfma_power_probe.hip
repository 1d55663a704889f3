// Probe (round 4): what do the two bf16 MFMA shapes cost at the wall -- rate, clock and board power -- when nothing but MFMAs runs?
// The GEMM kernels of this library use v_mfma_f32_16x16x32_bf16 (a lane owns 4 consecutive output channels: cheap epilogues); the
// 32x32x16 shape reads half as many operand registers per FLOP and is the one MI355X_MICROARCH.md quotes the 2.5 PFLOP/s peak for.
// The step is power-managed (profiles/HISTORY.md section 5, round 4), so the energy per FLOP of the shape is a throughput question.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe scripts/probes/mfma_power_probe.hip && /tmp/mfma_probe <shape 16|32> <seconds> <fill 0|1>
// Every CU gets one block of 8 waves (2 per SIMD, like the GEMM kernels); a wave cycles through 4 A and 5 B fragments of random (or
// zero) bf16 data and 20 (16x16: 80 registers) / 5 (32x32: 80 registers) independent accumulators. Prints TFLOP/s; run rocm-smi beside it.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void probe(const bf16x8* __restrict__ src, float* __restrict__ out, int iters) {
  const int tid = threadIdx.x + blockIdx.x * 512;
  bf16x8 a[4], b[5];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = src[(size_t)tid * 9 + i];
#pragma unroll
  for (int i = 0; i < 5; ++i) b[i] = src[(size_t)tid * 9 + 4 + i];
  float s = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4 acc[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[i], a[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
  } else {
    f32x16 acc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j)     // 4 x 5 MFMAs of 32x32x16 = 2x the FLOPs of the 20 MFMAs of 16x16x32 above
#pragma unroll
        for (int i = 0; i < 5; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[i], a[j], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) s += acc[i][0] + acc[i][15];
  }
  out[tid] = s;
}

int main(int argc, char** argv) {
  const int shape = argc > 1 ? atoi(argv[1]) : 16;
  const double seconds = argc > 2 ? atof(argv[2]) : 4.0;
  const int fill = argc > 3 ? atoi(argv[3]) : 1;
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const size_t nthreads = (size_t)cus * 512;
  std::vector<unsigned short> h(nthreads * 9 * 8);
  unsigned x = 12345u;
  for (auto& v : h) {
    x = x * 1664525u + 1013904223u;
    // random bf16 in (-1, 1): sign, exponent 119..126, 7 random mantissa bits -- or zeros
    v = fill ? (unsigned short)(((x >> 31) << 15) | ((119 + ((x >> 20) & 7)) << 7) | ((x >> 8) & 127)) : 0;
  }
  bf16x8* src;
  float* out;
  hipMalloc(&src, h.size() * 2);
  hipMalloc(&out, nthreads * 4);
  hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  const int iters = 20000;
  const double flop_per_launch = (shape == 16 ? 20.0 * 2 * 16 * 16 * 32 : 20.0 * 2 * 32 * 32 * 16) * iters * (nthreads / 64.0);
  auto launch = [&]() {
    if (shape == 16) hipLaunchKernelGGL(probe<16>, dim3(cus), dim3(512), 0, 0, src, out, iters);
    else hipLaunchKernelGGL(probe<32>, dim3(cus), dim3(512), 0, 0, src, out, iters);
  };
  launch();
  hipDeviceSynchronize();
  const auto t0 = std::chrono::steady_clock::now();
  int n = 0;
  double el = 0;
  while (el < seconds) {
    launch();
    hipDeviceSynchronize();
    ++n;
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  printf("mfma %dx%d bf16, %s operands: %.1f TFLOP/s over %.1f s (%d launches, %d CUs x 8 waves)\n", shape, shape, fill ? "random" : "zero",
         flop_per_launch * n / el / 1e12, el, n, cus);
  return 0;
}
