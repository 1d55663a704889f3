/* The linear shapes of one SDXL 1024^2 bs-8 step through mi355x_sd_linear from plain C (no torch in the process: a GPU box runs
 * this in seconds). For every shape: the launch count per step (profiles/r04_g_per_shape_ms.txt), microseconds per launch over
 * back-to-back launches that rotate through four A / C buffers (so that no launch finds its operands left in the caches by the one
 * before it), TFLOP/s, and an FNV-1a hash of the output -- two BUILDS of the library whose only difference is a schedule must print
 * the same hashes (profiles/r04_s13_early_prologue_ab.txt). Across tile shapes and loops the hashes agree once
 * MI355X_SD_GEMM_NO_BIAS_ACC=1 is set (every tile accumulates K in the same order, but the pipelined loops start their
 * accumulators at the bias where the generic loop adds it last: tests/test_gpu_gemm_variants.py compares them that way).
 *
 *   gcc -std=c11 -O2 -I/opt/rocm/include -Iinclude scripts/c/gemm_probe.c -Lpaddlemix_amd -lmi355x_sd -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,/opt/rocm/lib -o /tmp/gemm_probe
 *   LD_LIBRARY_PATH=paddlemix_amd /tmp/gemm_probe [reps=20]
 * Isolated launches run at a higher clock than the same kernels inside the step: use this for A/B
 * between variants, bench.py / scripts/c/step_bench.c for what a change is worth in the step. */
#define _POSIX_C_SOURCE 200809L
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mi355x_sd.h"

#define CK(x)                                                                                         \
  do {                                                                                                \
    int rc_ = (x);                                                                                    \
    if (rc_) {                                                                                        \
      fprintf(stderr, "%s:%d: %s -> %d: %s\n", __FILE__, __LINE__, #x, rc_, mi355x_sd_last_error()); \
      return 2;                                                                                       \
    }                                                                                                 \
  } while (0)
#define HK(x)                                                                              \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      return 3;                                                                            \
    }                                                                                      \
  } while (0)

typedef struct {
  int M, N, K, geglu, resid, per_step;
} Shape;
/* N of a GEGLU row is the interleaved [value | gate] width (the output has N / 2 columns) */
static const Shape SHAPES[] = {
    {8192, 10240, 1280, 1, 0, 60}, {8192, 1280, 1280, 0, 1, 192}, {8192, 3840, 1280, 0, 0, 60}, {8192, 1280, 5120, 0, 1, 60},
    {32768, 5120, 640, 1, 0, 10},  {32768, 640, 640, 0, 1, 40},    {32768, 1920, 640, 0, 0, 10}, {32768, 640, 2560, 0, 1, 10},
    {131072, 320, 640, 0, 0, 2},   {8192, 1280, 2560, 0, 0, 2},    {131072, 320, 960, 0, 0, 1},
};
enum { NBUF = 4 };

static uint64_t g_s = 88172645463325252ULL;
static uint32_t rnd(void) {
  g_s ^= g_s >> 12, g_s ^= g_s << 25, g_s ^= g_s >> 27;
  return (uint32_t)((g_s * 2685821657736338717ULL) >> 32);
}
/* 16-bit elements of the build, uniform in [-scale, scale): bf16 = top half of the fp32 pattern (rounded), fp16 by hand */
static uint16_t to_elem(float x, int f16) {
  uint32_t u;
  memcpy(&u, &x, 4);
  if (!f16) return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  const uint32_t sign = (u >> 16) & 0x8000u;
  const int e = (int)((u >> 23) & 0xff) - 127 + 15;
  if (e <= 0) return (uint16_t)sign;
  if (e >= 31) return (uint16_t)(sign | 0x7bffu);
  return (uint16_t)(sign | ((uint32_t)e << 10) | ((u >> 13) & 0x3ffu));
}
static int upload16(void** dev, int64_t n, float scale, int f16) {
  uint16_t* h = (uint16_t*)malloc((size_t)n * 2);
  if (!h) return 1;
  for (int64_t i = 0; i < n; ++i) h[i] = to_elem(scale * ((float)(rnd() >> 8) * (1.0f / 8388608.0f) - 1.0f), f16);
  if (hipMalloc(dev, (size_t)n * 2) != hipSuccess || hipMemcpy(*dev, h, (size_t)n * 2, hipMemcpyHostToDevice) != hipSuccess) return 1;
  free(h);
  return 0;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  CK(mi355x_sd_init(0));
  const int f16 = mi355x_sd_elem_dtype() == MI355X_SD_ELEM_F16;
  void* splitk = NULL;
  HK(hipMalloc(&splitk, 64u << 20));
  CK(mi355x_sd_set_workspace(splitk, 64u << 20));
  hipStream_t st;
  HK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  HK(hipEventCreate(&e0));
  HK(hipEventCreate(&e1));
  double class_ms = 0.0, class_gflop = 0.0;
  printf("# elem %s, %d launches per shape after 3 warm-up launches, %d-buffer rotation\n", f16 ? "fp16" : "bf16", reps, NBUF);
  for (size_t s = 0; s < sizeof(SHAPES) / sizeof(SHAPES[0]); ++s) {
    const Shape sh = SHAPES[s];
    const int Nout = sh.geglu ? sh.N / 2 : sh.N;
    void *A[NBUF], *C[NBUF], *Wt = NULL, *R = NULL;
    float* bias = NULL;
    for (int b = 0; b < NBUF; ++b) {
      if (upload16(&A[b], (int64_t)sh.M * sh.K, 1.0f, f16)) return 3;
      HK(hipMalloc(&C[b], (size_t)sh.M * Nout * 2));
    }
    if (upload16(&Wt, (int64_t)sh.N * sh.K, 1.7f / sqrtf((float)sh.K), f16)) return 3;
    if (sh.resid && upload16(&R, (int64_t)sh.M * Nout, 1.0f, f16)) return 3;
    {
      float* hb = (float*)malloc((size_t)sh.N * 4);
      for (int i = 0; i < sh.N; ++i) hb[i] = 0.03f * ((float)(rnd() >> 8) * (1.0f / 8388608.0f) - 1.0f);
      HK(hipMalloc((void**)&bias, (size_t)sh.N * 4));
      HK(hipMemcpy(bias, hb, (size_t)sh.N * 4, hipMemcpyHostToDevice));
      free(hb);
    }
    for (int i = 0; i < 3 + reps; ++i) {
      if (i == 3) HK(hipEventRecord(e0, st));
      CK(mi355x_sd_linear(A[i % NBUF], sh.K, Wt, C[i % NBUF], Nout, sh.M, sh.N, sh.K, bias, NULL, 0, 0, R, Nout, 1.0f,
                          sh.geglu ? MI355X_SD_GEGLU : 0, st));
    }
    HK(hipEventRecord(e1, st));
    HK(hipStreamSynchronize(st));
    float ms = 0.f;
    HK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps, gflop = 2.0 * sh.M * (double)sh.N * sh.K * 1e-9;
    uint64_t hash = 1469598103934665603ULL;
    {
      const size_t nb = (size_t)sh.M * Nout * 2;
      unsigned char* hc = (unsigned char*)malloc(nb);
      HK(hipMemcpy(hc, C[0], nb, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < nb; ++i) hash = (hash ^ hc[i]) * 1099511628211ULL;
      free(hc);
    }
    printf("linear %6dx%5dx%4d%s%s  x%3d/step  %8.2f us  %7.1f TFLOP/s  %7.3f ms/step  out %016llx\n", sh.M, sh.N, sh.K,
           sh.geglu ? "g" : " ", sh.resid ? "+R" : "  ", sh.per_step, us, gflop / us * 1e3, us * sh.per_step * 1e-3,
           (unsigned long long)hash);
    class_ms += us * sh.per_step * 1e-3;
    class_gflop += gflop * sh.per_step;
    for (int b = 0; b < NBUF; ++b) {
      HK(hipFree(A[b]));
      HK(hipFree(C[b]));
    }
    HK(hipFree(Wt));
    HK(hipFree(bias));
    if (R) HK(hipFree(R));
  }
  printf("linear shapes of one step: %.3f ms isolated, %.0f GFLOP, %.1f TFLOP/s\n", class_ms, class_gflop, class_gflop / class_ms);
  return 0;
}
