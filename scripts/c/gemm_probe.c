/* The linear shapes of one SDXL 1024^2 bs-8 step through mi355x_sd_linear from plain C (no torch in the process: a GPU box runs
 * this in seconds). For every shape: the launch count per step (profiles/r04_g_per_shape_ms.txt), microseconds per launch over
 * back-to-back launches that rotate through four A / C buffers (so that no launch finds its operands left in the caches by the one
 * before it), TFLOP/s, and an FNV-1a hash of the output -- two BUILDS of the library whose only difference is a schedule must print
 * the same hashes (profiles/r04_s13_early_prologue_ab.txt). Across tile shapes and loops the hashes agree once
 * MI355X_SD_GEMM_NO_BIAS_ACC=1 is set (every tile accumulates K in the same order, but the pipelined loops start their
 * accumulators at the bias where the generic loop adds it last: tests/test_gpu_gemm_variants.py compares them that way).
 *
 *   gcc -std=c11 -O2 -I/opt/rocm/include -Iinclude -Iscripts/c scripts/c/gemm_probe.c -Lpaddlemix_amd -lmi355x_sd -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,/opt/rocm/lib -o /tmp/gemm_probe
 *   LD_LIBRARY_PATH=paddlemix_amd /tmp/gemm_probe [reps=20]
 * Isolated launches run at a higher clock than the same kernels inside the step (profiles/HISTORY.md section 5, round 4): use this for A/B
 * between variants, bench.py / scripts/c/step_bench.c for what a change is worth in the step. */
#include "probe_common.h"

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

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  const unsigned mask = argc > 2 ? (unsigned)strtoul(argv[2], NULL, 0) : 0xFFFFFFFFu;   /* bit s: run shape s */
  /* weight copies rotated through (1: every launch multiplies by the same, cache-resident matrix -- what a step never does; enough
   * copies to exceed the 256-MB Infinity Cache: every launch fetches its weights from HBM, like the step's layers) */
  int wrot = argc > 3 ? atoi(argv[3]) : 1;
  if (wrot < 1) wrot = 1;
  if (wrot > 512) wrot = 512;
  CK(mi355x_sd_init(0));
  const int f16 = mi355x_sd_elem_dtype() == MI355X_SD_ELEM_F16;
  void* splitk = NULL;
  HK(hipMalloc(&splitk, 64u << 20));
  hipStream_t st;
  HK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  HK(hipEventCreate(&e0));
  HK(hipEventCreate(&e1));
  double class_ms = 0.0, class_gflop = 0.0;
  printf("# elem %s, %d launches per shape after 3 warm-up launches, %d-buffer rotation\n", f16 ? "fp16" : "bf16", reps, NBUF);
  for (size_t s = 0; s < sizeof(SHAPES) / sizeof(SHAPES[0]); ++s) {
    const Shape sh = SHAPES[s];
    if (!((mask >> s) & 1u)) continue;
    const int Nout = sh.geglu ? sh.N / 2 : sh.N;
    void *A[NBUF], *C[NBUF], *Wt = NULL, *R = NULL;
    void* Wr[512];
    float* bias = NULL;
    if (upload16_rot(A, NBUF, (int64_t)sh.M * sh.K, 1.0f, f16)) return 3;
    for (int b = 0; b < NBUF; ++b) HK(hipMalloc(&C[b], (size_t)sh.M * Nout * 2));
    if (upload16(&Wt, (int64_t)sh.N * sh.K, 1.7f / sqrtf((float)sh.K), f16)) return 3;
    Wr[0] = Wt;
    for (int w = 1; w < wrot; ++w) {   /* identical copies: the output hash does not depend on the rotation */
      HK(hipMalloc(&Wr[w], (size_t)sh.N * sh.K * 2));
      HK(hipMemcpy(Wr[w], Wt, (size_t)sh.N * sh.K * 2, hipMemcpyDeviceToDevice));
    }
    if (sh.resid && upload16(&R, (int64_t)sh.M * Nout, 1.0f, f16)) return 3;
    if (upload32(&bias, sh.N, 0.03f)) return 3;
    for (int i = 0; i < 3 + reps; ++i) {
      if (i == 3) HK(hipEventRecord(e0, st));
      CK(mi355x_sd_linear(A[i % NBUF], sh.K, Wr[i % wrot], C[i % NBUF], Nout, sh.M, sh.N, sh.K, bias, NULL, 0, 0, R, Nout, 1.0f,
                          sh.geglu ? MI355X_SD_GEGLU : 0, splitk, 64u << 20, st));
    }
    HK(hipEventRecord(e1, st));
    HK(hipStreamSynchronize(st));
    float ms = 0.f;
    HK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps, gflop = 2.0 * sh.M * (double)sh.N * sh.K * 1e-9;
    const uint64_t hash = device_fnv(C[0], (size_t)sh.M * Nout * 2);
    printf("linear %6dx%5dx%4d%s%s  x%3d/step  %8.2f us  %7.1f TFLOP/s  %7.3f ms/step  out %016llx\n", sh.M, sh.N, sh.K,
           sh.geglu ? "g" : " ", sh.resid ? "+R" : "  ", sh.per_step, us, gflop / us * 1e3, us * sh.per_step * 1e-3,
           (unsigned long long)hash);
    class_ms += us * sh.per_step * 1e-3;
    class_gflop += gflop * sh.per_step;
    for (int b = 0; b < NBUF; ++b) {
      HK(hipFree(A[b]));
      HK(hipFree(C[b]));
    }
    for (int w = 0; w < wrot; ++w) HK(hipFree(Wr[w]));
    HK(hipFree(bias));
    if (R) HK(hipFree(R));
  }
  printf("linear shapes of one step: %.3f ms isolated, %.0f GFLOP, %.1f TFLOP/s\n", class_ms, class_gflop, class_gflop / class_ms);
  return 0;
}
