/* The 3x3 convolutions of one SDXL 1024^2 bs-8 step through mi355x_sd_conv3x3 from plain C (no torch in the process): NHWC rows in,
 * weights in the 64-channel-block packing the planner uses (MI355X_SD_CONV_KB64), bias + a per-image row bias (the time embedding
 * of a resnet's first conv) in the epilogue. Per shape: launches per step (profiles/r04_g_per_shape_ms.txt), microseconds per launch
 * over back-to-back launches rotating through four input / output sets, TFLOP/s (2 * B * Ho * Wo * Cout * 9 * Cin), FNV-1a of the
 * output.
 *
 *   gcc -std=c11 -O2 -I/opt/rocm/include -Iinclude scripts/c/conv_probe.c -Lpaddlemix_amd -lmi355x_sd -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,/opt/rocm/lib -o /tmp/conv_probe
 *   LD_LIBRARY_PATH=paddlemix_amd /tmp/conv_probe [reps=10] */
#include "probe_common.h"

typedef struct {
  int Hs, Cin, Cout, stride, up, per_step; /* square Hs x Hs input of B = 8 images */
} Shape;
static const Shape SHAPES[] = {
    {32, 1280, 1280, 1, 0, 10}, {128, 320, 320, 1, 0, 7}, {64, 640, 640, 1, 0, 6},  {128, 640, 320, 1, 0, 2}, {64, 640, 640, 1, 1, 1},
    {32, 1280, 1280, 1, 1, 1},  {32, 2560, 1280, 1, 0, 2}, {128, 960, 320, 1, 0, 1}, {64, 1920, 640, 1, 0, 1}, {64, 1280, 640, 1, 0, 1},
    {64, 960, 640, 1, 0, 1},    {32, 1920, 1280, 1, 0, 1}, {64, 320, 640, 1, 0, 1},  {32, 640, 1280, 1, 0, 1}, {64, 640, 640, 2, 0, 1},
    {128, 320, 320, 2, 0, 1},
};
enum { NBUF = 4, B = 8 };

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
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
  printf("# elem %s, %d launches per shape after 2 warm-up launches, %d-buffer rotation\n", f16 ? "fp16" : "bf16", reps, NBUF);
  for (size_t s = 0; s < sizeof(SHAPES) / sizeof(SHAPES[0]); ++s) {
    const Shape sh = SHAPES[s];
    const int Ho = ((sh.Hs << sh.up) + 2 - 3) / sh.stride + 1;
    const int64_t rows_in = (int64_t)B * sh.Hs * sh.Hs, rows_out = (int64_t)B * Ho * Ho;
    void *X[NBUF], *Y[NBUF], *Wt = NULL;
    float *bias = NULL, *rowbias = NULL;
    if (upload16_rot(X, NBUF, rows_in * sh.Cin, 1.0f, f16)) return 3;
    for (int b = 0; b < NBUF; ++b) HK(hipMalloc(&Y[b], (size_t)rows_out * sh.Cout * 2));
    if (upload16(&Wt, (int64_t)sh.Cout * 9 * sh.Cin, 1.7f / sqrtf(9.0f * sh.Cin), f16)) return 3;
    if (upload32(&bias, sh.Cout, 0.03f) || upload32(&rowbias, (int64_t)B * sh.Cout, 0.3f)) return 3;
    for (int i = 0; i < 2 + reps; ++i) {
      if (i == 2) HK(hipEventRecord(e0, st));
      CK(mi355x_sd_conv3x3(X[i % NBUF], sh.Cin, B, sh.Hs, sh.Hs, sh.Cin, sh.stride, sh.up, Wt, Y[i % NBUF], sh.Cout, sh.Cout, bias, rowbias,
                           sh.Cout, NULL, 0, 1.0f, MI355X_SD_CONV_KB64, splitk, 64u << 20, st));
    }
    HK(hipEventRecord(e1, st));
    HK(hipStreamSynchronize(st));
    float ms = 0.f;
    HK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps, gflop = 2.0 * rows_out * (double)sh.Cout * 9.0 * sh.Cin * 1e-9;
    printf("conv3x3 %6lldx%4dx%5d%s%s  x%2d/step  %8.2f us  %7.1f TFLOP/s  %7.3f ms/step  out %016llx\n", (long long)rows_out, sh.Cout,
           9 * sh.Cin, sh.stride == 2 ? "s2" : "  ", sh.up ? "up" : "  ", sh.per_step, us, gflop / us * 1e3, us * sh.per_step * 1e-3,
           (unsigned long long)device_fnv(Y[0], (size_t)rows_out * sh.Cout * 2));
    class_ms += us * sh.per_step * 1e-3;
    class_gflop += gflop * sh.per_step;
    for (int b = 0; b < NBUF; ++b) {
      HK(hipFree(X[b]));
      HK(hipFree(Y[b]));
    }
    HK(hipFree(Wt));
    HK(hipFree(bias));
    HK(hipFree(rowbias));
  }
  printf("3x3 convolutions of one step: %.3f ms isolated, %.0f GFLOP, %.1f TFLOP/s\n", class_ms, class_gflop, class_gflop / class_ms);
  return 0;
}
