/* The attention launches of one SDXL 1024^2 bs-8 step through mi355x_sd_sdpa / mi355x_sd_sdpa_ex from plain C (no torch in the
 * process). Self-attention as the UNet plan issues it: q, k, v are the three thirds of one fused QKV buffer [B, S, 3 * heads * 64]
 * (token stride 3C), the softmax scale folded into the queries (MI355X_SD_SDPA_LOG2); cross-attention: q [B, S, C], k / v halves of
 * one [B, 77, 2C] buffer, scale 64^-0.5. Per shape: launches per step (profiles/r04_g_per_shape_ms.txt), microseconds per launch over
 * back-to-back launches rotating through four input / output sets, TFLOP/s (4 * B * H * Sq * Skv * D), FNV-1a of the output.
 *
 *   gcc -std=c11 -O2 -I/opt/rocm/include -Iinclude scripts/c/attn_probe.c -Lpaddlemix_amd -lmi355x_sd -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,/opt/rocm/lib -o /tmp/attn_probe
 *   LD_LIBRARY_PATH=paddlemix_amd /tmp/attn_probe [reps=20] */
#include "probe_common.h"

typedef struct {
  int B, H, Sq, Skv, D, self, per_step;
} Shape;
static const Shape SHAPES[] = {
    {8, 10, 4096, 4096, 64, 1, 10}, {8, 20, 1024, 1024, 64, 1, 60}, {8, 20, 1024, 77, 64, 0, 60}, {8, 10, 4096, 77, 64, 0, 10}};
enum { NBUF = 4 };

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  CK(mi355x_sd_init(0));
  const int f16 = mi355x_sd_elem_dtype() == MI355X_SD_ELEM_F16;
  hipStream_t st;
  HK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  HK(hipEventCreate(&e0));
  HK(hipEventCreate(&e1));
  double class_ms = 0.0, class_gflop = 0.0;
  printf("# elem %s, %d launches per shape after 3 warm-up launches, %d-buffer rotation\n", f16 ? "fp16" : "bf16", reps, NBUF);
#ifdef ONLY_FIRST_SHAPE /* (power measurements: one shape, many launches) */
  for (size_t s = 0; s < 1; ++s) {
#else
  for (size_t s = 0; s < sizeof(SHAPES) / sizeof(SHAPES[0]); ++s) {
#endif
    const Shape sh = SHAPES[s];
    const int C = sh.H * sh.D;
    void *Q[NBUF], *KV[NBUF], *O[NBUF];
    /* self: one [B, S, 3C] buffer, queries pre-scaled into exponent units (|q.k| of a few units); cross: q and a [B, Skv, 2C] buffer */
    if (upload16_rot(Q, NBUF, (int64_t)sh.B * sh.Sq * (sh.self ? 3 : 1) * C, sh.self ? 0.6f : 1.7f, f16)) return 3;
    for (int b = 0; b < NBUF; ++b) KV[b] = NULL;
    if (!sh.self && upload16_rot(KV, NBUF, (int64_t)sh.B * sh.Skv * 2 * C, 1.7f, f16)) return 3;
    for (int b = 0; b < NBUF; ++b) HK(hipMalloc(&O[b], (size_t)sh.B * sh.Sq * C * 2));
    for (int i = 0; i < 3 + reps; ++i) {
      if (i == 3) HK(hipEventRecord(e0, st));
      const int b = i % NBUF;
      if (sh.self) {
        const char* base = (const char*)Q[b];
        CK(mi355x_sd_sdpa_ex(base, base + (size_t)C * 2, base + (size_t)C * 4, NULL, O[b], sh.B, sh.H, sh.Sq, sh.Skv, sh.D,
                             (int64_t)sh.Sq * 3 * C, 3 * C, (int64_t)sh.Skv * 3 * C, 3 * C, (int64_t)sh.Skv * 3 * C, 3 * C,
                             (int64_t)sh.Sq * C, C, 0, 0, 0, 1.0f, MI355X_SD_SDPA_LOG2, st));
      } else {
        const char* kv = (const char*)KV[b];
        CK(mi355x_sd_sdpa(Q[b], kv, kv + (size_t)C * 2, NULL, O[b], sh.B, sh.H, sh.Sq, sh.Skv, sh.D, (int64_t)sh.Sq * C, C,
                          (int64_t)sh.Skv * 2 * C, 2 * C, (int64_t)sh.Skv * 2 * C, 2 * C, (int64_t)sh.Sq * C, C, 0, 0, 0,
                          1.0f / sqrtf((float)sh.D), st));
      }
    }
    HK(hipEventRecord(e1, st));
    HK(hipStreamSynchronize(st));
    float ms = 0.f;
    HK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps, gflop = 4.0 * sh.B * sh.H * (double)sh.Sq * sh.Skv * sh.D * 1e-9;
    printf("sdpa %dx%2dx%4dx%4dx%d %s  x%3d/step  %8.2f us  %7.1f TFLOP/s  %7.3f ms/step  out %016llx\n", sh.B, sh.H, sh.Sq, sh.Skv,
           sh.D, sh.self ? "self " : "cross", sh.per_step, us, gflop / us * 1e3, us * sh.per_step * 1e-3,
           (unsigned long long)device_fnv(O[0], (size_t)sh.B * sh.Sq * C * 2));
    if (getenv("MI355X_SD_ATTN_STAMP") && sh.self) { /* debug-switch build (-lmi355x_sd_dbg): the kernel left clock stamps in the last query row */
      unsigned long long d[4];
      HK(hipMemcpy(d, (const char*)Q[0] + (((size_t)(sh.B - 1) * sh.Sq * 3 * C + (size_t)(sh.Sq - 1) * 3 * C + C - 16) * 2), sizeof(d), hipMemcpyDeviceToHost));
      printf("   shader clock while this launch ran: %.0f MHz (block 8: %llu s_memtime ticks in %llu wall ticks of 10 ns), %.0f MHz (a block of the second half)\n",
             100.0 * d[0] / d[1], d[0], d[1], 100.0 * d[2] / d[3]);
      printf("   shader clock, second stamp: %llu s_memtime ticks in %llu wall ticks\n", d[2], d[3]);
    }
    class_ms += us * sh.per_step * 1e-3;
    class_gflop += gflop * sh.per_step;
    for (int b = 0; b < NBUF; ++b) {
      HK(hipFree(Q[b]));
      if (KV[b]) HK(hipFree(KV[b]));
      HK(hipFree(O[b]));
    }
  }
  printf("attention launches of one step: %.3f ms isolated, %.0f GFLOP, %.1f TFLOP/s\n", class_ms, class_gflop, class_gflop / class_ms);
  return 0;
}
