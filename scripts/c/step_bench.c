/* The denoising step timed from plain C through seam B1 (include/mi355x_sd.h, mi355x_sd_unet_*): no Python, no torch in the
 * process -- what a compiled host sees, and a measurement that costs a GPU box seconds instead of the minute or two a first
 * `import torch` takes there.
 *
 *   gcc -std=c11 -O2 -I/opt/rocm/include -Iinclude scripts/c/step_bench.c -Lpaddlemix_amd -lmi355x_sd -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,/opt/rocm/lib -o /tmp/step_bench
 *   LD_LIBRARY_PATH=paddlemix_amd /tmp/step_bench scripts/c/sdxl_unet_config.json 8 128 128 77 [steps=30] [warmup=3] [eager]
 *
 * One step = one mi355x_sd_unet_forward (hipGraph replay unless "eager") + the scheduler's axpby on the latents, as in bench.py's
 * timed region; weights are uniform numbers at the scale of a trained layer (1.7 / sqrt(fan_in)), inputs are resident in HBM.
 * Prints one JSON line. Random-init weights, synthetic inputs: a throughput number, never a parity statement (the parity of this
 * executor against the Python-planned program is tests/test_gpu_cexec.py: bit-identical). */
#define _POSIX_C_SOURCE 200809L
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

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

/* xorshift64*: two 24-bit uniforms per state, [-1, 1) */
static uint64_t g_s = 88172645463325252ULL;
static void fill(float* p, int64_t n, float scale, float offset) {
  int64_t i = 0;
  for (; i + 1 < n; i += 2) {
    g_s ^= g_s >> 12, g_s ^= g_s << 25, g_s ^= g_s >> 27;
    const uint64_t r = g_s * 2685821657736338717ULL;
    p[i] = offset + scale * ((float)(r >> 40) * (1.0f / 8388608.0f) - 1.0f);
    p[i + 1] = offset + scale * ((float)((r >> 16) & 0xffffff) * (1.0f / 8388608.0f) - 1.0f);
  }
  if (i < n) {
    g_s ^= g_s >> 12, g_s ^= g_s << 25, g_s ^= g_s >> 27;
    p[i] = offset + scale * ((float)((g_s * 2685821657736338717ULL) >> 40) * (1.0f / 8388608.0f) - 1.0f);
  }
}
static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static int upload(float** dev, int64_t n, float scale) {
  float* h = (float*)malloc((size_t)n * 4);
  if (!h) return 1;
  fill(h, n, scale, 0.0f);
  if (hipMalloc((void**)dev, (size_t)n * 4) != hipSuccess) return 1;
  if (hipMemcpy(*dev, h, (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess) return 1;
  free(h);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 6) {
    fprintf(stderr, "usage: %s config.json B H W L [steps] [warmup] [eager]\n", argv[0]);
    return 1;
  }
  const int B = atoi(argv[2]), H = atoi(argv[3]), W = atoi(argv[4]), L = atoi(argv[5]);
  const int steps = argc > 6 ? atoi(argv[6]) : 30, warmup = argc > 7 ? atoi(argv[7]) : 3;
  const int use_graph = !(argc > 8 && !strcmp(argv[8], "eager"));
  FILE* f = fopen(argv[1], "rb");
  if (!f) {
    perror(argv[1]);
    return 1;
  }
  static char json[1 << 16];
  json[fread(json, 1, sizeof(json) - 1, f)] = 0;
  fclose(f);

  const double t0 = now_s();
  CK(mi355x_sd_init(0));
  void* h = NULL;
  CK(mi355x_sd_unet_create(json, &h));
  const int np = mi355x_sd_unet_num_params(h);
  int cross_dim = 0, in_ch = 4, out_ch = 4, pdim = 0;
  int64_t n_param = 0, cap = 0;
  float* w = NULL;
  for (int i = 0; i < np; ++i) {
    const char* name;
    int64_t shp[4];
    int nd;
    CK(mi355x_sd_unet_param_info(h, i, &name, shp, &nd));
    int64_t n = 1;
    for (int d = 0; d < nd; ++d) n *= shp[d];
    if (n > cap) {
      free(w);
      w = (float*)malloc((size_t)(cap = n) * 4);
      if (!w) return 1;
    }
    const size_t ln = strlen(name);
    if (ln > 5 && !strcmp(name + ln - 5, ".bias")) fill(w, n, 0.03f, 0.0f);
    else if (nd == 1) fill(w, n, 0.03f, 1.0f); /* norm gamma */
    else if (nd == 2) fill(w, n, 1.7f / sqrtf((float)shp[0]), 0.0f); /* Linear [in, out] */
    else fill(w, n, 1.7f / sqrtf((float)(shp[1] * shp[2] * shp[3])), 0.0f); /* conv OIHW */
    CK(mi355x_sd_unet_load_weight(h, name, w, shp, nd, MI355X_SD_DTYPE_F32));
    if (strstr(name, "attn2.to_k.weight") && !cross_dim) cross_dim = (int)shp[0];
    if (!strcmp(name, "conv_in.weight")) in_ch = (int)shp[1];
    if (!strcmp(name, "conv_out.weight")) out_ch = (int)shp[0];
    if (!strcmp(name, "add_embedding.linear_1.weight")) pdim = (int)shp[0];
    n_param += n;
  }
  free(w);
  size_t wbytes = 0, sbytes = 0;
  CK(mi355x_sd_unet_weight_bytes(h, &wbytes));
  void *dw = NULL, *ws = NULL;   /* (ws holds the handle's split-K scratch too since ABI 12) */
  HK(hipMalloc(&dw, wbytes));
  CK(mi355x_sd_unet_finalize_weights(h, dw, wbytes, NULL));
  CK(mi355x_sd_unet_plan(h, B, H, W, L, &sbytes));
  HK(hipMalloc(&ws, sbytes));
  CK(mi355x_sd_unet_bind_workspace(h, ws, sbytes));

  int atd = 0;
  const char* p = strstr(json, "\"addition_time_embed_dim\"");
  if (p && pdim) atd = atoi(strchr(p, ':') + 1);
  const int td = pdim ? pdim - 6 * atd : 0;
  const int64_t ns = (int64_t)B * in_ch * H * W, no = (int64_t)B * out_ch * H * W;
  float *ds, *de, *dout, *dt, *dcoef, *dte = NULL, *dti = NULL;
  if (upload(&ds, ns, 1.7f) || upload(&de, (int64_t)B * L * cross_dim, 1.7f)) return 3;
  HK(hipMalloc((void**)&dout, (size_t)no * 4));
  const float ht = 501.0f, coef[2] = {1.0f, -0.05f}; /* latents <- 1 * latents - 0.05 * eps: an Euler step's shape */
  HK(hipMalloc((void**)&dt, 4));
  HK(hipMalloc((void**)&dcoef, 8));
  HK(hipMemcpy(dt, &ht, 4, hipMemcpyHostToDevice));
  HK(hipMemcpy(dcoef, coef, 8, hipMemcpyHostToDevice));
  if (td) {
    if (upload(&dte, (int64_t)B * td, 1.7f)) return 3;
    float* hti = (float*)malloc((size_t)B * 6 * 4);
    for (int b = 0; b < B; ++b) {
      const float ids[6] = {1024.f, 1024.f, 0.f, 0.f, 1024.f, 1024.f};
      memcpy(hti + b * 6, ids, sizeof(ids));
    }
    HK(hipMalloc((void**)&dti, (size_t)B * 6 * 4));
    HK(hipMemcpy(dti, hti, (size_t)B * 6 * 4, hipMemcpyHostToDevice));
    free(hti);
  }
  hipStream_t st;
  HK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  HK(hipEventCreate(&e0));
  HK(hipEventCreate(&e1));
  const double t_setup = now_s() - t0;

  for (int i = 0; i < warmup + steps; ++i) {
    if (i == warmup) {
      HK(hipStreamSynchronize(st));
      HK(hipEventRecord(e0, st));
    }
    CK(mi355x_sd_unet_forward(h, st, ds, dt, de, dte, dti, NULL, dout, use_graph));
    if (in_ch == out_ch) CK(mi355x_sd_axpby(ds, dout, ds, dcoef, ns, st));
  }
  HK(hipEventRecord(e1, st));
  HK(hipStreamSynchronize(st));
  float ms = 0.f;
  HK(hipEventElapsedTime(&ms, e0, e1));
  /* FNV-1a of the last prediction: two builds of the library that must agree bit for bit (schedule changes) print the same value */
  uint64_t hash = 1469598103934665603ULL;
  {
    float* ho = (float*)malloc((size_t)no * 4);
    HK(hipMemcpy(ho, dout, (size_t)no * 4, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < no; ++i) {
      if (!isfinite(ho[i])) {
        fprintf(stderr, "non-finite output at %lld\n", (long long)i);
        return 4;
      }
    }
    const unsigned char* hb = (const unsigned char*)ho;
    for (size_t i = 0; i < (size_t)no * 4; ++i) hash = (hash ^ hb[i]) * 1099511628211ULL;
    free(ho);
  }
  printf("{\"bench\": \"c_abi_step\", \"elem\": \"%s\", \"B\": %d, \"H\": %d, \"W\": %d, \"L\": %d, \"steps\": %d, \"warmup\": %d, "
         "\"graph\": %d, \"ms_per_step\": %.4f, \"steps_per_s\": %.4f, \"launches\": %d, \"params\": %lld, \"weight_bytes\": %zu, "
         "\"workspace_bytes\": %zu, \"setup_s\": %.1f, \"out_fnv\": \"%016llx\"}\n",
         mi355x_sd_elem_dtype() == MI355X_SD_ELEM_F16 ? "fp16" : "bf16", B, H, W, L, steps, warmup, use_graph, ms / steps,
         1e3 * steps / ms, mi355x_sd_unet_num_launches(h), (long long)n_param, wbytes, sbytes, t_setup, (unsigned long long)hash);
  CK(mi355x_sd_unet_destroy(h));
  return 0;
}
