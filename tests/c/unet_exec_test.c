/* Plain-C client of seam B1 (include/mi355x_sd.h, mi355x_sd_unet_*): no torch, no Python -- hipMalloc, the handle API, a file.
 *
 *   gcc -std=c11 -O2 -I/opt/rocm/include -Iinclude tests/c/unet_exec_test.c -Lpaddlemix_amd -lmi355x_sd -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,/opt/rocm/lib -o /tmp/unet_exec_test
 *   LD_LIBRARY_PATH=paddlemix_amd /tmp/unet_exec_test <config.json> <B> <H> <W> <L> <out.bin> [resid_f32] [plan_ex flags]
 *
 * plan_ex flags (MI355X_SD_UNET_ENC_MASK = 1, _CONTROLNET = 4): the optional inputs of mi355x_sd_unet_forward_ex -- the encoder
 * mask keeps the first 3/4 of the text tokens, the ControlNet residuals are LCG numbers (seed 20 + index) scaled by 0.3.
 *
 * Weights and inputs come from a 64-bit LCG so that tests/test_gpu_cexec.py can regenerate them and run the Python-planned model
 * on the same numbers: the two outputs must be bit-identical. Exit code 0 = ran; the comparison is the Python side's job. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mi355x_sd.h"

#define CK(x)                                                                                   \
  do {                                                                                          \
    int rc_ = (x);                                                                              \
    if (rc_) {                                                                                  \
      fprintf(stderr, "%s:%d: %s -> %d: %s\n", __FILE__, __LINE__, #x, rc_, mi355x_sd_last_error()); \
      return 2;                                                                                 \
    }                                                                                           \
  } while (0)
#define HK(x)                                                              \
  do {                                                                     \
    hipError_t e_ = (x);                                                   \
    if (e_ != hipSuccess) {                                                \
      fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      return 3;                                                            \
    }                                                                      \
  } while (0)

static uint64_t g_state;
/* uniform in [-1, 1): top 24 bits of a 64-bit LCG (Knuth's MMIX constants) */
static float lcg_uniform(void) {
  g_state = g_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return (float)((double)(g_state >> 40) / 8388608.0 - 1.0);
}
static void fill(float* p, int64_t n, uint64_t seed, float scale, float offset) {
  g_state = seed;
  for (int64_t i = 0; i < n; ++i) p[i] = offset + scale * lcg_uniform();
}
/* FNV-1a of the parameter name: the per-tensor seed */
static uint64_t name_seed(const char* s) {
  uint64_t h = 1469598103934665603ULL;
  for (; *s; ++s) h = (h ^ (unsigned char)*s) * 1099511628211ULL;
  return h;
}

int main(int argc, char** argv) {
  if (argc < 7) {
    fprintf(stderr, "usage: %s config.json B H W L out.bin [resid_f32] [plan_ex flags]\n", argv[0]);
    return 1;
  }
  const int B = atoi(argv[2]), H = atoi(argv[3]), W = atoi(argv[4]), L = atoi(argv[5]);
  const int resid_f32 = argc > 7 ? atoi(argv[7]) : 0;
  const int flags = argc > 8 ? atoi(argv[8]) : 0;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 1;
  static char json[1 << 16];
  const size_t nj = fread(json, 1, sizeof(json) - 1, f);
  json[nj] = 0;
  fclose(f);

  CK(mi355x_sd_init(0));
  void* h = NULL;
  CK(mi355x_sd_unet_create(json, &h));
  if (resid_f32) CK(mi355x_sd_unet_set_option(h, "residual_f32", 1));

  /* ---- weights: every parameter the handle enumerates, Paddle layouts, fp32 host memory ---- */
  const int np = mi355x_sd_unet_num_params(h);
  int cross_dim = 0, in_ch = 4, out_ch = 4, pdim = 0;
  for (int i = 0; i < np; ++i) {
    const char* name;
    int64_t shp[4];
    int nd;
    CK(mi355x_sd_unet_param_info(h, i, &name, shp, &nd));
    int64_t n = 1;
    for (int d = 0; d < nd; ++d) n *= shp[d];
    float* w = (float*)malloc((size_t)n * 4);
    const size_t ln = strlen(name);
    const int is_bias = ln > 5 && !strcmp(name + ln - 5, ".bias");
    if (is_bias) fill(w, n, name_seed(name), 0.03f, 0.0f);
    else if (nd == 1) fill(w, n, name_seed(name), 0.03f, 1.0f);                      /* norm gamma */
    else if (nd == 2) fill(w, n, name_seed(name), 1.7f / sqrtf((float)shp[0]), 0.0f); /* Linear [in, out] */
    else fill(w, n, name_seed(name), 1.7f / sqrtf((float)(shp[1] * shp[2] * shp[3])), 0.0f);
    CK(mi355x_sd_unet_load_weight(h, name, w, shp, nd, MI355X_SD_DTYPE_F32));
    if (strstr(name, "attn2.to_k.weight") && !cross_dim) cross_dim = (int)shp[0];
    if (!strcmp(name, "conv_in.weight")) in_ch = (int)shp[1];
    if (!strcmp(name, "conv_out.weight")) out_ch = (int)shp[0];
    if (!strcmp(name, "add_embedding.linear_1.weight")) pdim = (int)shp[0];
    free(w);
  }
  size_t wbytes = 0, sbytes = 0;
  CK(mi355x_sd_unet_weight_bytes(h, &wbytes));
  void *dw = NULL, *ws = NULL;   /* (ws holds the handle's split-K scratch too since ABI 12) */
  HK(hipMalloc(&dw, wbytes));
  CK(mi355x_sd_unet_finalize_weights(h, dw, wbytes, NULL));
  CK(flags ? mi355x_sd_unet_plan_ex(h, B, H, W, L, flags, &sbytes) : mi355x_sd_unet_plan(h, B, H, W, L, &sbytes));
  HK(hipMalloc(&ws, sbytes));
  CK(mi355x_sd_unet_bind_workspace(h, ws, sbytes));

  /* ---- inputs ---- */
  const int64_t ns = (int64_t)B * in_ch * H * W, ne = (int64_t)B * L * cross_dim, no = (int64_t)B * out_ch * H * W;
  int atd = 0;
  {
    const char* p = strstr(json, "\"addition_time_embed_dim\"");
    if (p && pdim) atd = atoi(strchr(p, ':') + 1);
  }
  const int td = pdim ? pdim - 6 * atd : 0;
  float *hs = (float*)malloc(ns * 4), *he = (float*)malloc(ne * 4), *ho = (float*)malloc(no * 4);
  float *hte = td ? (float*)malloc((size_t)B * td * 4) : NULL, *hti = td ? (float*)malloc((size_t)B * 6 * 4) : NULL;
  fill(hs, ns, 11, 1.7f, 0.0f);
  fill(he, ne, 12, 1.7f, 0.0f);
  if (td) {
    fill(hte, (int64_t)B * td, 13, 1.7f, 0.0f);
    for (int b = 0; b < B; ++b) {
      const float ids[6] = {1024.f, 1024.f, 0.f, 0.f, 1024.f, 1024.f};
      memcpy(hti + b * 6, ids, sizeof(ids));
    }
  }
  const float ht = 501.0f;
  float *ds, *de, *dout, *dt, *dte = NULL, *dti = NULL;
  HK(hipMalloc((void**)&ds, ns * 4));
  HK(hipMalloc((void**)&de, ne * 4));
  HK(hipMalloc((void**)&dout, no * 4));
  HK(hipMalloc((void**)&dt, 4));
  HK(hipMemcpy(ds, hs, ns * 4, hipMemcpyHostToDevice));
  HK(hipMemcpy(de, he, ne * 4, hipMemcpyHostToDevice));
  HK(hipMemcpy(dt, &ht, 4, hipMemcpyHostToDevice));
  if (td) {
    HK(hipMalloc((void**)&dte, (size_t)B * td * 4));
    HK(hipMalloc((void**)&dti, (size_t)B * 6 * 4));
    HK(hipMemcpy(dte, hte, (size_t)B * td * 4, hipMemcpyHostToDevice));
    HK(hipMemcpy(dti, hti, (size_t)B * 6 * 4, hipMemcpyHostToDevice));
  }
  /* ---- optional inputs (plan_ex flags) ---- */
  float* dmask = NULL;
  const float* dres[64];
  const float* dmid = NULL;
  int nres = 0;
  if (flags & MI355X_SD_UNET_ENC_MASK) {
    float* hm = (float*)malloc((size_t)B * L * 4);
    for (int b = 0; b < B; ++b)
      for (int l = 0; l < L; ++l) hm[b * L + l] = l < (3 * L) / 4 ? 1.0f : 0.0f;
    HK(hipMalloc((void**)&dmask, (size_t)B * L * 4));
    HK(hipMemcpy(dmask, hm, (size_t)B * L * 4, hipMemcpyHostToDevice));
    free(hm);
  }
  if (flags & MI355X_SD_UNET_CONTROLNET) {
    nres = mi355x_sd_unet_num_skips(h);
    if (nres < 0 || nres >= 64) return 5;
    for (int i = 0; i <= nres; ++i) {
      int C_, H_, W_;
      CK(mi355x_sd_unet_skip_shape(h, i, &C_, &H_, &W_));
      const int64_t n = (int64_t)B * C_ * H_ * W_;
      float* hr = (float*)malloc(n * 4);
      fill(hr, n, 20 + (uint64_t)i, 0.3f, 0.0f);
      float* dr;
      HK(hipMalloc((void**)&dr, n * 4));
      HK(hipMemcpy(dr, hr, n * 4, hipMemcpyHostToDevice));
      free(hr);
      if (i < nres) dres[i] = dr;
      else dmid = dr;
    }
  }
  hipStream_t st;
  HK(hipStreamCreate(&st));
  /* eager, then graph capture + two replays: all must agree (checked by the Python side on the last result + a flag here) */
  if (flags) CK(mi355x_sd_unet_forward_ex(h, st, ds, dt, de, dte, dti, NULL, dmask, NULL, nres ? dres : NULL, nres, dmid, dout, 0));
  else CK(mi355x_sd_unet_forward(h, st, ds, dt, de, dte, dti, NULL, dout, 0));
  HK(hipStreamSynchronize(st));
  HK(hipMemcpy(ho, dout, no * 4, hipMemcpyDeviceToHost));
  float* ho2 = (float*)malloc(no * 4);
  for (int rep = 0; rep < 3; ++rep) {
    if (flags) CK(mi355x_sd_unet_forward_ex(h, st, ds, dt, de, dte, dti, NULL, dmask, NULL, nres ? dres : NULL, nres, dmid, dout, 1));
    else CK(mi355x_sd_unet_forward(h, st, ds, dt, de, dte, dti, NULL, dout, 1));
  }
  HK(hipStreamSynchronize(st));
  HK(hipMemcpy(ho2, dout, no * 4, hipMemcpyDeviceToHost));
  if (memcmp(ho, ho2, no * 4)) {
    fprintf(stderr, "graph replay differs from the eager launches\n");
    return 4;
  }
  f = fopen(argv[6], "wb");
  if (!f) return 1;
  fwrite(ho, 4, no, f);
  fclose(f);
  printf("ok: %d parameters, %zu weight bytes, %zu workspace bytes, %d launches, out[0]=%g\n", np, wbytes, sbytes,
         mi355x_sd_unet_num_launches(h), ho[0]);
  CK(mi355x_sd_unet_destroy(h));
  return 0;
}
