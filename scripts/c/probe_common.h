/* Shared by the plain-C probes of this directory: error macros, a xorshift generator, 16-bit element conversion for the build in
 * use, device uploads, FNV-1a of a device buffer. Header-only, C11. */
#ifndef SD_PROBE_COMMON_H
#define SD_PROBE_COMMON_H
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

static uint64_t g_s = 88172645463325252ULL;
static inline uint32_t rnd(void) {
  g_s ^= g_s >> 12, g_s ^= g_s << 25, g_s ^= g_s >> 27;
  return (uint32_t)((g_s * 2685821657736338717ULL) >> 32);
}
static inline float uniform1(void) { return (float)(rnd() >> 8) * (1.0f / 8388608.0f) - 1.0f; } /* [-1, 1) */
/* 16-bit element of the build: bf16 = top half of the fp32 pattern (round to nearest even), fp16 by hand (normal range) */
static inline uint16_t to_elem(float x, int f16) {
  uint32_t u;
  memcpy(&u, &x, 4);
  if (!f16) return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  const uint32_t sign = (u >> 16) & 0x8000u;
  const int e = (int)((u >> 23) & 0xff) - 127 + 15;
  if (e <= 0) return (uint16_t)sign;
  if (e >= 31) return (uint16_t)(sign | 0x7bffu);
  return (uint16_t)(sign | ((uint32_t)e << 10) | ((u >> 13) & 0x3ffu));
}
/* n elements uniform in [-scale, scale) in device memory; 0 = ok */
static inline int upload16(void** dev, int64_t n, float scale, int f16) {
  uint16_t* h = (uint16_t*)malloc((size_t)n * 2);
  if (!h) return 1;
  for (int64_t i = 0; i < n; ++i) h[i] = to_elem(scale * uniform1(), f16);
  const int bad = hipMalloc(dev, (size_t)n * 2) != hipSuccess || hipMemcpy(*dev, h, (size_t)n * 2, hipMemcpyHostToDevice) != hipSuccess;
  free(h);
  return bad;
}
/* the same contents at nbuf different addresses (a rotation defeats the caches by address; one host pass instead of nbuf) */
static inline int upload16_rot(void** dev, int nbuf, int64_t n, float scale, int f16) {
  if (upload16(&dev[0], n, scale, f16)) return 1;
  for (int b = 1; b < nbuf; ++b)
    if (hipMalloc(&dev[b], (size_t)n * 2) != hipSuccess || hipMemcpy(dev[b], dev[0], (size_t)n * 2, hipMemcpyDeviceToDevice) != hipSuccess) return 1;
  return 0;
}
static inline int upload32(float** dev, int64_t n, float scale) {
  float* h = (float*)malloc((size_t)n * 4);
  if (!h) return 1;
  for (int64_t i = 0; i < n; ++i) h[i] = scale * uniform1();
  const int bad = hipMalloc((void**)dev, (size_t)n * 4) != hipSuccess || hipMemcpy(*dev, h, (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess;
  free(h);
  return bad;
}
/* FNV-1a of nbytes of device memory (0 on a copy failure) */
static inline uint64_t device_fnv(const void* dev, size_t nbytes) {
  unsigned char* h = (unsigned char*)malloc(nbytes);
  uint64_t hash = 1469598103934665603ULL;
  if (!h || hipMemcpy(h, dev, nbytes, hipMemcpyDeviceToHost) != hipSuccess) {
    free(h);
    return 0;
  }
  for (size_t i = 0; i < nbytes; ++i) hash = (hash ^ h[i]) * 1099511628211ULL;
  free(h);
  return hash;
}
#endif
