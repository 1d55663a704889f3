// Layout + rate probe of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 x e4m3, unit scales) on gfx950.
// build: hipcc --offload-arch=gfx950 -O3 scripts/probes/f8_mfma_probe.hip -o /tmp/f8probe && /tmp/f8probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;

// OCP e4m3 encode of small integers / halves (exact for |v| <= 8 with step .5 here)
static unsigned char e4m3(float v) {
  if (v == 0.f) return 0;
  unsigned char s = v < 0 ? 0x80 : 0;
  float a = fabsf(v);
  int e = 0;
  while (a >= 2.f) { a *= 0.5f; ++e; }
  while (a < 1.f) { a *= 2.f; --e; }
  int m = (int)((a - 1.f) * 8.f + 0.5f);
  return s | (unsigned char)(((e + 7) << 3) | m);
}

// hypothesis: lane l supplies A[row = l & 15][k = (l >> 4) * 32 + j], j = 0..31 (byte j of the 32-byte operand);
// B likewise with col = l & 15; C[row = (l >> 4) * 4 + r][col = l & 15]
__global__ void layout_kernel(const unsigned char* A, const unsigned char* B, float* C) {
  const int l = threadIdx.x;
  i32x8 a, b;
  memcpy(&a, A + (l & 15) * 128 + (l >> 4) * 32, 32);
  memcpy(&b, B + (l & 15) * 128 + (l >> 4) * 32, 32);
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
  for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

template <int F8>
__global__ void rate_kernel(float* out, int iters) {
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x; b[i] = 0x38383838 ^ threadIdx.x; }
  f32x4 c[8];
  for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
  bf16x8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(float)(threadIdx.x & 7); hb[i] = (__bf16)1.0f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (F8) c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[i], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      else c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, c[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  unsigned char hA[16 * 128], hB[16 * 128];
  float ref[256] = {0};
  float fa[16][128], fb[16][128];
  for (int r = 0; r < 16; ++r)
    for (int k = 0; k < 128; ++k) {
      fa[r][k] = (float)(((r * 7 + k * 3) % 9) - 4) * 0.5f;
      fb[r][k] = (float)(((k * 5 + r * 2) % 7) - 3);
      hA[r * 128 + k] = e4m3(fa[r][k]);
      hB[r * 128 + k] = e4m3(fb[r][k]);
    }
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n)
      for (int k = 0; k < 128; ++k) ref[m * 16 + n] += fa[m][k] * fb[n][k];
  unsigned char *dA, *dB;
  float* dC;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, 256 * 4);
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
  hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  float hC[256];
  hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) bad += fabsf(hC[i] - ref[i]) > 1e-3f;
  printf("layout hypothesis (A row l&15, k (l>>4)*32+j; C row (l>>4)*4+r, col l&15): %s (%d mismatches; C[0][0]=%g ref %g, C[5][9]=%g ref %g)\n",
         bad ? "WRONG" : "OK", bad, hC[0], ref[0], hC[5 * 16 + 9], ref[5 * 16 + 9]);
  float* dOut;
  hipMalloc(&dOut, 256 * 8 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int f8 = 0; f8 < 2; ++f8) {
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (f8) hipLaunchKernelGGL(rate_kernel<1>, dim3(256 * 8), dim3(256), 0, 0, dOut, iters);
      else hipLaunchKernelGGL(rate_kernel<0>, dim3(256 * 8), dim3(256), 0, 0, dOut, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)256 * 8 * 4 * iters * 8 * (f8 ? 2.0 * 16 * 16 * 128 : 2.0 * 16 * 16 * 32);
    printf("%s: %.2f ms, %.0f TFLOP/s (issue-bound peak of this chip)\n", f8 ? "fp8 16x16x128" : "bf16 16x16x32", ms, flop / ms / 1e9);
  }
  return 0;
}
