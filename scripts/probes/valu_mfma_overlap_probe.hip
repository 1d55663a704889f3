// What bounds the d = 64 attention loop: do v_exp_f32 / plain VALU work and MFMAs of ONE SIMD overlap, and what does a wave64
// v_exp_f32 cost? Per loop iteration a wave issues the instruction mix of one 64-key tile of attention_il_kernel
// (16 x v_mfma_f32_32x32x16_bf16, 32 x v_exp_f32, 32 x v_add_f32, 16 x v_cvt_pk_bf16_f32) or a part of it, no memory traffic.
// Grid = 256 CUs x k blocks of 4 waves (k = waves per SIMD). Output: nanoseconds and shader cycles (s_memtime) per iteration.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/valu_mfma_overlap_probe.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>

#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// MODE bit 0: MFMAs; bit 1: exponentials; bit 2: adds + converts; bit 3: exponentials replaced by v_fma_f32 (a plain VALU op)
template <int MODE>
__global__ __launch_bounds__(256, 2) void mix_kernel(float* out, unsigned long long* cyc, int iters, float seed) {
  f32x16 acc[4];
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) a[i] = (__bf16)(seed + i), b[i] = (__bf16)(seed - i);
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = seed * r;
  float s[32], psum = 0.f;
  unsigned pk[16];
  for (int r = 0; r < 32; ++r) s[r] = seed * 0.001f * (r + threadIdx.x);
  for (int r = 0; r < 16; ++r) pk[r] = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {   // one MFMA per chunk of 2 exp + 2 add + 1 cvt
      if (MODE & 1) acc[c & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c & 3], 0, 0, 0);
      asm volatile("" : "+v"(s[2 * c]), "+v"(s[2 * c + 1]));   // opaque: nothing is hoisted out of the loop, no instruction is added
      float e0 = s[2 * c], e1 = s[2 * c + 1];
      if (MODE & 2) {
        e0 = __builtin_amdgcn_exp2f(e0);
        e1 = __builtin_amdgcn_exp2f(e1);
      }
      if (MODE & 8) {
        e0 = __builtin_fmaf(e0, 0.999f, 0.0001f);
        e1 = __builtin_fmaf(e1, 0.999f, 0.0001f);
      }
      if (MODE & 4) {
        psum += e0;
        psum += e1;
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        const bf16x2 p2 = {(__bf16)e0, (__bf16)e1};
        pk[c] = __builtin_bit_cast(unsigned, p2);
      }
      asm volatile("" ::"v"(e0), "v"(e1), "v"(pk[c]));   // results stay live
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float r = psum;
  for (int j = 0; j < 4; ++j)
    for (int q = 0; q < 16; ++q) r += acc[j][q];
  for (int q = 0; q < 32; ++q) r += s[q];
  for (int q = 0; q < 16; ++q) r += (float)pk[q];
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, float* out, unsigned long long* cyc, int k) {
  const int iters = 4000, grid = 256 * k;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  mix_kernel<MODE><<<grid, 256>>>(out, cyc, 100, 1.0f);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    mix_kernel<MODE><<<grid, 256>>>(out, cyc, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  unsigned long long h[8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  // each SIMD runs k waves: k wave-iterations per SIMD per `iteration time`
  printf("%-34s %d wave(s)/SIMD: %8.1f ns per iteration per wave, %8.1f ns per wave-iteration per SIMD, s_memtime %8.1f per iteration\n", name, k,
         1e6 * best / iters, 1e6 * best / iters / k, (double)h[0] / iters);
}

int main() {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 256 * 3 * 256 * 4);
  hipMalloc(&cyc, 256 * 3 * 8);
  for (int k = 1; k <= 3; ++k) {
    run<1>("16 MFMA", out, cyc, k);
    run<2>("32 exp", out, cyc, k);
    run<8>("32 fma", out, cyc, k);
    run<4>("32 add + 16 cvt", out, cyc, k);
    run<6>("32 exp + 32 add + 16 cvt", out, cyc, k);
    run<7>("16 MFMA + 32 exp + 32 add + 16 cvt", out, cyc, k);
    run<3>("16 MFMA + 32 exp", out, cyc, k);
    run<9>("16 MFMA + 32 fma", out, cyc, k);
    run<13>("16 MFMA + 32 fma + 32 add + 16 cvt", out, cyc, k);
  }
  return 0;
}
