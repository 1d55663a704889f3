// Common device helpers for the MI355X (gfx950 / CDNA4) Stable-Diffusion denoising kernels.
// Written for wave64 + MFMA; no CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace sd {

// The 16-bit storage / MFMA operand type of this build. The default library (libmi355x_sd.so) is bfloat16 -- the dtype
// BASELINE.json's configurations name; `-DMI355X_SD_F16` compiles the SAME kernels for IEEE half
// (libmi355x_sd_f16.so): identical MFMA rates and byte counts, 3 more mantissa bits in every stored activation and
// weight -> whole-UNet parity 1.3-2e-3 instead of 1e-2, at fp16's range (65504). The kernels keep
// the historical name `bf16` for "the element type of the build"; mi355x_sd_elem_dtype() reports which one a library is.
#ifdef MI355X_SD_F16
typedef _Float16 elem16_t;
#else
typedef __bf16 elem16_t;
#endif
typedef elem16_t bf16;
typedef __attribute__((ext_vector_type(2))) elem16_t bf16x2;
typedef __attribute__((ext_vector_type(4))) elem16_t bf16x4;
typedef __attribute__((ext_vector_type(8))) elem16_t bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int kWave = 64;

// A/B switches of the launchers (environment variables MI355X_SD_*: tile overrides, "do it the other way" forms that tests compare
// bit for bit, time stamps). They exist in the DEBUG-SWITCH build only (libmi355x_sd_dbg.so, -DMI355X_SD_DEBUG_SWITCHES: what
// tests/test_gpu_switches.py, tests/test_gpu_gemm_variants.py and the A/B scripts load); the production libraries never read the
// environment -- a kernel choice there depends on the arguments alone.
inline const char* sd_switch(const char* name) {
#ifdef MI355X_SD_DEBUG_SWITCHES
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// the two MFMA shapes the kernels use, on the build's element type (same issue rate for bf16 and f16)
__device__ __forceinline__ f32x4 mfma_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
#ifdef MI355X_SD_F16
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}
// transposing LDS read (ds_read_b64_tr_b16): 4 elements of the build's type
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
__device__ __forceinline__ bf16x4 ds_read_tr16(lds_bf16x4* p) {   // (the instruction moves 16-bit lanes; the type is irrelevant)
  typedef __attribute__((ext_vector_type(4))) __bf16 raw4;
  typedef __attribute__((address_space(3))) raw4 lds_raw4;
  return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_raw4*)p));
}
__device__ __forceinline__ f32x16 mfma_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
#ifdef MI355X_SD_F16
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}

// fp32 += the dot product of two packed element pairs (v_dot2c_f32_bf16 / v_dot2c_f32_f16): the VALU form of a short
// contraction for the few outputs that are not worth an MFMA tile (conv_out's 4 channels) -- no unpack / convert per element
__device__ __forceinline__ float dot2_acc(uint32_t a, uint32_t b, float c) {
#ifdef MI355X_SD_F16
  typedef __attribute__((ext_vector_type(2))) _Float16 pair_t;
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(pair_t, a), __builtin_bit_cast(pair_t, b), c, false);
#else
  typedef __attribute__((ext_vector_type(2))) __bf16 pair_t;
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(pair_t, a), __builtin_bit_cast(pair_t, b), c, false);
#endif
}

__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }  // RNE; lowers to v_cvt_pk_bf16_f32

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  bf16x2 p = {(bf16)lo, (bf16)hi};
  return *reinterpret_cast<uint32_t*>(&p);
}

// x * sigmoid(x); v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division: results are stored as bf16
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// erf-GELU (paddle F.gelu(approximate=False)) = x * Phi(x), evaluated as x * sigmoid(p(x)) with p an odd degree-5
// polynomial fitted (minimax, scripts/fit_gelu.py) to logit(Phi): |error| <= 2.6e-5 absolute on all of R -- 19x below
// the half-ulp of the fp16 the value is stored in (75x for bf16), where the tanh form of the same cost is off by
// 4.7e-4. The coefficients carry the factor -log2(e), so the sigmoid is one v_exp_f32 and one v_rcp_f32: 4 full-rate
// and 2 quarter-rate VALU instructions in all. History: libm erff (ulp-accurate, branchy) made the GEGLU epilogue ~9 % of
// the FF1 GEMM (13.7 -> 12.3 ms per SDXL step with the branch-free Abramowitz-Stegun 7.1.26 erfc form, ~25 instruction
// slots); this form halves the slots again but FF1 stayed at 12.4 ms -- the epilogue arithmetic is no longer what it waits on.
// The argument of p is clamped to +-8 (beyond, Phi is 0 / 1 to 1e-15 and the fitted quintic would eventually turn over).
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -8.0f, 8.0f);
  const float t = xc * xc;
  const float q = __builtin_fmaf(__builtin_fmaf(0.0010127116f, t, -0.10676638f), t, -2.3011315f) * xc;   // -log2(e) * p(x)
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(q));
}

// tanh-GELU (paddle F.gelu(approximate=True)): 0.5 x (1 + tanh(u)) = x * sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3)
__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  return x * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * u));
}

// 8 consecutive values of a modulation / affine vector: fp32 (M16 = false: this library's own programs keep them in fp32) or the
// build's 16-bit element type (M16 = true: what the reference's fused ops receive -- gate / scale / shift are chunks of a 16-bit
// linear output, weight / bias are 16-bit parameters; paddlemix/triton_ops/triton_ops.py:777-786)
template <bool M16>
__device__ __forceinline__ void load_mod8(const void* base, size_t idx, float (&o)[8]) {
  if constexpr (M16) {
    const u32x4 raw = *reinterpret_cast<const u32x4*>(static_cast<const bf16*>(base) + idx);
    const bf16x8 t = *reinterpret_cast<const bf16x8*>(&raw);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (float)t[j];
  } else {
    const float* f = static_cast<const float*>(base) + idx;
    const f32x4 a = *reinterpret_cast<const f32x4*>(f), b = *reinterpret_cast<const f32x4*>(f + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = a[j];
      o[4 + j] = b[j];
    }
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Bijective XCD-aware remap of a 1-D block id: hardware places block b on XCD b % 8; give every XCD a
// contiguous chunk of the logical id space so neighbouring tiles share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// Implicit-GEMM conv: (tap, first channel) of the 64-wide K tile number kt, and the step to the next K tile, for the two K
// orders (GemmArgs::kb64). cg8 = this lane's channel offset inside the K tile.
__device__ __forceinline__ void conv_k_init(int kb64, int kt, int cg8, int Cin, int& tap, int& ch) {
  if (kb64) {
    const int cb = kt / 9;
    tap = kt - cb * 9;
    ch = cb * 64 + cg8;
  } else {
    ch = kt * 64 + cg8;
    tap = ch / Cin;
    ch -= tap * Cin;
  }
}
__device__ __forceinline__ void conv_k_next(int kb64, int Cin, int& tap, int& ch) {
  if (kb64) {
    if (++tap == 9) {
      tap = 0;
      ch += 64;
    }
  } else {
    ch += 64;
    while (ch >= Cin) {
      ch -= Cin;
      ++tap;
    }
  }
}

// Logical tile id -> (tile_m, tile_n), "grouped" order: ids sweep GM row-tiles first, then the column, so the
// tiles an XCD runs concurrently (a contiguous id range after xcd_remap) form a compact 2-D patch that shares
// A row-panels and W column-panels in that XCD's L2. gm > 0: groups of gm row-tiles (round 1: 8); gm < 0: the transposed
// order, groups of -gm column-tiles swept over the rows (production: -4, gemm.hip gemm_gm(); -8 for the 256 x 160 family).
__device__ __forceinline__ void tile_coords(int lid, int ntm, int ntn, int gm, int& tile_m, int& tile_n) {
  if (gm == 0) gm = -4;
  if (gm > 0) {
    const int per_group = gm * ntn;
    const int g = lid / per_group;
    const int first_m = g * gm;
    const int gsz = min(ntm - first_m, gm);
    const int r = lid - g * per_group;
    tile_n = r / gsz;
    tile_m = first_m + (r - tile_n * gsz);
  } else {
    const int gn = -gm;
    const int per_group = gn * ntm;
    const int g = lid / per_group;
    const int first_n = g * gn;
    const int gsz = min(ntn - first_n, gn);
    const int r = lid - g * per_group;
    tile_m = r / gsz;
    tile_n = first_n + (r - tile_m * gsz);
  }
}

}  // namespace sd

// status codes of the C ABI (include/mi355x_sd.h)
#define SD_OK 0
#define SD_ERR_INVALID 1
#define SD_ERR_UNSUPPORTED 2
#define SD_ERR_HIP 3
