// The reference's fused custom ops of the SD3 / DiT path with THEIR signatures (seam B4), as stand-alone HBM-bound kernels:
//   fused_adaLN_scale_residual(x, mha_out, gate_msa, scale_mlp, shift_mlp, weight, bias, epsilon) -> (resi_out, adaLN_out)
//       paddlemix/triton_ops/triton_ops.py:758-920; unfused definition :842-847:
//       resi_out = mha_out * gate[:, None] + x ;  adaLN_out = layer_norm(resi_out, weight, bias, eps) * (1 + scale[:, None]) + shift[:, None]
//   split_concat(x [B,S1,3C], y [B,S2,3C]) -> (q, k, v), each [B, S1+S2, C] = concat(x chunk i, y chunk i) along the sequence
//       triton_ops.py:1692-1752 (kernel :1652-1689)
// Inside this library's own SD3 program the same arithmetic lives in GEMM epilogues / row maps (sd3.py); these entry points exist
// so that a PaddleMIX maintainer can swap the Triton ops one for one.
// One wave per row, 16-byte accesses, fp32 statistics (shifted one-pass, like layernorm_kernel), two outputs written in the pass
// that read the inputs: 2 reads + 2 writes of the row = the op's minimum traffic.
#include "common.h"
#include "kernels.h"

namespace sd {

// M16: gate / scale / shift / weight / bias are the build's 16-bit element type (what the reference passes) instead of fp32
template <int NCH, bool M16>
__global__ __launch_bounds__(256) void fused_adaln_scale_residual_kernel(
    const bf16* __restrict__ x, int ldx, const bf16* __restrict__ mha, int ldm, const void* __restrict__ gate,
    const void* __restrict__ scale, const void* __restrict__ shift, int ld_mod, int rows_per_batch,
    const void* __restrict__ weight, const void* __restrict__ bias, float eps, int rows, int C, bf16* __restrict__ resi,
    int ldr, bf16* __restrict__ out, int ldo) {
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (blockDim.x >> 6);
  const int cv = C >> 3;
  const float invC = 1.0f / (float)C;
  for (int row = wave_g; row < rows; row += nwaves) {
    const size_t mrow = (size_t)(row / rows_per_batch) * ld_mod;
    float v[NCH][8];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int cc = lane + 64 * i;
      if (cc < cv) {
        const u32x4 rx = *reinterpret_cast<const u32x4*>(x + (size_t)row * ldx + cc * 8);
        const u32x4 rm = *reinterpret_cast<const u32x4*>(mha + (size_t)row * ldm + cc * 8);
        const bf16x8 tx = *reinterpret_cast<const bf16x8*>(&rx), tm = *reinterpret_cast<const bf16x8*>(&rm);
        float g8[8], r8[8];
        load_mod8<M16>(gate, mrow + cc * 8, g8);
#pragma unroll
        for (int j = 0; j < 8; ++j) r8[j] = __builtin_fmaf((float)tm[j], g8[j], (float)tx[j]);
        // the residual leaves as a 16-bit tensor; the LayerNorm of the unfused definition sees THAT tensor
        const u32x4 pk = {pack_bf16(r8[0], r8[1]), pack_bf16(r8[2], r8[3]), pack_bf16(r8[4], r8[5]), pack_bf16(r8[6], r8[7])};
        *reinterpret_cast<u32x4*>(resi + (size_t)row * ldr + cc * 8) = pk;
        const bf16x8 rr = *reinterpret_cast<const bf16x8*>(&pk);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] = (float)rr[j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
      }
    }
    const float K = __shfl(v[0][0], 0, 64);
    float a = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (lane + 64 * i < cv) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = v[i][j] - K;
          a += d;
          q = __builtin_fmaf(d, d, q);
        }
      }
    a = wave_sum(a);
    q = wave_sum(q);
    const float m = a * invC;
    const float rstd = rsqrtf(fmaxf(q * invC - m * m, 0.f) + eps);
    const float mean = K + m;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int cc = lane + 64 * i;
      if (cc < cv) {
        float sc[8], sh[8], wv[8], bv[8], o[8];
        load_mod8<M16>(scale, mrow + cc * 8, sc);
        load_mod8<M16>(shift, mrow + cc * 8, sh);
        if (weight) load_mod8<M16>(weight, (size_t)cc * 8, wv);
        if (bias) load_mod8<M16>(bias, (size_t)cc * 8, bv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float y = (v[i][j] - mean) * rstd;
          if (weight) y *= wv[j];
          if (bias) y += bv[j];
          o[j] = __builtin_fmaf(y, 1.0f + sc[j], sh[j]);
        }
        const u32x4 pk = {pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7])};
        *reinterpret_cast<u32x4*>(out + (size_t)row * ldo + cc * 8) = pk;
      }
    }
  }
}

int launch_fused_adaln_scale_residual(const bf16* x, int ldx, const bf16* mha, int ldm, const void* gate, const void* scale,
                                      const void* shift, int ld_mod, int mod16, int rows_per_batch, const void* weight,
                                      const void* bias, float eps, int rows, int C, bf16* resi, int ldr, bf16* out, int ldo,
                                      hipStream_t stream) {
  if (rows <= 0 || C <= 0 || rows_per_batch <= 0) return SD_ERR_INVALID;
  if ((C & 7) || (ldx & 7) || (ldm & 7) || (ldr & 7) || (ldo & 7) || (ld_mod & (mod16 ? 7 : 3)) || C > 4096) return SD_ERR_UNSUPPORTED;
  if (mod16 && ((reinterpret_cast<uintptr_t>(weight) | reinterpret_cast<uintptr_t>(bias)) & 15)) return SD_ERR_UNSUPPORTED;
  const int cv = C >> 3;
  int blocks = (rows + 3) / 4;
  if (blocks > 2048) blocks = 2048;
#define SD_FA_LAUNCH(NCH, M16)                                                                                                     \
  hipLaunchKernelGGL((fused_adaln_scale_residual_kernel<NCH, M16>), dim3(blocks), dim3(256), 0, stream, x, ldx, mha, ldm, gate, scale, \
                     shift, ld_mod, rows_per_batch, weight, bias, eps, rows, C, resi, ldr, out, ldo)
  if (mod16) {
    if (cv <= 128) SD_FA_LAUNCH(2, true);
    else if (cv <= 256) SD_FA_LAUNCH(4, true);
    else SD_FA_LAUNCH(8, true);
  } else {
    if (cv <= 128) SD_FA_LAUNCH(2, false);
    else if (cv <= 256) SD_FA_LAUNCH(4, false);
    else SD_FA_LAUNCH(8, false);
  }
#undef SD_FA_LAUNCH
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

// out_i[b][s][:] = (s < S1 ? x[b][s] : y[b][s - S1])[i*C : (i+1)*C], i = 0 (q), 1 (k), 2 (v)
__global__ void split_concat_kernel(const bf16* __restrict__ x, const bf16* __restrict__ y, bf16* __restrict__ o0,
                                    bf16* __restrict__ o1, bf16* __restrict__ o2, int B, int S1, int S2, int cv) {
  const long total = (long)B * (S1 + S2) * 3 * cv;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
    const int cc = (int)(id % cv);
    const long t = id / cv;
    const int i = (int)(t % 3);
    const long rs = t / 3;                    // b * (S1 + S2) + s
    const int s = (int)(rs % (S1 + S2));
    const long b = rs / (S1 + S2);
    const bf16* src = (s < S1) ? x + ((b * S1 + s) * 3 + i) * (size_t)cv * 8 : y + ((b * S2 + (s - S1)) * 3 + i) * (size_t)cv * 8;
    bf16* dst = (i == 0 ? o0 : i == 1 ? o1 : o2) + (size_t)rs * cv * 8;
    *reinterpret_cast<u32x4*>(dst + cc * 8) = *reinterpret_cast<const u32x4*>(src + cc * 8);
  }
}

int launch_split_concat(const bf16* x, const bf16* y, bf16* o0, bf16* o1, bf16* o2, int B, int S1, int S2, int C, hipStream_t stream) {
  if (B <= 0 || S1 < 0 || S2 < 0 || S1 + S2 <= 0 || C <= 0) return SD_ERR_INVALID;
  if (C & 7) return SD_ERR_UNSUPPORTED;
  const long total = (long)B * (S1 + S2) * 3 * (C >> 3);
  long nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(split_concat_kernel, dim3((unsigned)nb), dim3(256), 0, stream, x, y, o0, o1, o2, B, S1, S2, C >> 3);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

}  // namespace sd
