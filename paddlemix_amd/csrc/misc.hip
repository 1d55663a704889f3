// Small kernels around the UNet body (gfx950): sinusoidal timestep embedding, SiLU, the 4-channel conv_in /
// conv_out 3x3 convolutions (too thin for the MFMA tile: direct kernels), strided row copy, and the
// a*x + b*y latent update used by the scheduler steps.
//
// Reference semantics:
//   get_timestep_embedding            ppdiffusers/ppdiffusers/models/embeddings.py:26-64  (fp32 math, then cast to model dtype
//                                     unet_2d_condition.py:946-951; SDXL time_ids path :1003-1008)
//   conv_in / conv_out                unet_2d_condition.py:1064, 1196 (3x3, pad 1), NCHW fp32 <-> NHWC bf16 at the boundary
//   scheduler latent updates          schedulers/scheduling_euler_discrete.py:438-473, scheduling_ddim.py:410-457
#include "common.h"
#include "kernels.h"

namespace sd {

// out[(i / group) * ldo + (i % group) * dim + j], i in [0, n), timestep t[i % t_count]
__global__ void timestep_embedding_kernel(const float* __restrict__ t, int t_count, int n, int dim, int group,
                                          int flip_sin_to_cos, float freq_shift, float scale, float max_period,
                                          bf16* __restrict__ out, int ldo) {
  const int half = dim / 2;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * half) return;
  const int i = idx / half, j = idx - i * half;
  const float exponent = -logf(max_period) * (float)j / ((float)half - freq_shift);
  const float freq = expf(exponent);
  const float arg = scale * (t[i % t_count] * freq);
  const float sn = sinf(arg), cs = cosf(arg);
  bf16* o = out + (size_t)(i / group) * ldo + (size_t)(i % group) * dim;
  if (flip_sin_to_cos) {
    o[j] = (bf16)cs;
    o[half + j] = (bf16)sn;
  } else {
    o[j] = (bf16)sn;
    o[half + j] = (bf16)cs;
  }
  if ((dim & 1) && j == 0) o[dim - 1] = (bf16)0.f;
}

int launch_timestep_embedding(const float* t, int t_count, int n, int dim, int group, int flip_sin_to_cos,
                              float freq_shift, float scale, float max_period, bf16* out, int ldo,
                              hipStream_t stream) {
  if (n <= 0 || dim < 2 || group <= 0 || t_count <= 0) return SD_ERR_INVALID;
  const int total = n * (dim / 2);
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, t, t_count, n, dim,
                     group, flip_sin_to_cos, freq_shift, scale, max_period, out, ldo);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

template <bool IN_F32, bool OUT_F32>
__global__ void silu_kernel(const void* __restrict__ x, void* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = IN_F32 ? reinterpret_cast<const float*>(x)[i] : (float)reinterpret_cast<const bf16*>(x)[i];
    const float r = silu_f(v);
    if (OUT_F32)
      reinterpret_cast<float*>(y)[i] = r;
    else
      reinterpret_cast<bf16*>(y)[i] = (bf16)r;
  }
}

int launch_silu(const void* x, void* y, long n, int in_f32, int out_f32, hipStream_t stream) {
  if (n <= 0) return SD_ERR_INVALID;
  long nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  dim3 g((unsigned)nb), b(256);
  if (in_f32 && out_f32)
    hipLaunchKernelGGL((silu_kernel<true, true>), g, b, 0, stream, x, y, n);
  else if (in_f32)
    hipLaunchKernelGGL((silu_kernel<true, false>), g, b, 0, stream, x, y, n);
  else if (out_f32)
    hipLaunchKernelGGL((silu_kernel<false, true>), g, b, 0, stream, x, y, n);
  else
    hipLaunchKernelGGL((silu_kernel<false, false>), g, b, 0, stream, x, y, n);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

// conv_in: x NCHW fp32 [B,Cin,H,W] (optionally multiplied by *in_scale: the scheduler's scale_model_input folded in),
// rounded to bf16 like the reference's sample.cast(self.dtype); w packed [9*Cin][Cout] bf16, k = (ky*3+kx)*Cin + ci.
// One thread = CI_PX horizontally adjacent pixels x 8 output channels: one weight chunk (load + 8 converts) feeds CI_PX x 8
// fmas, the image row comes from blockIdx.y (no 64-bit index arithmetic). Per output the taps are accumulated in the order
// (ky, kx, ci) with padding taps contributing exact zeros. (First version: one pixel per thread and four 64-bit divisions per
// 16 output bytes -- ~1300 VALU instructions per chunk, 192 us for the SDXL bs-8 launch whose 84 MB need ~20 us of HBM.)
constexpr int CI_PX = 4;
// CIN > 0: Cin is a compile-time constant (4 = the SD latents): per kernel row the thread's CIN x (CI_PX + 2) input patch and
// its 3 x CIN weight chunks are loaded in ONE batch and the (kx, ci) loop unrolls, so a thread waits out 3 L2 round trips
// instead of 36 (the fully rolled loop was latency-bound at 127 us; fully unrolled it needs ~400 registers). CIN == 0: runtime
// Cin, rolled loop.
template <bool OUT_F32, int CIN>
__global__ __launch_bounds__(256) void conv_in3x3_kernel(const float* __restrict__ x, const float* __restrict__ in_scale,
                                  const bf16* __restrict__ w, const float* __restrict__ bias, void* __restrict__ y,
                                  int B, int Cin, int H, int W, int Cout, int ldy) {
  const int cv = Cout >> 3;
  const int ngrp = (W + CI_PX - 1) / CI_PX;
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= ngrp * cv) return;
  const int grp = id / cv, cc = id - grp * cv;
  const int px0 = grp * CI_PX;
  const float xs = in_scale ? *in_scale : 1.0f;
  for (int row = blockIdx.y; row < B * H; row += gridDim.y) {   // row = b * H + py
  const int b = row / H, py = row - b * H;
  float acc[CI_PX][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float bj = bias ? bias[cc * 8 + j] : 0.f;
#pragma unroll
    for (int p = 0; p < CI_PX; ++p) acc[p][j] = bj;
  }
  if constexpr (CIN > 0) {
#pragma unroll 1
    for (int ky = 0; ky < 3; ++ky) {   // rolled on purpose: one batch of loads (CIN x 6 inputs + 3 x CIN weight chunks) per kernel row
      const int iy = py + ky - 1;
      if ((unsigned)iy >= (unsigned)H) continue;
      float xp[CIN][CI_PX + 2];
      u32x4 wr[3][CIN];
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) {
        const float* xr = x + (((size_t)b * CIN + ci) * H + iy) * W;
#pragma unroll
        for (int t = 0; t < CI_PX + 2; ++t) {
          const int ix = px0 + t - 1;
          float v = 0.f;
          if ((unsigned)ix < (unsigned)W) v = xr[ix];
          xp[ci][t] = v;
        }
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
          wr[kx][ci] = *reinterpret_cast<const u32x4*>(w + (size_t)((ky * 3 + kx) * CIN + ci) * Cout + cc * 8);
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int t = 0; t < CI_PX + 2; ++t) xp[ci][t] = (float)(bf16)(xp[ci][t] * xs);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
          const bf16x8 wv = *reinterpret_cast<const bf16x8*>(&wr[kx][ci]);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float wj = (float)wv[j];
#pragma unroll
            for (int p = 0; p < CI_PX; ++p) acc[p][j] = __builtin_fmaf(xp[ci][p + kx], wj, acc[p][j]);
          }
        }
      }
    }
  } else {
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = py + ky - 1;
    if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      for (int ci = 0; ci < Cin; ++ci) {
        const float* xr = x + (((size_t)b * Cin + ci) * H + iy) * W;
        float xv[CI_PX];
#pragma unroll
        for (int p = 0; p < CI_PX; ++p) {
          const int ix = px0 + p + kx - 1;
          xv[p] = ((unsigned)ix < (unsigned)W) ? (float)(bf16)(xr[ix] * xs) : 0.f;
        }
        const u32x4 raw = *reinterpret_cast<const u32x4*>(w + (size_t)((ky * 3 + kx) * Cin + ci) * Cout + cc * 8);
        const bf16x8 wv = *reinterpret_cast<const bf16x8*>(&raw);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float wj = (float)wv[j];
#pragma unroll
          for (int p = 0; p < CI_PX; ++p) acc[p][j] = __builtin_fmaf(xv[p], wj, acc[p][j]);
        }
      }
    }
  }
  }
#pragma unroll
  for (int p = 0; p < CI_PX; ++p) {
    if (px0 + p >= W) break;
    const size_t pix = (size_t)row * W + px0 + p;
    if constexpr (OUT_F32) {   // fp32 residual-stream mode: the stream starts unrounded
      float* yr = reinterpret_cast<float*>(y) + pix * ldy + cc * 8;
      *reinterpret_cast<f32x4*>(yr) = f32x4{acc[p][0], acc[p][1], acc[p][2], acc[p][3]};
      *reinterpret_cast<f32x4*>(yr + 4) = f32x4{acc[p][4], acc[p][5], acc[p][6], acc[p][7]};
    } else {
      u32x4 pk = {pack_bf16(acc[p][0], acc[p][1]), pack_bf16(acc[p][2], acc[p][3]), pack_bf16(acc[p][4], acc[p][5]),
                  pack_bf16(acc[p][6], acc[p][7])};
      *reinterpret_cast<u32x4*>(reinterpret_cast<bf16*>(y) + pix * ldy + cc * 8) = pk;
    }
  }
  }
}

int launch_conv_in3x3(const float* x_nchw, const float* in_scale, const bf16* w, const float* bias, void* y, int out_f32, int B,
                      int Cin, int H, int W, int Cout, int ldy, hipStream_t stream) {
  if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return SD_ERR_INVALID;
  if ((Cout & 7) || (ldy & 7) || (long)B * H >= (1L << 31)) return SD_ERR_UNSUPPORTED;
  const int per_row = ((W + CI_PX - 1) / CI_PX) * (Cout >> 3);
  const dim3 grid((per_row + 255) / 256, B * H < 65535 ? B * H : 65535);
#define SD_CI(F_, C_) \
  hipLaunchKernelGGL((conv_in3x3_kernel<F_, C_>), grid, dim3(256), 0, stream, x_nchw, in_scale, w, bias, y, B, Cin, H, W, Cout, ldy)
  if (Cin == 4) { if (out_f32) SD_CI(true, 4); else SD_CI(false, 4); }
  else          { if (out_f32) SD_CI(true, 0); else SD_CI(false, 0); }
#undef SD_CI
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

// conv_out: x NHWC bf16 (already GroupNorm+SiLU'd) -> y NCHW fp32 [B,Cout<=4,H,W]; w [Cout][9][Cin] bf16 in LDS.
// One wave per output pixel: lanes split the 9*Cin reduction in 16-B chunks, every chunk is 4 packed-pair dot products
// (v_dot2c) per output channel, then a wave reduction. (First version: 8 converts + 8 fmas per chunk and channel plus an
// integer division per trip -- 177 us for the 84 MB of the SDXL bs-8 launch; with dot products and no division: 137 us.)
constexpr int CO_MAX = 4;
// NI = chunks per lane = ceil(9 * Cin / 8 / 64), compile-time: the (tap, chunk) of lane's i-th chunk is the same for every
// pixel and lives in registers, and ALL of a pixel's input loads are issued before the first dot product (the rolled loop
// waited out one L2 round trip per chunk: 6 per pixel at Cin = 320 -- latency-bound at 137 us). Chunks past 9*Cin/8 and
// padding taps contribute zeros.
template <int NI>
__global__ void conv_out3x3_kernel(const bf16* __restrict__ x, int ldx, const bf16* __restrict__ w,
                                   const float* __restrict__ bias, float* __restrict__ y, int B, int Cin, int H, int W,
                                   int Cout) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  bf16* ws = reinterpret_cast<bf16*>(dsm);  // [Cout][9*Cin]
  const int K = 9 * Cin;
  for (int i = threadIdx.x; i < Cout * K / 8; i += blockDim.x)
    reinterpret_cast<u32x4*>(ws)[i] = reinterpret_cast<const u32x4*>(w)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const int cvin = Cin >> 3;
  const int nch = 9 * cvin;
  int dy[NI], dx[NI], woff[NI], coff[NI];
  bool valid[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int ch = lane + 64 * i;
    valid[i] = ch < nch;
    const int tap = valid[i] ? ch / cvin : 0;
    const int cc = valid[i] ? ch - tap * cvin : 0;
    const int ky = (tap * 11) >> 5;   // tap / 3 for tap < 9
    dy[i] = ky - 1;
    dx[i] = tap - ky * 3 - 1;
    woff[i] = tap * Cin + cc * 8;
    coff[i] = cc * 8;
  }
  const int HWp = H * W;
  const int npix = B * HWp;
  for (int pix = blockIdx.x * wpb + (threadIdx.x >> 6); pix < npix; pix += gridDim.x * wpb) {
    const int b = pix / HWp;
    const int rem = pix - b * HWp;
    const int py = rem / W, px = rem - py * W;
    u32x4 xv[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int iy = py + dy[i], ix = px + dx[i];
      xv[i] = u32x4{0u, 0u, 0u, 0u};
      if (valid[i] && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
        xv[i] = *reinterpret_cast<const u32x4*>(x + (((size_t)b * H + iy) * W + ix) * ldx + coff[i]);
    }
    float acc[CO_MAX] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NI; ++i) {
#pragma unroll
      for (int co = 0; co < CO_MAX; ++co) {
        if (co < Cout) {
          const u32x4 wv = *reinterpret_cast<const u32x4*>(ws + (size_t)co * K + woff[i]);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[co] = dot2_acc(xv[i][j], wv[j], acc[co]);
        }
      }
    }
#pragma unroll
    for (int co = 0; co < CO_MAX; ++co) acc[co] = wave_sum(acc[co]);
    if (lane == 0) {   // (statically indexed: a lane-indexed pick of acc[] goes through scratch memory)
#pragma unroll
      for (int co = 0; co < CO_MAX; ++co)
        if (co < Cout) y[(((size_t)b * Cout + co) * H + py) * W + px] = acc[co] + (bias ? bias[co] : 0.f);
    }
  }
}

int launch_conv_out3x3(const bf16* x, int ldx, const bf16* w, const float* bias, float* y_nchw, int B, int Cin, int H,
                       int W, int Cout, hipStream_t stream) {
  if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return SD_ERR_INVALID;
  if (Cout > CO_MAX || (Cin & 7) || (ldx & 7)) return SD_ERR_UNSUPPORTED;
  const size_t lds = (size_t)Cout * 9 * Cin * 2;
  if (lds > 64 * 1024) return SD_ERR_UNSUPPORTED;
  const long npix = (long)B * H * W;
  if (npix >= (1L << 31)) return SD_ERR_UNSUPPORTED;
  long nb = (npix + 3) / 4;
  if (nb > 2048) nb = 2048;
  const int ni = (9 * (Cin >> 3) + 63) / 64;
#define SD_CO(NI_) \
  hipLaunchKernelGGL(conv_out3x3_kernel<NI_>, dim3((unsigned)nb), dim3(256), lds, stream, x, ldx, w, bias, y_nchw, B, Cin, H, W, Cout)
  switch (ni) {
    case 1: SD_CO(1); break;
    case 2: SD_CO(2); break;
    case 3: SD_CO(3); break;
    case 4: SD_CO(4); break;
    case 5: case 6: SD_CO(6); break;
    case 7: case 8: case 9: SD_CO(9); break;
    case 10: case 11: case 12: SD_CO(12); break;
    default: return SD_ERR_UNSUPPORTED;   // Cin > 680 (lds caps Cin at 904 anyway)
  }
#undef SD_CO
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

// x[b*HW + p][c] += r[b][c][p]: a ControlNet residual (NCHW fp32, as the reference hands them over,
// unet_2d_condition.py:1121-1132, 1151-1155) added in place to a bf16 NHWC row view (row stride ldx, so it can be a
// channel slice of a concat buffer). 64 pixels x 64 channels per block through LDS: the fp32 reads run along pixels,
// the bf16 read-modify-write along channels.
template <bool XF32>
__global__ __launch_bounds__(256) void add_nchw_kernel(void* __restrict__ xv, int ldx, const float* __restrict__ r, int C,
                                                       long HW) {
  __shared__ float tile[64][65];
  const long p0 = (long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64, b = blockIdx.z;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int cc = ty; cc < 64; cc += 4) {
    const int c = c0 + cc;
    tile[cc][tx] = (c < C && p0 + tx < HW) ? r[((size_t)b * C + c) * HW + p0 + tx] : 0.f;
  }
  __syncthreads();
  const int cg = threadIdx.x & 7;          // 8 channels
  for (int pp = threadIdx.x >> 3; pp < 64; pp += 32) {
    const long p = p0 + pp;
    const int c = c0 + cg * 8;
    if (p >= HW || c >= C) continue;
    if constexpr (XF32) {   // fp32 residual-stream rows
      float* xr = reinterpret_cast<float*>(xv) + ((size_t)b * HW + p) * ldx + c;
      f32x4 a = *reinterpret_cast<const f32x4*>(xr), d = *reinterpret_cast<const f32x4*>(xr + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a[j] += tile[cg * 8 + j][pp];
        d[j] += tile[cg * 8 + 4 + j][pp];
      }
      *reinterpret_cast<f32x4*>(xr) = a;
      *reinterpret_cast<f32x4*>(xr + 4) = d;
    } else {
      bf16* xr = reinterpret_cast<bf16*>(xv) + ((size_t)b * HW + p) * ldx + c;
      const u32x4 raw = *reinterpret_cast<const u32x4*>(xr);
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(&raw);
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (float)v[j] + tile[cg * 8 + j][pp];
      u32x4 pk = {pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7])};
      *reinterpret_cast<u32x4*>(xr) = pk;
    }
  }
}

int launch_add_nchw(void* x, int x_f32, int ldx, const float* r, int B, int C, long HW, hipStream_t stream) {
  if (B <= 0 || C <= 0 || HW <= 0) return SD_ERR_INVALID;
  if ((C & 7) || (ldx & 7) || B > 65535 || (C + 63) / 64 > 65535) return SD_ERR_UNSUPPORTED;
  if (x_f32)
    hipLaunchKernelGGL(add_nchw_kernel<true>, dim3((unsigned)((HW + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)B), dim3(256), 0,
                       stream, x, ldx, r, C, HW);
  else
  hipLaunchKernelGGL(add_nchw_kernel<false>, dim3((unsigned)((HW + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)B), dim3(256), 0,
                     stream, x, ldx, r, C, HW);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

// CLIPTextEmbeddings.forward (PPD/transformers/clip/modeling.py:214-231): out[i] = token_embedding[ids[i]] +
// position_embedding[i % seq_len], summed in fp32 and stored as a bf16 row. ids are validated by the caller (host).
__global__ void embed_tokens_kernel(const int* __restrict__ ids, long n_tokens, int seq_len, const bf16* __restrict__ tok,
                                    const bf16* __restrict__ pos, int D, bf16* __restrict__ out, int ldo) {
  const int cv = D >> 3;
  const long total = n_tokens * cv;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
    const long t = id / cv;
    const int cc = (int)(id - t * cv);
    const u32x4 ra = *reinterpret_cast<const u32x4*>(tok + (size_t)ids[t] * D + cc * 8);
    u32x4 rb = {0u, 0u, 0u, 0u};   // pos == nullptr: token embedding only (T5)
    if (pos) rb = *reinterpret_cast<const u32x4*>(pos + (size_t)(t % seq_len) * D + cc * 8);
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(&ra), b = *reinterpret_cast<const bf16x8*>(&rb);
    u32x4 pk = {pack_bf16((float)a[0] + (float)b[0], (float)a[1] + (float)b[1]),
                pack_bf16((float)a[2] + (float)b[2], (float)a[3] + (float)b[3]),
                pack_bf16((float)a[4] + (float)b[4], (float)a[5] + (float)b[5]),
                pack_bf16((float)a[6] + (float)b[6], (float)a[7] + (float)b[7])};
    *reinterpret_cast<u32x4*>(out + (size_t)t * ldo + cc * 8) = pk;
  }
}

int launch_embed_tokens(const int* ids, long n_tokens, int seq_len, const bf16* tok, const bf16* pos, int D, bf16* out,
                        int ldo, hipStream_t stream) {
  if (n_tokens <= 0 || seq_len <= 0 || D <= 0) return SD_ERR_INVALID;
  if ((D & 7) || (ldo & 7)) return SD_ERR_UNSUPPORTED;
  long nb = (n_tokens * (D >> 3) + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(embed_tokens_kernel, dim3((unsigned)nb), dim3(256), 0, stream, ids, n_tokens, seq_len, tok, pos, D,
                     out, ldo);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

// Elementwise activations of the text encoders on bf16 rows (ACT2FN, PPD/transformers/activations.py):
// kind 0 = quick_gelu  x * sigmoid(1.702 x)  (CLIP ViT-L text), 1 = gelu (erf; OpenCLIP bigG), 2 = silu.
__global__ void activation_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, long n8, int kind) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const u32x4 raw = reinterpret_cast<const u32x4*>(x)[i];
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(&raw);
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = (float)v[j];
      o[j] = kind == 0 ? f / (1.0f + __expf(-1.702f * f)) : (kind == 1 ? gelu_erf_f(f) : silu_f(f));
    }
    u32x4 pk = {pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7])};
    reinterpret_cast<u32x4*>(y)[i] = pk;
  }
}

// T5DenseGatedActDense (PPD/transformers/t5/modeling.py:164-167): y[r][j] = act(x[r][j]) * x[r][F + j] on the fused
// [wi_0 | wi_1] projection; kind as in activation_kernel plus 3 = gelu_new (tanh approximation, T5 v1.1 "gated-gelu").
__global__ void gated_activation_kernel(const bf16* __restrict__ x, int ldx, bf16* __restrict__ y, int ldy, long rows, int F8,
                                        int F, int kind) {
  const long total = rows * F8;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
    const long r = id / F8;
    const int cc = (int)(id - r * F8);
    const u32x4 ra = *reinterpret_cast<const u32x4*>(x + (size_t)r * ldx + cc * 8);
    const u32x4 rb = *reinterpret_cast<const u32x4*>(x + (size_t)r * ldx + F + cc * 8);
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(&ra), b = *reinterpret_cast<const bf16x8*>(&rb);
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = (float)a[j];
      const float g = kind == 0 ? f / (1.0f + __expf(-1.702f * f)) : kind == 1 ? gelu_erf_f(f) : kind == 2 ? silu_f(f) : gelu_tanh_f(f);
      o[j] = g * (float)b[j];
    }
    u32x4 pk = {pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7])};
    *reinterpret_cast<u32x4*>(y + (size_t)r * ldy + cc * 8) = pk;
  }
}

int launch_gated_activation(const bf16* x, int ldx, bf16* y, int ldy, long rows, int F, int kind, hipStream_t stream) {
  if (rows <= 0 || F <= 0 || kind < 0 || kind > 3) return SD_ERR_INVALID;
  if ((F & 7) || (ldx & 7) || (ldy & 7)) return SD_ERR_UNSUPPORTED;
  long nb = (rows * (F >> 3) + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(gated_activation_kernel, dim3((unsigned)nb), dim3(256), 0, stream, x, ldx, y, ldy, rows, F >> 3, F, kind);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

int launch_activation(const bf16* x, bf16* y, long n, int kind, hipStream_t stream) {
  if (n <= 0 || kind < 0 || kind > 2) return SD_ERR_INVALID;
  if (n & 7) return SD_ERR_UNSUPPORTED;
  long nb = ((n >> 3) + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(activation_kernel, dim3((unsigned)nb), dim3(256), 0, stream, x, y, n >> 3, kind);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

// 1x1 convolution on a small NCHW fp32 tensor (AutoencoderKL.post_quant_conv, autoencoder_kl.py:121,292-293, folded
// with the 1 / scaling_factor of the pipelines): y[b,co,p] = bias[co] + sum_ci w[co][ci] * bf16(x[b,ci,p] * in_scale).
constexpr int C1_MAX = 16;
__global__ void conv1x1_nchw_kernel(const float* __restrict__ x, float in_scale, const bf16* __restrict__ w,
                                    const float* __restrict__ bias, float* __restrict__ y, int B, int Cin, int Cout,
                                    long HW) {
  const long total = (long)B * HW;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
    const long b = id / HW, px = id - b * HW;
    float xv[C1_MAX];
#pragma unroll
    for (int ci = 0; ci < C1_MAX; ++ci)
      if (ci < Cin) xv[ci] = (float)(bf16)(x[((size_t)b * Cin + ci) * HW + px] * in_scale);
    for (int co = 0; co < Cout; ++co) {
      float acc = bias ? bias[co] : 0.f;
#pragma unroll
      for (int ci = 0; ci < C1_MAX; ++ci)
        if (ci < Cin) acc = __builtin_fmaf(xv[ci], (float)w[co * Cin + ci], acc);
      y[((size_t)b * Cout + co) * HW + px] = acc;
    }
  }
}

int launch_conv1x1_nchw(const float* x, float in_scale, const bf16* w, const float* bias, float* y, int B, int Cin,
                        int Cout, long HW, hipStream_t stream) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || HW <= 0) return SD_ERR_INVALID;
  if (Cin > C1_MAX || Cout > C1_MAX) return SD_ERR_UNSUPPORTED;
  long nb = ((long)B * HW + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(conv1x1_nchw_kernel, dim3((unsigned)nb), dim3(256), 0, stream, x, in_scale, w, bias, y, B, Cin,
                     Cout, HW);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

// Row softmax of an fp32 score matrix into bf16 probabilities: y[r][j] = softmax_j(x[r][j]) (the scale is already in x).
// The single-head, head_dim = C attention of the VAE mid block (attention_processor.py:552-586 with upcast_softmax)
// is run as GEMM -> this -> GEMM because a 512-wide head does not fit the fused kernel's register budget.
// One 256-thread block per row; the row (<= 64 KB at 16384 keys) is re-read from L2 for the three passes.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, long ldx, bf16* __restrict__ y,
                                                           long ldy, int n) {
  __shared__ float red[4];
  const float* xr = x + (size_t)blockIdx.x * ldx;
  bf16* yr = y + (size_t)blockIdx.x * ldy;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n4 = n >> 2;
  float m = -INFINITY;
  for (int i = tid; i < n4; i += 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(xr)[i];
    m = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
  }
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int i = tid; i < n4; i += 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(xr)[i];
    s += __expf(v[0] - m) + __expf(v[1] - m) + __expf(v[2] - m) + __expf(v[3] - m);
  }
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  s = (red[0] + red[1]) + (red[2] + red[3]);   // fixed order: deterministic
  const float inv = 1.0f / s;
  for (int i = tid; i < n4; i += 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(xr)[i];
    u32x2 pk = {pack_bf16(__expf(v[0] - m) * inv, __expf(v[1] - m) * inv),
                pack_bf16(__expf(v[2] - m) * inv, __expf(v[3] - m) * inv)};
    reinterpret_cast<u32x2*>(yr)[i] = pk;
  }
}

int launch_softmax_rows(const float* x, long ldx, bf16* y, long ldy, long rows, int n, hipStream_t stream) {
  if (rows <= 0 || n <= 0) return SD_ERR_INVALID;
  if ((n & 3) || (ldx & 3) || (ldy & 3) || rows > 0x7fffffffL) return SD_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, stream, x, ldx, y, ldy, n);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

__global__ void copy_rows_kernel(const bf16* __restrict__ x, int ldx, bf16* __restrict__ y, int ldy, long rows, int cv) {
  const long total = rows * cv;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
    const long r = id / cv;
    const int cc = (int)(id - r * cv);
    *reinterpret_cast<u32x4*>(y + (size_t)r * ldy + cc * 8) = *reinterpret_cast<const u32x4*>(x + (size_t)r * ldx + cc * 8);
  }
}

// fp32 rows -> 16-bit rows (the operand copies of the fp32 residual-stream mode: inputs of the down / upsampling convs)
__global__ void cast_rows_kernel(const float* __restrict__ x, int ldx, bf16* __restrict__ y, int ldy, long rows, int cv) {
  const long total = rows * cv;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
    const long r = id / cv;
    const int cc = (int)(id - r * cv);
    const float* xr = x + (size_t)r * ldx + cc * 8;
    const f32x4 a = *reinterpret_cast<const f32x4*>(xr), b = *reinterpret_cast<const f32x4*>(xr + 4);
    u32x4 pk = {pack_bf16(a[0], a[1]), pack_bf16(a[2], a[3]), pack_bf16(b[0], b[1]), pack_bf16(b[2], b[3])};
    *reinterpret_cast<u32x4*>(y + (size_t)r * ldy + cc * 8) = pk;
  }
}

int launch_cast_rows(const float* x, int ldx, bf16* y, int ldy, long rows, int C, hipStream_t stream) {
  if (rows <= 0 || C <= 0) return SD_ERR_INVALID;
  if ((C & 7) || (ldx & 3) || (ldy & 7)) return SD_ERR_UNSUPPORTED;
  const long total = rows * (C >> 3);
  long nb = (total + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(cast_rows_kernel, dim3((unsigned)nb), dim3(256), 0, stream, x, ldx, y, ldy, rows, C >> 3);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

int launch_copy_rows(const bf16* x, int ldx, bf16* y, int ldy, long rows, int C, hipStream_t stream) {
  if (rows <= 0 || C <= 0) return SD_ERR_INVALID;
  if ((C & 7) || (ldx & 7) || (ldy & 7)) return SD_ERR_UNSUPPORTED;
  const long total = rows * (C >> 3);
  long nb = (total + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)nb), dim3(256), 0, stream, x, ldx, y, ldy, rows, C >> 3);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

// out = coef[0]*x + coef[1]*y  (fp32 latents; coefficients live in device memory so a captured graph can be
// replayed with per-step values)
__global__ void axpby_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out,
                             const float* __restrict__ coef, long n) {
  const float a = coef[0], b = coef[1];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = a * x[i] + b * y[i];
}

// Classifier-free-guidance combine + linear scheduler update in one pass over the latents
// (pipeline_stable_diffusion.py:882-891 followed by the epsilon-prediction step of Euler / DDIM(eta=0) / flow matching):
// out = coef[0] * x + coef[1] * (eu + gs * (et - eu)); eu / et = the unconditional / text halves of the UNet output.
__global__ void cfg_axpby_kernel(const float* __restrict__ x, const float* __restrict__ eu, const float* __restrict__ et,
                                 float* __restrict__ out, const float* __restrict__ coef, float gs, long n) {
  const float a = coef[0], b = coef[1];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float u = eu[i];
    out[i] = a * x[i] + b * (u + gs * (et[i] - u));
  }
}

int launch_cfg_axpby(const float* x, const float* eu, const float* et, float* out, const float* coef, float gs, long n,
                     hipStream_t stream) {
  if (n <= 0) return SD_ERR_INVALID;
  long nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(cfg_axpby_kernel, dim3((unsigned)nb), dim3(256), 0, stream, x, eu, et, out, coef, gs, n);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

// additive attention bias of a keep-mask: (1 - mask) * -10000 (unet_2d_condition.py:921-927, 1-D masks of 1 = attend, 0 = mask out)
__global__ void mask_to_bias_kernel(const float* __restrict__ mask, float* __restrict__ bias, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) bias[i] = (1.0f - mask[i]) * -10000.0f;
}
int launch_mask_to_bias(const float* mask, float* bias, long n, hipStream_t stream) {
  if (n <= 0) return SD_ERR_INVALID;
  long nb = (n + 255) / 256;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(mask_to_bias_kernel, dim3((unsigned)nb), dim3(256), 0, stream, mask, bias, n);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

int launch_axpby(const float* x, const float* y, float* out, const float* coef, long n, hipStream_t stream) {
  if (n <= 0) return SD_ERR_INVALID;
  long nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)nb), dim3(256), 0, stream, x, y, out, coef, n);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

// PatchEmbed.proj as a GEMM operand (embeddings.py:148-155, 209-219): x NCHW fp32 -> rows [B*(H/p)*(W/p), C*p*p] bf16 with
// the column order (c, py, px) of the flattened conv weight [D, C, p, p].
__global__ void patchify_kernel(const float* __restrict__ x, int B, int C, int H, int W, int p, bf16* __restrict__ out,
                                int ldo) {
  const int hp = H / p, wp = W / p, K = C * p * p;
  const long total = (long)B * hp * wp * K;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
    const int k = (int)(id % K);
    const long tok = id / K;
    const int px = k % p, py = (k / p) % p, c = k / (p * p);
    const int tx = (int)(tok % wp), ty = (int)((tok / wp) % hp), b = (int)(tok / ((long)wp * hp));
    out[(size_t)tok * ldo + k] = (bf16)x[(((size_t)b * C + c) * H + ty * p + py) * W + tx * p + px];
  }
}

int launch_patchify(const float* x_nchw, int B, int C, int H, int W, int p, bf16* out, int ldo, hipStream_t stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || p <= 0 || H % p || W % p) return SD_ERR_INVALID;
  const long total = (long)B * H * W * C;
  long nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)nb), dim3(256), 0, stream, x_nchw, B, C, H, W, p, out, ldo);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

// unpatchify (transformer_sd3.py:349-356): rows [B*h*w, p*p*C] (column order (py, px, c)) -> NCHW fp32 [B, C, h*p, w*p]
__global__ void unpatchify_kernel(const bf16* __restrict__ x, int ldx, int B, int C, int H, int W, int p,
                                  float* __restrict__ out) {
  const int hp = H / p, wp = W / p;
  const long total = (long)B * C * H * W;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
    const int xw = (int)(id % W), yh = (int)((id / W) % H), c = (int)((id / ((long)W * H)) % C);
    const int b = (int)(id / ((long)W * H * C));
    const int tx = xw / p, px = xw % p, ty = yh / p, py = yh % p;
    const size_t tok = ((size_t)b * hp + ty) * wp + tx;
    out[id] = (float)x[tok * ldx + (py * p + px) * C + c];
  }
}

int launch_unpatchify(const bf16* x, int ldx, int B, int C, int H, int W, int p, float* out_nchw, hipStream_t stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || p <= 0 || H % p || W % p) return SD_ERR_INVALID;
  const long total = (long)B * C * H * W;
  long nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(unpatchify_kernel, dim3((unsigned)nb), dim3(256), 0, stream, x, ldx, B, C, H, W, p, out_nchw);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

// DiagonalGaussianDistribution (PPD/models/vae.py:744-763) from the encoder's moment rows: m [B*HW][ld >= 2L] fp32 (the
// fp32 output of the conv_out GEMM, channels (mean_0..L-1, logvar_0..L-1)) -> NCHW fp32 mean, logvar clipped to [-30, 20],
// and (noise != NULL) sample = (mean + exp(0.5 logvar) * noise) * out_scale. One thread per output element: the row
// reads hit 2L*4 <= 128 B per pixel, the NCHW writes are unit-stride.
__global__ void latent_dist_kernel(const float* __restrict__ m, int ld, int B, int L, long HW,
                                   const float* __restrict__ noise, float out_scale, float* __restrict__ mean,
                                   float* __restrict__ logvar, float* __restrict__ sample) {
  const long total = (long)B * L * HW;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
    const long pix = id % HW;
    const int c = (int)((id / HW) % L);
    const int b = (int)(id / (HW * L));
    const float* row = m + ((size_t)b * HW + pix) * ld;
    const float mu = row[c];
    const float lv = fminf(fmaxf(row[L + c], -30.f), 20.f);
    mean[id] = mu;
    logvar[id] = lv;
    if (sample) sample[id] = (mu + (noise ? __expf(0.5f * lv) * noise[id] : 0.f)) * out_scale;
  }
}

int launch_latent_dist(const float* m, int ld, int B, int L, long HW, const float* noise, float out_scale, float* mean,
                       float* logvar, float* sample, hipStream_t stream) {
  if (B <= 0 || L <= 0 || HW <= 0 || ld < 2 * L) return SD_ERR_INVALID;
  const long total = (long)B * L * HW;
  long nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(latent_dist_kernel, dim3((unsigned)nb), dim3(256), 0, stream, m, ld, B, L, HW, noise, out_scale,
                     mean, logvar, sample);
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

}  // namespace sd
