// C ABI of libmi355x_sd.so (declared in include/mi355x_sd.h): argument validation + kernel launches.
#include "../../include/mi355x_sd.h"

#include <stdio.h>
#include <string.h>

#include "common.h"
#include "kernels.h"

namespace {
thread_local char g_err[512] = "";
int fail(int code, const char* fmt, const char* a = "") {
  snprintf(g_err, sizeof(g_err), fmt, a);
  return code;
}
int finish(int rc, const char* what) {
  if (rc == SD_OK) return SD_OK;
  if (rc == SD_ERR_HIP) {
    hipError_t e = hipGetLastError();
    snprintf(g_err, sizeof(g_err), "%s: HIP error: %s", what, hipGetErrorString(e));
  } else if (rc == SD_ERR_UNSUPPORTED) {
    snprintf(g_err, sizeof(g_err), "%s: unsupported shape/stride/alignment (see include/mi355x_sd.h)", what);
  } else {
    snprintf(g_err, sizeof(g_err), "%s: invalid argument", what);
  }
  return rc;
}
inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
}  // namespace

namespace sd {
void set_last_error(const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg ? msg : ""); }
}  // namespace sd

using namespace sd;

// ---- layout probe (tests/test_gpu_probe.py) -------------------------------------------------------------------
__global__ void probe_kernel(float* out) {
  __shared__ __attribute__((aligned(16))) bf16 lds[256];
  const int l = threadIdx.x;
  for (int i = l; i < 256; i += 64) lds[i] = (bf16)(float)i;
  __syncthreads();
  // 16x16x32: a lane supplies A[row = l&15][k = (l>>4)*8 + i], B[k = (l>>4)*8 + i][col = l&15]
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    const int k = (l >> 4) * 8 + i;
    a[i] = (bf16)(float)((((l & 15) * 7 + k * 3) % 11) - 5);
    b[i] = (bf16)(float)(((k * 5 + (l & 15) * 2) % 13) - 6);
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = mfma_16x16x32(a, b, c);
  for (int i = 0; i < 4; ++i) out[l * 24 + i] = c[i];
  // 32x32x16: A[row = l&31][k = (l>>5)*8 + i], B[k][col = l&31]
  for (int i = 0; i < 8; ++i) {
    const int k = (l >> 5) * 8 + i;
    a[i] = (bf16)(float)((((l & 31) * 7 + k * 3) % 11) - 5);
    b[i] = (bf16)(float)(((k * 5 + (l & 31) * 2) % 13) - 6);
  }
  f32x16 d;
  for (int i = 0; i < 16; ++i) d[i] = 0.f;
  d = mfma_32x32x16(a, b, d);
  for (int i = 0; i < 16; ++i) out[l * 24 + 4 + i] = d[i];
  const bf16x4 t = ds_read_tr16((lds_bf16x4*)(lds + l * 4));
  for (int i = 0; i < 4; ++i) out[l * 24 + 20 + i] = (float)t[i];
}

// Scratch argument of the GEMM-class entry points (ABI 12): NULL or 16-byte aligned DEVICE memory. A host pointer is refused here, not
// at the first split-K launch that writes through it (round 5: a host buffer bound on a misuse path killed a later launch with a page
// fault on a box without XNACK). The runtime query sits on every GEMM launch, so the last accepted (pointer, size) is remembered per
// thread and not asked about again; a pointer the runtime cannot classify (hipErrorInvalidValue: plain malloc memory on some ROCm
// releases, unregistered ranges) is NOT device memory as far as this library can tell and is refused as well.
static int check_ws(const void* ws, size_t bytes, const char* who) {
  if (!ws) return SD_OK;
  if (reinterpret_cast<uintptr_t>(ws) & 15) return fail(SD_ERR_INVALID, "%s: the workspace pointer must be 16-byte aligned", who);
  static thread_local const void* ok_ptr = nullptr;
  static thread_local size_t ok_bytes = 0;
  if (ws == ok_ptr && bytes <= ok_bytes) return SD_OK;
  hipPointerAttribute_t at;
  const hipError_t e = hipPointerGetAttributes(&at, ws);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(SD_ERR_INVALID, "%s: the workspace must be device memory (the HIP runtime does not know this pointer)", who);
  }
  if (at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeManaged)
    return fail(SD_ERR_INVALID, "%s: the workspace must be device memory", who);
  ok_ptr = ws;
  ok_bytes = bytes;
  return SD_OK;
}

extern "C" {

int mi355x_sd_abi_version(void) { return MI355X_SD_ABI_VERSION; }

int mi355x_sd_elem_dtype(void) {
#ifdef MI355X_SD_F16
  return MI355X_SD_ELEM_F16;
#else
  return MI355X_SD_ELEM_BF16;
#endif
}

const char* mi355x_sd_last_error(void) { return g_err; }

int mi355x_sd_init(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(SD_ERR_HIP, "mi355x_sd_init: no HIP device visible");
  if (device < 0 || device >= n) return fail(SD_ERR_INVALID, "mi355x_sd_init: device index out of range");
  if (hipSetDevice(device) != hipSuccess) return fail(SD_ERR_HIP, "mi355x_sd_init: hipSetDevice failed");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(SD_ERR_HIP, "mi355x_sd_init: no properties");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(SD_ERR_UNSUPPORTED, "mi355x_sd_init: device is %s, this library is built for gfx950 only",
                prop.gcnArchName);
  return SD_OK;
}

int mi355x_sd_linear(const void* A, int lda, const void* W, void* C, int ldc, int M, int N, int K, const float* bias,
                     const float* rowbias, int rows_per_batch, int ld_rowbias, const void* R, int ldr, float out_scale,
                     int flags, void* ws, size_t ws_bytes, void* stream) {
  if (!A || !W || !C) return fail(SD_ERR_INVALID, "mi355x_sd_linear: null pointer");
  if (int rc = check_ws(ws, ws_bytes, "mi355x_sd_linear")) return rc;
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.ws_base = ws; g.ws_bytes = ws ? ws_bytes : 0;
  g.A = (const bf16*)A; g.W = (const bf16*)W; g.C = C;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldc = ldc;
  g.bias = bias; g.rowbias = rowbias; g.rows_per_batch = rows_per_batch; g.ld_rowbias = ld_rowbias;
  g.R = (const bf16*)R; g.ldr = ldr; g.out_scale = out_scale;
  g.geglu = (flags & MI355X_SD_GEGLU) ? 1 : 0;
  g.out_f32 = (flags & MI355X_SD_OUT_F32) ? 1 : 0;
  g.r_f32 = (flags & MI355X_SD_R_F32) ? 1 : 0;
  g.silu = (flags & MI355X_SD_SILU) ? 1 : 0;
  return finish(launch_gemm(g, S(stream)), "mi355x_sd_linear");
}

int mi355x_sd_linear_ex(const void* A, int lda, int a_rows_per_batch, int64_t a_batch_stride, const void* W,
                        const float* w_scale, void* C,
                        int ldc, int c_rows_per_batch, int64_t c_batch_stride, int M, int N, int K, const float* bias,
                        const float* rowbias, int ld_rowbias, const float* gate, int ld_gate, int rows_per_batch,
                        const void* R, int ldr, float out_scale, int flags, void* ws, size_t ws_bytes, void* stream) {
  if (!A || !W || !C) return fail(SD_ERR_INVALID, "mi355x_sd_linear_ex: null pointer");
  if (int rc = check_ws(ws, ws_bytes, "mi355x_sd_linear_ex")) return rc;
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.ws_base = ws; g.ws_bytes = ws ? ws_bytes : 0;
  g.A = (const bf16*)A; g.W = (const bf16*)W; g.C = C;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldc = ldc;
  g.a_rpb = a_rows_per_batch; g.a_bstride = (long)a_batch_stride;
  g.c_rpb = c_rows_per_batch; g.c_bstride = (long)c_batch_stride;
  g.bias = bias; g.rowbias = rowbias; g.rows_per_batch = rows_per_batch; g.ld_rowbias = ld_rowbias;
  g.gate = gate; g.ld_gate = ld_gate;
  g.wscale = w_scale;
  g.R = (const bf16*)R; g.ldr = ldr; g.out_scale = out_scale;
  g.geglu = (flags & MI355X_SD_GEGLU) ? 1 : 0;
  g.out_f32 = (flags & MI355X_SD_OUT_F32) ? 1 : 0;
  g.r_f32 = (flags & MI355X_SD_R_F32) ? 1 : 0;
  g.silu = (flags & MI355X_SD_SILU) ? 1 : 0;
  g.gelu_tanh = (flags & MI355X_SD_GELU_TANH) ? 1 : 0;
  return finish(launch_gemm(g, S(stream)), "mi355x_sd_linear_ex");
}

int mi355x_sd_row_stats(const void* x, int rows, int C, int ldx, float eps, float* stats, void* stream) {
  if (!x || !stats) return fail(SD_ERR_INVALID, "mi355x_sd_row_stats: null pointer");
  return finish(launch_row_stats((const bf16*)x, rows, C, ldx, eps, stats, S(stream)), "mi355x_sd_row_stats");
}

int mi355x_sd_linear_ln(const void* A, int lda, const float* row_stats, const void* W, const float* w_rowsum, void* C,
                        int ldc, int M, int N, int K, const float* bias, int flags, void* ws, size_t ws_bytes, void* stream) {
  if (!A || !W || !C || !row_stats || !w_rowsum) return fail(SD_ERR_INVALID, "mi355x_sd_linear_ln: null pointer");
  if (int rc = check_ws(ws, ws_bytes, "mi355x_sd_linear_ln")) return rc;
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.ws_base = ws; g.ws_bytes = ws ? ws_bytes : 0;
  g.A = (const bf16*)A; g.W = (const bf16*)W; g.C = C;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldc = ldc;
  g.rowstat = row_stats; g.wsum = w_rowsum;
  g.bias = bias; g.out_scale = 1.0f;
  g.geglu = (flags & MI355X_SD_GEGLU) ? 1 : 0;
  g.out_f32 = (flags & MI355X_SD_OUT_F32) ? 1 : 0;
  g.r_f32 = (flags & MI355X_SD_R_F32) ? 1 : 0;
  g.silu = (flags & MI355X_SD_SILU) ? 1 : 0;
  g.gelu_tanh = (flags & MI355X_SD_GELU_TANH) ? 1 : 0;
  return finish(launch_gemm(g, S(stream)), "mi355x_sd_linear_ln");
}

int mi355x_sd_linear_f8(const void* A8, int lda, int a_rows_per_batch, int64_t a_batch_stride, const float* a_scale,
                        const void* W8, const float* w_scale, void* C, int ldc, int c_rows_per_batch,
                        int64_t c_batch_stride, int M, int N, int K, const float* bias, const float* gate, int ld_gate,
                        int rows_per_batch, const void* R, int ldr, int flags, void* stream) {
  if (!A8 || !W8 || !C || !a_scale || !w_scale) return fail(SD_ERR_INVALID, "mi355x_sd_linear_f8: null pointer");
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = (const bf16*)A8; g.W = (const bf16*)W8; g.C = C;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldc = ldc;
  g.a_rpb = a_rows_per_batch; g.a_bstride = (long)a_batch_stride;
  g.c_rpb = c_rows_per_batch; g.c_bstride = (long)c_batch_stride;
  g.ascale = a_scale; g.wscale = w_scale;
  g.bias = bias; g.gate = gate; g.ld_gate = ld_gate; g.rows_per_batch = rows_per_batch;
  g.R = (const bf16*)R; g.ldr = ldr; g.out_scale = 1.0f;
  g.gelu_tanh = (flags & MI355X_SD_GELU_TANH) ? 1 : 0;
  if (flags & ~MI355X_SD_GELU_TANH) return fail(SD_ERR_UNSUPPORTED, "mi355x_sd_linear_f8: only MI355X_SD_GELU_TANH is supported");
  return finish(launch_gemm_f8(g, S(stream)), "mi355x_sd_linear_f8");
}

int mi355x_sd_adaln_f8(const void* x, int rows, int C, int ldx, const float* scale, const float* shift, int ld_mod,
                       int rows_per_batch, float eps, void* y8, int ldy, float* y_scale, float* y_l2, void* stream) {
  if (!x || !scale || !shift || !y8 || !y_scale) return fail(SD_ERR_INVALID, "mi355x_sd_adaln_f8: null pointer");
  return finish(launch_adaln_f8((const bf16*)x, rows, C, ldx, scale, shift, ld_mod, rows_per_batch, eps, (unsigned char*)y8,
                                ldy, y_scale, y_l2, S(stream)), "mi355x_sd_adaln_f8");
}

int mi355x_sd_linear_f8_q(const void* A8, int lda, const float* a_scale, const float* a_l2, const void* W8,
                          const float* w_scale, float w_norm_max, void* C8, int ldc, float* c_scale, int M, int N, int K,
                          const float* bias, float bias_abs_max, int flags, void* stream) {
  if (!A8 || !W8 || !C8 || !a_scale || !a_l2 || !w_scale || !c_scale)
    return fail(SD_ERR_INVALID, "mi355x_sd_linear_f8_q: null pointer");
  if (flags & ~MI355X_SD_GELU_TANH) return fail(SD_ERR_UNSUPPORTED, "mi355x_sd_linear_f8_q: only MI355X_SD_GELU_TANH is supported");
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = (const bf16*)A8; g.W = (const bf16*)W8; g.C = C8;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldc = ldc;
  g.ascale = a_scale; g.wscale = w_scale; g.bias = bias; g.out_scale = 1.0f;
  g.out_f8 = 1; g.a_l2 = a_l2; g.w_norm_max = w_norm_max; g.bias_abs_max = bias_abs_max; g.oscale = c_scale;
  g.gelu_tanh = (flags & MI355X_SD_GELU_TANH) ? 1 : 0;
  return finish(launch_gemm_f8(g, S(stream)), "mi355x_sd_linear_f8_q");
}

int mi355x_sd_quantize_rows(const void* x, int64_t rows, int C, int ldx, int x_rows_per_batch, int64_t x_batch_stride,
                            void* y8, int ldy, float* y_scale, void* stream) {
  if (!x || !y8 || !y_scale) return fail(SD_ERR_INVALID, "mi355x_sd_quantize_rows: null pointer");
  return finish(launch_quantize_rows((const bf16*)x, (long)rows, C, ldx, x_rows_per_batch, (long)x_batch_stride,
                                     (unsigned char*)y8, ldy, y_scale, S(stream)), "mi355x_sd_quantize_rows");
}

int mi355x_sd_adaln(const void* x, int rows, int C, int ldx, const float* scale, const float* shift, int ld_mod,
                    int rows_per_batch, float eps, void* y, int ldy, void* stream) {
  if (!x || !scale || !shift || !y) return fail(SD_ERR_INVALID, "mi355x_sd_adaln: null pointer");
  return finish(launch_adaln((const bf16*)x, rows, C, ldx, scale, shift, ld_mod, 0, rows_per_batch, eps, (bf16*)y, ldy,
                             S(stream)),
                "mi355x_sd_adaln");
}

int mi355x_sd_adaln_ex(const void* x, int rows, int C, int ldx, const void* scale, const void* shift, int ld_mod, int mod_dtype,
                       int rows_per_batch, float eps, void* y, int ldy, void* stream) {
  if (!x || !scale || !shift || !y) return fail(SD_ERR_INVALID, "mi355x_sd_adaln_ex: null pointer");
  if (mod_dtype != MI355X_SD_MOD_F32 && mod_dtype != MI355X_SD_MOD_ELEM)
    return fail(SD_ERR_INVALID, "mi355x_sd_adaln_ex: mod_dtype must be MI355X_SD_MOD_F32 or MI355X_SD_MOD_ELEM");
  return finish(launch_adaln((const bf16*)x, rows, C, ldx, scale, shift, ld_mod, mod_dtype == MI355X_SD_MOD_ELEM, rows_per_batch, eps,
                             (bf16*)y, ldy, S(stream)),
                "mi355x_sd_adaln_ex");
}

int mi355x_sd_patchify(const float* x_nchw, int B, int C, int H, int W, int patch, void* out, int ldo, void* stream) {
  if (!x_nchw || !out) return fail(SD_ERR_INVALID, "mi355x_sd_patchify: null pointer");
  return finish(launch_patchify(x_nchw, B, C, H, W, patch, (bf16*)out, ldo, S(stream)), "mi355x_sd_patchify");
}

int mi355x_sd_unpatchify(const void* x, int ldx, int B, int C, int H, int W, int patch, float* out_nchw, void* stream) {
  if (!x || !out_nchw) return fail(SD_ERR_INVALID, "mi355x_sd_unpatchify: null pointer");
  return finish(launch_unpatchify((const bf16*)x, ldx, B, C, H, W, patch, out_nchw, S(stream)), "mi355x_sd_unpatchify");
}

int mi355x_sd_conv3x3(const void* X, int ldx, int B, int Hs, int Ws, int Cin, int stride, int upsample, const void* W,
                      void* C, int ldc, int Cout, const float* bias, const float* rowbias, int ld_rowbias,
                      const void* R, int ldr, float out_scale, int flags, void* ws, size_t ws_bytes, void* stream) {
  if (!X || !W || !C) return fail(SD_ERR_INVALID, "mi355x_sd_conv3x3: null pointer");
  if (int rc = check_ws(ws, ws_bytes, "mi355x_sd_conv3x3")) return rc;
  if (B <= 0 || Hs <= 0 || Ws <= 0 || Cin <= 0 || Cout <= 0 || (stride != 1 && stride != 2) ||
      (upsample != 0 && upsample != 1))
    return fail(SD_ERR_INVALID, "mi355x_sd_conv3x3: bad shape");
  if (flags & MI355X_SD_GEGLU) return fail(SD_ERR_UNSUPPORTED, "mi355x_sd_conv3x3: GEGLU epilogue is linear-only");
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.ws_base = ws; g.ws_bytes = ws ? ws_bytes : 0;
  g.A = (const bf16*)X; g.W = (const bf16*)W; g.C = C;
  g.conv = 1; g.Hs = Hs; g.Ws = Ws; g.Cin = Cin; g.stride = stride; g.up = upsample;
  g.pad = (flags & MI355X_SD_PAD_BR) ? 0 : 1;
  g.kb64 = (flags & MI355X_SD_CONV_KB64) ? 1 : 0;
  if (g.kb64 && (Cin & 63)) return fail(SD_ERR_UNSUPPORTED, "mi355x_sd_conv3x3: MI355X_SD_CONV_KB64 needs Cin % 64 == 0");
  if (!g.pad && (stride != 2 || upsample))
    return fail(SD_ERR_UNSUPPORTED, "mi355x_sd_conv3x3: MI355X_SD_PAD_BR is the stride-2 downsampler's padding");
  const int Hin = Hs << upsample, Win = Ws << upsample;
  g.Ho = (Hin + 1 + g.pad - 3) / stride + 1;
  g.Wo = (Win + 1 + g.pad - 3) / stride + 1;
  g.M = B * g.Ho * g.Wo; g.N = Cout; g.K = 9 * Cin; g.lda = ldx; g.ldc = ldc;
  g.bias = bias; g.rowbias = rowbias; g.rows_per_batch = g.Ho * g.Wo; g.ld_rowbias = ld_rowbias;
  g.R = (const bf16*)R; g.ldr = ldr; g.out_scale = out_scale;
  g.out_f32 = (flags & MI355X_SD_OUT_F32) ? 1 : 0;
  g.r_f32 = (flags & MI355X_SD_R_F32) ? 1 : 0;
  g.silu = (flags & MI355X_SD_SILU) ? 1 : 0;
  return finish(launch_gemm(g, S(stream)), "mi355x_sd_conv3x3");
}

int mi355x_sd_sdpa(const void* q, const void* k, const void* v, const float* bias, void* out, int B, int H, int Sq,
                   int Skv, int D, int64_t q_bs, int q_ts, int64_t k_bs, int k_ts, int64_t v_bs, int v_ts,
                   int64_t o_bs, int o_ts, int64_t bias_bs, int64_t bias_hs, int64_t bias_qs, float scale,
                   void* stream) {
  if (!q || !k || !v || !out) return fail(SD_ERR_INVALID, "mi355x_sd_sdpa: null pointer");
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.Q = (const bf16*)q; a.K = (const bf16*)k; a.V = (const bf16*)v; a.O = (bf16*)out;
  a.B = B; a.H = H; a.Sq = Sq; a.Skv = Skv; a.D = D;
  a.q_bs = q_bs; a.k_bs = k_bs; a.v_bs = v_bs; a.o_bs = o_bs;
  a.q_ts = q_ts; a.k_ts = k_ts; a.v_ts = v_ts; a.o_ts = o_ts;
  a.bias = bias; a.bias_bs = bias_bs; a.bias_hs = bias_hs; a.bias_qs = bias_qs;
  a.scale = scale;
  return finish(launch_attention(a, S(stream)), "mi355x_sd_sdpa");
}

int mi355x_sd_sdpa_ex(const void* q, const void* k, const void* v, const float* bias, void* out, int B, int H, int Sq,
                      int Skv, int D, int64_t q_bs, int q_ts, int64_t k_bs, int k_ts, int64_t v_bs, int v_ts,
                      int64_t o_bs, int o_ts, int64_t bias_bs, int64_t bias_hs, int64_t bias_qs, float scale, int flags,
                      void* stream) {
  if (!q || !k || !v || !out) return fail(SD_ERR_INVALID, "mi355x_sd_sdpa_ex: null pointer");
  if ((flags & MI355X_SD_SDPA_LOG2) && (bias || D != 64))
    return fail(SD_ERR_UNSUPPORTED, "mi355x_sd_sdpa_ex: MI355X_SD_SDPA_LOG2 needs D == 64 and no mask");
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.Q = (const bf16*)q; a.K = (const bf16*)k; a.V = (const bf16*)v; a.O = (bf16*)out;
  a.B = B; a.H = H; a.Sq = Sq; a.Skv = Skv; a.D = D;
  a.q_bs = q_bs; a.k_bs = k_bs; a.v_bs = v_bs; a.o_bs = o_bs;
  a.q_ts = q_ts; a.k_ts = k_ts; a.v_ts = v_ts; a.o_ts = o_ts;
  a.bias = bias; a.bias_bs = bias_bs; a.bias_hs = bias_hs; a.bias_qs = bias_qs;
  a.log2 = (flags & MI355X_SD_SDPA_LOG2) ? 1 : 0;
  a.scale = a.log2 ? 1.0f : scale;
  return finish(launch_attention(a, S(stream)), "mi355x_sd_sdpa_ex");
}

int mi355x_sd_sdpa_accum(const void* q, const void* k, const void* v, const float* bias, void* out, int B, int H, int Sq,
                         int Skv, int D, int64_t q_bs, int q_ts, int64_t k_bs, int k_ts, int64_t v_bs, int v_ts,
                         int64_t o_bs, int o_ts, int64_t bias_bs, int64_t bias_hs, int64_t bias_qs, float scale,
                         float out_scale, void* stream) {
  if (!q || !k || !v || !out) return fail(SD_ERR_INVALID, "mi355x_sd_sdpa_accum: null pointer");
  if (out_scale == 0.f) return SD_OK;   // out += 0 * attention
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.Q = (const bf16*)q; a.K = (const bf16*)k; a.V = (const bf16*)v; a.O = (bf16*)out;
  a.B = B; a.H = H; a.Sq = Sq; a.Skv = Skv; a.D = D;
  a.q_bs = q_bs; a.k_bs = k_bs; a.v_bs = v_bs; a.o_bs = o_bs;
  a.q_ts = q_ts; a.k_ts = k_ts; a.v_ts = v_ts; a.o_ts = o_ts;
  a.bias = bias; a.bias_bs = bias_bs; a.bias_hs = bias_hs; a.bias_qs = bias_qs;
  a.scale = scale;
  a.accum = out_scale;
  return finish(launch_attention(a, S(stream)), "mi355x_sd_sdpa_accum");
}

int mi355x_sd_groupnorm_workspace_floats(int B, int HW, int C) { return groupnorm_partial_floats(B, HW, C); }

int mi355x_sd_groupnorm_stats(const void* x, int B, int HW, int C, int ldx, int groups, float eps, const float* gamma,
                              const float* beta, float* workspace, float* scale_shift, void* stream) {
  if (!x || !gamma || !beta || !workspace || !scale_shift)
    return fail(SD_ERR_INVALID, "mi355x_sd_groupnorm_stats: null pointer");
  return finish(launch_groupnorm_stats(x, 0, B, HW, C, ldx, groups, eps, gamma, beta, workspace, scale_shift, S(stream)),
                "mi355x_sd_groupnorm_stats");
}

int mi355x_sd_groupnorm_stats_ex(const void* x, int B, int HW, int C, int ldx, int groups, float eps, const float* gamma,
                                 const float* beta, float* workspace, float* scale_shift, int x_f32, void* stream) {
  if (!x || !gamma || !beta || !workspace || !scale_shift)
    return fail(SD_ERR_INVALID, "mi355x_sd_groupnorm_stats_ex: null pointer");
  return finish(launch_groupnorm_stats(x, x_f32, B, HW, C, ldx, groups, eps, gamma, beta, workspace, scale_shift, S(stream)),
                "mi355x_sd_groupnorm_stats_ex");
}

int mi355x_sd_groupnorm_act_fits(int HW, int C, int groups) { return groupnorm_act_fits(HW, C, groups); }

int mi355x_sd_groupnorm_act(const void* x, int B, int HW, int C, int ldx, int groups, float eps, const float* gamma, const float* beta,
                            int silu, void* y, int ldy, void* stream) {
  if (!x || !gamma || !beta || !y) return fail(SD_ERR_INVALID, "mi355x_sd_groupnorm_act: null pointer");
  return finish(launch_groupnorm_act((const bf16*)x, B, HW, C, ldx, groups, eps, gamma, beta, silu, (bf16*)y, ldy, S(stream)),
                "mi355x_sd_groupnorm_act");
}

int mi355x_sd_scale_shift_act(const void* x, int B, int HW, int C, int ldx, const float* scale_shift, int silu, void* y,
                              int ldy, void* stream) {
  if (!x || !scale_shift || !y) return fail(SD_ERR_INVALID, "mi355x_sd_scale_shift_act: null pointer");
  return finish(launch_scale_shift_act(x, 0, B, HW, C, ldx, scale_shift, silu, (bf16*)y, ldy, nullptr, 0, S(stream)),
                "mi355x_sd_scale_shift_act");
}

int mi355x_sd_scale_shift_act_ex(const void* x, int B, int HW, int C, int ldx, const float* scale_shift, int silu, void* y,
                                 int ldy, int x_f32, void* raw16, int ld_raw, void* stream) {
  if (!x || !scale_shift || !y) return fail(SD_ERR_INVALID, "mi355x_sd_scale_shift_act_ex: null pointer");
  return finish(launch_scale_shift_act(x, x_f32, B, HW, C, ldx, scale_shift, silu, (bf16*)y, ldy, (bf16*)raw16, ld_raw,
                                       S(stream)),
                "mi355x_sd_scale_shift_act_ex");
}

int mi355x_sd_layernorm(const void* x, int rows, int C, int ldx, const float* gamma, const float* beta, float eps,
                        void* y, int ldy, void* stream) {
  if (!x || !y) return fail(SD_ERR_INVALID, "mi355x_sd_layernorm: null pointer");
  return finish(launch_layernorm(x, 0, rows, C, ldx, gamma, beta, eps, (bf16*)y, ldy, S(stream)),
                "mi355x_sd_layernorm");
}

int mi355x_sd_layernorm_ex(const void* x, int rows, int C, int ldx, const float* gamma, const float* beta, float eps,
                           void* y, int ldy, int x_f32, void* stream) {
  if (!x || !y) return fail(SD_ERR_INVALID, "mi355x_sd_layernorm_ex: null pointer");
  return finish(launch_layernorm(x, x_f32, rows, C, ldx, gamma, beta, eps, (bf16*)y, ldy, S(stream)),
                "mi355x_sd_layernorm_ex");
}

int mi355x_sd_fused_adaln_scale_residual(const void* x, int ldx, const void* mha_out, int ld_mha, const float* gate_msa,
                                         const float* scale_mlp, const float* shift_mlp, int ld_mod, int rows_per_batch,
                                         const float* weight, const float* bias, float epsilon, int rows, int C, void* resi_out,
                                         int ld_resi, void* adaln_out, int ld_out, void* stream) {
  if (!x || !mha_out || !gate_msa || !scale_mlp || !shift_mlp || !resi_out || !adaln_out)
    return fail(SD_ERR_INVALID, "mi355x_sd_fused_adaln_scale_residual: null pointer");
  return finish(launch_fused_adaln_scale_residual((const bf16*)x, ldx, (const bf16*)mha_out, ld_mha, gate_msa, scale_mlp, shift_mlp,
                                                  ld_mod, 0, rows_per_batch, weight, bias, epsilon, rows, C, (bf16*)resi_out, ld_resi,
                                                  (bf16*)adaln_out, ld_out, S(stream)),
                "mi355x_sd_fused_adaln_scale_residual");
}

int mi355x_sd_fused_adaln_scale_residual_ex(const void* x, int ldx, const void* mha_out, int ld_mha, const void* gate_msa,
                                            const void* scale_mlp, const void* shift_mlp, int ld_mod, int mod_dtype,
                                            int rows_per_batch, const void* weight, const void* bias, float epsilon, int rows, int C,
                                            void* resi_out, int ld_resi, void* adaln_out, int ld_out, void* stream) {
  if (!x || !mha_out || !gate_msa || !scale_mlp || !shift_mlp || !resi_out || !adaln_out)
    return fail(SD_ERR_INVALID, "mi355x_sd_fused_adaln_scale_residual_ex: null pointer");
  if (mod_dtype != MI355X_SD_MOD_F32 && mod_dtype != MI355X_SD_MOD_ELEM)
    return fail(SD_ERR_INVALID, "mi355x_sd_fused_adaln_scale_residual_ex: mod_dtype must be MI355X_SD_MOD_F32 or MI355X_SD_MOD_ELEM");
  return finish(launch_fused_adaln_scale_residual((const bf16*)x, ldx, (const bf16*)mha_out, ld_mha, gate_msa, scale_mlp, shift_mlp,
                                                  ld_mod, mod_dtype == MI355X_SD_MOD_ELEM, rows_per_batch, weight, bias, epsilon, rows,
                                                  C, (bf16*)resi_out, ld_resi, (bf16*)adaln_out, ld_out, S(stream)),
                "mi355x_sd_fused_adaln_scale_residual_ex");
}

int mi355x_sd_split_concat(const void* x, const void* y, void* q_out, void* k_out, void* v_out, int B, int S1, int S2, int C,
                           void* stream) {
  if (!x || !y || !q_out || !k_out || !v_out) return fail(SD_ERR_INVALID, "mi355x_sd_split_concat: null pointer");
  return finish(launch_split_concat((const bf16*)x, (const bf16*)y, (bf16*)q_out, (bf16*)k_out, (bf16*)v_out, B, S1, S2, C, S(stream)),
                "mi355x_sd_split_concat");
}

int mi355x_sd_timestep_embedding(const float* t, int t_count, int n, int dim, int group, int flip_sin_to_cos,
                                 float freq_shift, float scale, float max_period, void* out, int ldo, void* stream) {
  if (!t || !out) return fail(SD_ERR_INVALID, "mi355x_sd_timestep_embedding: null pointer");
  return finish(launch_timestep_embedding(t, t_count, n, dim, group, flip_sin_to_cos, freq_shift, scale, max_period,
                                          (bf16*)out, ldo, S(stream)),
                "mi355x_sd_timestep_embedding");
}

int mi355x_sd_silu(const void* x, void* y, int64_t n, int in_f32, int out_f32, void* stream) {
  if (!x || !y) return fail(SD_ERR_INVALID, "mi355x_sd_silu: null pointer");
  return finish(launch_silu(x, y, (long)n, in_f32, out_f32, S(stream)), "mi355x_sd_silu");
}

int mi355x_sd_conv_in3x3(const float* x_nchw, const float* in_scale, const void* w, const float* bias, void* y, int B,
                         int Cin, int H, int W, int Cout, int ldy, void* stream) {
  if (!x_nchw || !w || !y) return fail(SD_ERR_INVALID, "mi355x_sd_conv_in3x3: null pointer");
  return finish(launch_conv_in3x3(x_nchw, in_scale, (const bf16*)w, bias, y, 0, B, Cin, H, W, Cout, ldy, S(stream)),
                "mi355x_sd_conv_in3x3");
}

int mi355x_sd_conv_in3x3_ex(const float* x_nchw, const float* in_scale, const void* w, const float* bias, void* y, int B,
                            int Cin, int H, int W, int Cout, int ldy, int out_f32, void* stream) {
  if (!x_nchw || !w || !y) return fail(SD_ERR_INVALID, "mi355x_sd_conv_in3x3_ex: null pointer");
  return finish(launch_conv_in3x3(x_nchw, in_scale, (const bf16*)w, bias, y, out_f32, B, Cin, H, W, Cout, ldy, S(stream)),
                "mi355x_sd_conv_in3x3_ex");
}

int mi355x_sd_cast_rows(const float* x, int ldx, void* y, int ldy, int64_t rows, int C, void* stream) {
  if (!x || !y) return fail(SD_ERR_INVALID, "mi355x_sd_cast_rows: null pointer");
  return finish(launch_cast_rows(x, ldx, (bf16*)y, ldy, (long)rows, C, S(stream)), "mi355x_sd_cast_rows");
}

int mi355x_sd_conv_out3x3(const void* x, int ldx, const void* w, const float* bias, float* y_nchw, int B, int Cin, int H,
                          int W, int Cout, void* stream) {
  if (!x || !w || !y_nchw) return fail(SD_ERR_INVALID, "mi355x_sd_conv_out3x3: null pointer");
  return finish(launch_conv_out3x3((const bf16*)x, ldx, (const bf16*)w, bias, y_nchw, B, Cin, H, W, Cout, S(stream)),
                "mi355x_sd_conv_out3x3");
}

int mi355x_sd_copy_rows(const void* x, int ldx, void* y, int ldy, int64_t rows, int C, void* stream) {
  if (!x || !y) return fail(SD_ERR_INVALID, "mi355x_sd_copy_rows: null pointer");
  return finish(launch_copy_rows((const bf16*)x, ldx, (bf16*)y, ldy, (long)rows, C, S(stream)), "mi355x_sd_copy_rows");
}

int mi355x_sd_add_nchw(void* x, int ldx, const float* r_nchw, int B, int C, int64_t HW, void* stream) {
  if (!x || !r_nchw) return fail(SD_ERR_INVALID, "mi355x_sd_add_nchw: null pointer");
  return finish(launch_add_nchw(x, 0, ldx, r_nchw, B, C, (long)HW, S(stream)), "mi355x_sd_add_nchw");
}

int mi355x_sd_add_nchw_ex(void* x, int ldx, const float* r_nchw, int B, int C, int64_t HW, int x_f32, void* stream) {
  if (!x || !r_nchw) return fail(SD_ERR_INVALID, "mi355x_sd_add_nchw_ex: null pointer");
  return finish(launch_add_nchw(x, x_f32, ldx, r_nchw, B, C, (long)HW, S(stream)), "mi355x_sd_add_nchw_ex");
}

int mi355x_sd_latent_dist(const float* moments, int ld, int B, int L, int64_t HW, const float* noise_nchw, float out_scale,
                          float* mean_nchw, float* logvar_nchw, float* sample_nchw, void* stream) {
  if (!moments || !mean_nchw || !logvar_nchw) return fail(SD_ERR_INVALID, "mi355x_sd_latent_dist: null pointer");
  return finish(launch_latent_dist(moments, ld, B, L, (long)HW, noise_nchw, out_scale, mean_nchw, logvar_nchw, sample_nchw,
                                   S(stream)), "mi355x_sd_latent_dist");
}

int mi355x_sd_embed_tokens(const int32_t* ids, int64_t n_tokens, int seq_len, const void* token_table,
                           const void* position_table, int D, void* out, int ldo, void* stream) {
  if (!ids || !token_table || !out) return fail(SD_ERR_INVALID, "mi355x_sd_embed_tokens: null pointer");
  return finish(launch_embed_tokens(ids, (long)n_tokens, seq_len, (const bf16*)token_table, (const bf16*)position_table, D,
                                    (bf16*)out, ldo, S(stream)), "mi355x_sd_embed_tokens");
}

int mi355x_sd_rmsnorm(const void* x, int rows, int C, int ldx, const float* weight, float eps, void* y, int ldy,
                      void* stream) {
  if (!x || !weight || !y) return fail(SD_ERR_INVALID, "mi355x_sd_rmsnorm: null pointer");
  return finish(launch_rmsnorm((const bf16*)x, rows, C, ldx, weight, eps, (bf16*)y, ldy, S(stream)), "mi355x_sd_rmsnorm");
}

int mi355x_sd_gated_activation(const void* x, int ldx, void* y, int ldy, int64_t rows, int F, int kind, void* stream) {
  if (!x || !y) return fail(SD_ERR_INVALID, "mi355x_sd_gated_activation: null pointer");
  return finish(launch_gated_activation((const bf16*)x, ldx, (bf16*)y, ldy, (long)rows, F, kind, S(stream)),
                "mi355x_sd_gated_activation");
}

int mi355x_sd_activation(const void* x, void* y, int64_t n, int kind, void* stream) {
  if (!x || !y) return fail(SD_ERR_INVALID, "mi355x_sd_activation: null pointer");
  return finish(launch_activation((const bf16*)x, (bf16*)y, (long)n, kind, S(stream)), "mi355x_sd_activation");
}

int mi355x_sd_conv1x1_nchw(const float* x_nchw, float in_scale, const void* w, const float* bias, float* y_nchw, int B,
                           int Cin, int Cout, int64_t HW, void* stream) {
  if (!x_nchw || !w || !y_nchw) return fail(SD_ERR_INVALID, "mi355x_sd_conv1x1_nchw: null pointer");
  return finish(launch_conv1x1_nchw(x_nchw, in_scale, (const bf16*)w, bias, y_nchw, B, Cin, Cout, (long)HW, S(stream)),
                "mi355x_sd_conv1x1_nchw");
}

int mi355x_sd_softmax_rows(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int n, void* stream) {
  if (!x || !y) return fail(SD_ERR_INVALID, "mi355x_sd_softmax_rows: null pointer");
  return finish(launch_softmax_rows(x, (long)ldx, (bf16*)y, (long)ldy, (long)rows, n, S(stream)),
                "mi355x_sd_softmax_rows");
}

int mi355x_sd_axpby(const float* x, const float* y, float* out, const float* coef, int64_t n, void* stream) {
  if (!x || !y || !out || !coef) return fail(SD_ERR_INVALID, "mi355x_sd_axpby: null pointer");
  return finish(launch_axpby(x, y, out, coef, (long)n, S(stream)), "mi355x_sd_axpby");
}

int mi355x_sd_mask_to_bias(const float* mask, float* bias, int64_t n, void* stream) {
  if (!mask || !bias) return fail(SD_ERR_INVALID, "mi355x_sd_mask_to_bias: null pointer");
  return finish(launch_mask_to_bias(mask, bias, (long)n, S(stream)), "mi355x_sd_mask_to_bias");
}

int mi355x_sd_cfg_axpby(const float* x, const float* eps_uncond, const float* eps_text, float* out, const float* coef,
                        float guidance_scale, int64_t n, void* stream) {
  if (!x || !eps_uncond || !eps_text || !out || !coef) return fail(SD_ERR_INVALID, "mi355x_sd_cfg_axpby: null pointer");
  return finish(launch_cfg_axpby(x, eps_uncond, eps_text, out, coef, guidance_scale, (long)n, S(stream)), "mi355x_sd_cfg_axpby");
}

int mi355x_sd_graph_begin(void* stream) {
  if (hipStreamBeginCapture(S(stream), hipStreamCaptureModeThreadLocal) != hipSuccess)
    return finish(SD_ERR_HIP, "mi355x_sd_graph_begin");
  return SD_OK;
}

int mi355x_sd_graph_end(void* stream, void** graph_exec) {
  if (!graph_exec) return fail(SD_ERR_INVALID, "mi355x_sd_graph_end: null pointer");
  hipGraph_t graph = nullptr;
  if (hipStreamEndCapture(S(stream), &graph) != hipSuccess || !graph) return finish(SD_ERR_HIP, "mi355x_sd_graph_end");
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) return finish(SD_ERR_HIP, "mi355x_sd_graph_end(instantiate)");
  *graph_exec = exec;
  return SD_OK;
}

int mi355x_sd_graph_launch(void* graph_exec, void* stream) {
  if (!graph_exec) return fail(SD_ERR_INVALID, "mi355x_sd_graph_launch: null graph");
  if (hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(graph_exec), S(stream)) != hipSuccess)
    return finish(SD_ERR_HIP, "mi355x_sd_graph_launch");
  return SD_OK;
}

int mi355x_sd_graph_destroy(void* graph_exec) {
  if (graph_exec) (void)hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(graph_exec));
  return SD_OK;
}

int mi355x_sd_probe_layouts(float* out, void* stream) {
  if (!out) return fail(SD_ERR_INVALID, "mi355x_sd_probe_layouts: null pointer");
  hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, S(stream), out);
  return finish(hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP, "mi355x_sd_probe_layouts");
}

}  // extern "C"
