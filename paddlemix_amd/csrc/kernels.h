// Internal launch interfaces between the HIP kernels and the C ABI (capi.hip). Not installed.
#pragma once
#include "common.h"

namespace sd {

struct GemmArgs {
  const bf16* A;   // activations: rows x K (linear) or NHWC source tensor (conv)
  const bf16* W;   // [N][K], K contiguous
  void* C;         // bf16 (or fp32 when out_f32) rows x ldc
  int M, N, K;
  int lda;         // row / pixel stride of A in elements
  int ldc;
  // implicit-GEMM 3x3 convolution, pad 1 (conv != 0): K = 9*Cin ordered (ky, kx, cin)
  int conv;
  int Hs, Ws;      // stored source height / width
  int Ho, Wo;      // output height / width
  int Cin;
  int stride;      // 1 | 2
  int up;          // 1: nearest-2x upsample of the source folded into the gather
  int pad;         // 1: zero padding all round; 0: one row / column at the bottom / right only (MI355X_SD_PAD_BR)
  int kb64;        // conv K order: 0: k = tap*Cin + c ([O][3][3][Cin] weights); 1: k = ((c/64)*9 + tap)*64 + c%64 (MI355X_SD_CONV_KB64)
  // epilogue
  const float* bias;      // [N]
  const float* rowbias;   // [M / rows_per_batch][ld_rowbias] broadcast over rows of one batch item
  int rows_per_batch;
  int ld_rowbias;
  const bf16* R;          // residual rows x ldr (fp32 rows when r_f32: the fp32 residual stream, MI355X_SD_R_F32)
  int ldr;
  int r_f32;
  float out_scale;        // multiplies (acc + bias + rowbias + R)
  int geglu;              // W/bias rows interleaved [16 value | 16 gate]; writes N/2 columns value*gelu(gate)
  int out_f32;
  int silu;               // SiLU applied last
  int gelu_tanh;          // tanh-GELU applied last (FeedForward "gelu-approximate", attention.py:648-649)
  const float* gate;      // optional per-(batch, channel) gate: out = R + gate[m / rows_per_batch][n] * (acc + bias)
  int ld_gate;            //   (adaLN-Zero gated residual, attention.py:181-196)
  int a_rpb;              // A row m lives at (m / a_rpb) * a_bstride + (m % a_rpb) * lda   (0: plain m * lda)
  long a_bstride;
  const float* wscale;    // W is fp8 e4m3 [N][K] (1 byte / element) with per-output-channel scale: acc *= wscale[n]
  int w16;                // set by launch_gemm: W was widened to the 16-bit element type just in time (wscale still applies)
  // W8A8 with an e4m3 OUTPUT (ff.net.0 -> ff.net.2): row scale from the Cauchy-Schwarz bound |out[m][n]| <= a_l2[m] *
  // w_norm_max + bias_abs_max (no second pass over the row); C is bytes, oscale[m] receives the scale
  int c_wide;             // C rows allow 16-byte stores (set by launch_gemm / launch_gemm_f8)
  int out_f8;
  const float* a_l2;
  float w_norm_max, bias_abs_max;
  float* oscale;
  const float* ascale;    // W8A8: A and W are fp8 e4m3 (wscale per output channel, ascale per A row): acc *= ascale[m]*wscale[n]
  const float* rowstat;   // LayerNorm folded into the GEMM: per-row (rstd, -mean*rstd) from launch_row_stats and
  const float* wsum;      //   wsum[n] = sum_k W[n][k]: acc <- rstd[m]*acc - mean[m]*rstd[m]*wsum[n]   (before bias)
  int c_rpb;              // same remap for the C rows (joint text+image token buffers of the MMDiT attention)
  long c_bstride;
  unsigned long long* ts; // diagnostics (scripts/gemm_timeline.py, MI355X_SD_GEMM_TSTAMP=<device address>): per block 4 x 100-MHz
                          // wall-clock stamps (entry, first tile landed, K loop done, stores issued) + (XCC id, block id); NULL in production
  int gm;                 // tile rasterisation group (common.h tile_coords): set by the launchers (gemm_gm())
  int bias_acc;           // set by launch_gemm_pipe: the kernel starts its accumulators at bias[n] (no bias load in the epilogue)
  int epi_batch;          // set by the launchers: the general epilogue fetches a row-tile's operands in one batch (0: one group at a time)
  // split-K (filled in by launch_gemm, callers leave 0): blockIdx.y owns k-tiles [y*kc, (y+1)*kc) and stores its raw
  // fp32 accumulators to ws[y][M][N]; splitk_reduce_kernel sums the slices in fixed order and runs the epilogue.
  int splitk, kc;
  float* ws;
  // the CALL's scratch (ABI 12: an argument of the four GEMM-class entry points, owned by the caller / by the handle that plans
  // the call -- no process-wide binding): split-K partial sums or the just-in-time widened copy of an e4m3 matrix. NULL / 0: the
  // launch takes neither path.
  void* ws_base;
  size_t ws_bytes;
};
void set_last_error(const char* msg);   // text behind mi355x_sd_last_error() (capi.hip), for the entry points defined elsewhere
int launch_gemm(const GemmArgs& a, hipStream_t stream);
int gemm_gm();   // tile rasterisation group of the GEMM kernels (-4 = column groups of 4)
int launch_gemm256(const GemmArgs& a, hipStream_t stream);
bool gemm_small_applies(const GemmArgs& a);                  // 64 x 64 tile, six-stage ring (gemm_small.hip)
int launch_gemm_small(const GemmArgs& a, hipStream_t stream);
bool gemm_w4_applies(const GemmArgs& a, int bn = 256);       // four-wave 256 x 256 / 256 x 160 tile (gemm_w4.hip)
int launch_gemm_w4(const GemmArgs& a, hipStream_t stream, int bn = 256);   // SD_ERR_UNSUPPORTED: caller falls back
int launch_gemm_f8(const GemmArgs& a, hipStream_t stream);   // W8A8 (gemm256.hip); validates
void launch_splitk_reduce(const GemmArgs& a, hipStream_t stream);   // sums a.splitk slices of a.ws + epilogue (gemm.hip)   // phased 256x256 kernel (gemm256.hip); args pre-validated

struct AttnArgs {
  const bf16 *Q, *K, *V;
  bf16* O;
  int B, H, Sq, Skv, D;
  // element strides; head h starts at + h*D inside a token row
  long q_bs, k_bs, v_bs, o_bs;   // batch strides
  int q_ts, k_ts, v_ts, o_ts;    // token strides
  const float* bias;             // optional additive mask, indexed b*bias_bs + h*bias_hs + q*bias_qs + kv
  long bias_bs, bias_hs, bias_qs;
  float scale;
  float accum;   // != 0: O += accum * attention(...) instead of O = attention(...) (IP-Adapter's second key set)
  int log2;      // MI355X_SD_SDPA_LOG2: q already carries scale * log2(e) -- scores are base-2 exponents (D == 64, no mask)
};
int launch_attention(const AttnArgs& a, hipStream_t stream);

// GroupNorm over NHWC rows: stats -> per-(batch, channel) scale/shift, then fused normalise(+SiLU)
int launch_groupnorm_stats(const void* x, int x_f32, int B, int HW, int C, int ldx, int groups, float eps, const float* gamma,
                           const float* beta, float* partial, float* scale_shift, hipStream_t stream);
int groupnorm_partial_floats(int B, int HW, int C);
int launch_scale_shift_act(const void* x, int x_f32, int B, int HW, int C, int ldx, const float* scale_shift, int silu, bf16* y,
                           int ldy, bf16* raw16, int ld_raw, hipStream_t stream);
// GroupNorm (+SiLU) in one launch where a (batch, group) chunk fits a block's registers (norm.hip gn_fused_kernel); *_fits: 1 / 0
int groupnorm_act_fits(int HW, int C, int groups);
int launch_groupnorm_act(const bf16* x, int B, int HW, int C, int ldx, int groups, float eps, const float* gamma, const float* beta,
                         int silu, bf16* y, int ldy, hipStream_t stream);
// y = LN(x) * (1 + scale[b]) + shift[b]  (no affine; b = row / rows_per_batch): AdaLayerNormZero / Continuous
// mod16 != 0: scale / shift (and gate / weight / bias below) are the build's 16-bit element type instead of fp32
int launch_adaln(const bf16* x, int rows, int C, int ldx, const void* scale, const void* shift, int ld_mod, int mod16,
                 int rows_per_batch, float eps, bf16* y, int ldy, hipStream_t stream);
int launch_adaln_f8(const bf16* x, int rows, int C, int ldx, const float* scale, const float* shift, int ld_mod,
                    int rows_per_batch, float eps, unsigned char* y8, int ldy, float* yscale, float* yl2, hipStream_t stream);
int launch_quantize_rows(const bf16* x, long rows, int C, int ldx, int x_rpb, long x_bstride, unsigned char* y8, int ldy,
                         float* yscale, hipStream_t stream);
// the reference's fused custom ops with their own signatures (fused_ops.hip; triton_ops.py:758-920, 1692-1752)
int launch_fused_adaln_scale_residual(const bf16* x, int ldx, const bf16* mha, int ldm, const void* gate, const void* scale,
                                      const void* shift, int ld_mod, int mod16, int rows_per_batch, const void* weight,
                                      const void* bias, float eps, int rows, int C, bf16* resi, int ldr, bf16* out, int ldo,
                                      hipStream_t stream);
int launch_split_concat(const bf16* x, const bf16* y, bf16* o0, bf16* o1, bf16* o2, int B, int S1, int S2, int C, hipStream_t stream);
int launch_patchify(const float* x_nchw, int B, int C, int H, int W, int p, bf16* out, int ldo, hipStream_t stream);
int launch_unpatchify(const bf16* x, int ldx, int B, int C, int H, int W, int p, float* out_nchw, hipStream_t stream);
int launch_row_stats(const bf16* x, int rows, int C, int ldx, float eps, float* stats, hipStream_t stream);
int launch_layernorm(const void* x, int x_f32, int rows, int C, int ldx, const float* gamma, const float* beta, float eps, bf16* y,
                     int ldy, hipStream_t stream);

// small ops
int launch_timestep_embedding(const float* t, int t_count, int n, int dim, int group, int flip_sin_to_cos,
                              float freq_shift, float scale, float max_period, bf16* out, int ldo, hipStream_t stream);
int launch_silu(const void* x, void* y, long n, int in_f32, int out_f32, hipStream_t stream);
int launch_conv_in3x3(const float* x_nchw, const float* in_scale, const bf16* w, const float* bias, void* y, int out_f32, int B,
                      int Cin, int H, int W, int Cout, int ldy, hipStream_t stream);
int launch_conv_out3x3(const bf16* x, int ldx, const bf16* w, const float* bias, float* y_nchw, int B, int Cin, int H,
                       int W, int Cout, hipStream_t stream);
int launch_cast_rows(const float* x, int ldx, bf16* y, int ldy, long rows, int C, hipStream_t stream);
int launch_add_nchw(void* x, int x_f32, int ldx, const float* r, int B, int C, long HW, hipStream_t stream);
int launch_latent_dist(const float* m, int ld, int B, int L, long HW, const float* noise, float out_scale, float* mean,
                       float* logvar, float* sample, hipStream_t stream);
int launch_embed_tokens(const int* ids, long n_tokens, int seq_len, const bf16* tok, const bf16* pos, int D, bf16* out,
                        int ldo, hipStream_t stream);
int launch_gated_activation(const bf16* x, int ldx, bf16* y, int ldy, long rows, int F, int kind, hipStream_t stream);
int launch_rmsnorm(const bf16* x, int rows, int C, int ldx, const float* weight, float eps, bf16* y, int ldy,
                   hipStream_t stream);
int launch_activation(const bf16* x, bf16* y, long n, int kind, hipStream_t stream);
int launch_conv1x1_nchw(const float* x, float in_scale, const bf16* w, const float* bias, float* y, int B, int Cin,
                        int Cout, long HW, hipStream_t stream);
int launch_softmax_rows(const float* x, long ldx, bf16* y, long ldy, long rows, int n, hipStream_t stream);
int launch_copy_rows(const bf16* x, int ldx, bf16* y, int ldy, long rows, int C, hipStream_t stream);
int launch_cfg_axpby(const float* x, const float* eu, const float* et, float* out, const float* coef, float gs, long n,
                     hipStream_t stream);
int launch_axpby(const float* x, const float* y, float* out, const float* coef, long n, hipStream_t stream);
int launch_mask_to_bias(const float* mask, float* bias, long n, hipStream_t stream);

}  // namespace sd
