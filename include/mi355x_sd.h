/* mi355x_sd.h — C ABI of libmi355x_sd.so: the MI355X (gfx950) implementation of the ppdiffusers
 * Stable-Diffusion denoising hot path (UNet2DConditionModel forward).
 *
 * This is the drop-in boundary.  Every entry point is `extern "C"`, takes plain device pointers, sizes and a
 * hipStream_t (passed as void*), allocates nothing, is stream-ordered and is safe to capture in a hipGraph.
 * All functions return 0 on success or a non-zero MI355X_SD_ERR_* code; mi355x_sd_last_error() gives the text.
 *
 * The reference (PaddlePaddle/PaddleMIX @ 2024-10-24) reaches this arithmetic through Paddle ops and its own
 * custom-op mechanism; each entry point cites the reference interface it replaces (paths relative to the
 * reference checkout; PPD/ = ppdiffusers/ppdiffusers/):
 *   - the custom-op ABI itself: PD_BUILD_OP(...).SetKernelFn(...) + paddle::Tensor arguments,
 *     paddlemix/triton_ops/triton_ops.py:641-693, 926-972; Paddle is not available for AMD here, so the tensor
 *     arguments become (pointer, shape, stride) triples.  INTEGRATION.md shows the PD_BUILD_OP / ctypes stub a
 *     maintainer would add on the reference side.
 *
 * Conventions
 *   - activations: bf16, token-major rows [rows][C] == NHWC, with an explicit row stride `ld*` in elements
 *     (a channel concat is two producers writing one buffer); C and all strides multiples of 8.
 *   - weights: bf16 [N][K] with K contiguous (a Paddle Linear weight [in,out] transposed once at load;
 *     a conv weight OIHW repacked to [O][kh][kw][I]); biases / norm affine parameters: fp32.
 *   - accumulation and all statistics in fp32.
 */
#ifndef MI355X_SD_H
#define MI355X_SD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MI355X_SD_ABI_VERSION 12
#define MI355X_SD_OK 0
#define MI355X_SD_ERR_INVALID 1      /* bad argument (shape <= 0, null pointer ...)          */
#define MI355X_SD_ERR_UNSUPPORTED 2  /* well-formed but outside the implemented configurations */
#define MI355X_SD_ERR_HIP 3          /* HIP runtime error; see mi355x_sd_last_error()          */

int mi355x_sd_abi_version(void);
/* 16-bit element type of this build: the library comes in two builds of the same sources, libmi355x_sd.so (bfloat16,
 * the dtype of BASELINE.json's configurations) and libmi355x_sd_f16.so (IEEE half: same speed, 3 more mantissa bits,
 * the dtype the reference's own GPU pipelines default to -- paddle_dtype=paddle.float16). Wherever this header says
 * "bf16" read "the element type of the build". */
#define MI355X_SD_ELEM_BF16 0
#define MI355X_SD_ELEM_F16 1
int mi355x_sd_elem_dtype(void);
const char* mi355x_sd_last_error(void);
/* Selects `device` and verifies it is a gfx950 part. */
int mi355x_sd_init(int device);

/* SCRATCH OF THE GEMM-CLASS CALLS (ABI 12): mi355x_sd_linear, _linear_ex, _linear_ln and mi355x_sd_conv3x3 take `ws, ws_bytes` --
 * caller-owned device memory, 16-byte aligned, or NULL / 0 -- as ARGUMENTS (until ABI 11 a process-wide binding,
 * mi355x_sd_set_workspace, whose stale pointer a later call could write through; it is gone). The scratch belongs to the call:
 * launches that run concurrently (two streams, two handles) must be given different buffers, launches ordered on one stream may
 * share one. The model handles own theirs (mi355x_sd_unet_*: part of the arena bound with _bind_workspace; mi355x_sd_program_*: a
 * scratch region of the program). Two uses:
 *   split-K: launches that cannot fill the 256 CUs (batch-1 SD-1.5: 64..1024 rows against K up to 23040) split K over blockIdx.y,
 *     store fp32 slices here and reduce them in fixed order (deterministic); the split count is bounded by ws_bytes, so the same
 *     call with another ws_bytes may sum in another order (results equal to fp32 rounding, bit-stable for a fixed size);
 *   weight-only fp8 (mi355x_sd_linear_ex, w_scale != NULL): launches of M >= 4096 rows and more than 128 tiles widen the e4m3
 *     matrix once, just in time, into the first 2 * N * K bytes (up to ~19 MB for SD3-medium's 6144 x 1536 matrices) and multiply
 *     with the 16-bit kernels; a smaller scratch is not an error -- the launch takes the slower path that converts in the fragment
 *     load (same result to fp32 rounding).
 * NULL: neither path. 32 MiB covers every model of this repository. The library never allocates ("no hidden allocation"). A
 * pointer the HIP runtime does not know as device (or managed) memory is refused with MI355X_SD_ERR_INVALID before anything is
 * launched: the kernels write through it. */

/* ---- seam B1: the whole UNet2DConditionModel behind one handle (paddlemix_amd/csrc/unet_exec.hip) ----------------------------
 * What a compiled host (Paddle C++, a serving runtime) binds instead of the per-op entry points: the reference calls its UNet once
 * per denoising step as `unet(sample, t, encoder_hidden_states, added_cond_kwargs=...)` (pipeline_stable_diffusion.py:866-879) and
 * its deployment pipelines already treat it as an opaque predictor with named inputs (PaddleInferRuntimeModel.__call__,
 * PPD/models/paddleinfer_runtime.py:47-126; input names PPD/../deploy/sd15/export_model.py:78-90). Life cycle:
 *   create(config.json text) -> load_weight(name, host pointer) for every parameter (reference names, Paddle layouts: Linear
 *   [in,out], conv OIHW; enumerate them with num_params / param_info) -> weight_bytes -> finalize_weights(caller's device buffer)
 *   -> plan(B, H, W, L) -> bind_workspace(caller's device buffer) -> forward(...) per step -> destroy.
 * The library never allocates device memory; all tensors of forward() are device pointers; split-K GEMMs use the workspace of
 * mi355x_sd_set_workspace like the per-op calls. Built: the four SD block types, conv / linear projections, addition_embed_type
 * None | "text_time"; anything else in the config is refused at create (MI355X_SD_ERR_UNSUPPORTED), never ignored. */
#define MI355X_SD_DTYPE_F32 0
#define MI355X_SD_DTYPE_BF16 1
#define MI355X_SD_DTYPE_F16 2
int mi355x_sd_unet_create(const char* config_json, void** handle);
int mi355x_sd_unet_destroy(void* handle);
/* options: "residual_f32" = 1 (before plan) keeps the residual stream in fp32 (MI355X_SD_R_F32); "fold_softmax_scale" = 0 | 1 (default 1;
 * before the weights are packed): head_dim^-0.5 * log2(e) is folded into the self-attention to_q weights (head_dim 64) and those attentions
 * run as MI355X_SD_SDPA_LOG2 */
int mi355x_sd_unet_set_option(void* handle, const char* key, int value);
int mi355x_sd_unet_num_params(void* handle);
int mi355x_sd_unet_param_info(void* handle, int index, const char** name, int64_t* shape4, int* ndim);
int mi355x_sd_unet_load_weight(void* handle, const char* name, const void* host_ptr, const int64_t* shape, int ndim, int dtype);
int mi355x_sd_unet_weight_bytes(void* handle, size_t* bytes);
int mi355x_sd_unet_finalize_weights(void* handle, void* device_buffer, size_t bytes, void* stream);
/* finalize_weights in two halves for hosts that stage uploads themselves: the packed image into HOST memory, then the device
 * address it was copied to. packed_tensor locates one packed tensor inside the image (keys as in paddlemix_amd/unet.py). */
int mi355x_sd_unet_pack_weights(void* handle, void* host_buffer, size_t bytes);
int mi355x_sd_unet_attach_weights(void* handle, void* device_buffer, size_t bytes);
int mi355x_sd_unet_packed_tensor(void* handle, const char* key, size_t* offset, size_t* bytes, int* rows, int* cols);
int mi355x_sd_unet_plan(void* handle, int B, int H, int W, int L, size_t* workspace_bytes);
int mi355x_sd_unet_bind_workspace(void* handle, void* device_ptr, size_t bytes);
int mi355x_sd_unet_num_launches(void* handle);
/* All pointers are device memory, fp32: sample [B,Cin,H,W], timestep [1], encoder_hidden_states [B,L,D], text_embeds
 * [B, projection_class_embeddings_input_dim - 6*addition_time_embed_dim] and time_ids [B,6] (text_time only, else NULL), in_scale
 * optional scalar multiplied into the sample (scheduler.scale_model_input; NULL = 1), out [B,Cout,H,W]. Missing text_embeds /
 * time_ids on a text_time model is MI355X_SD_ERR_INVALID (the reference's ValueError, unet_2d_condition.py:993-1001).
 * use_graph != 0: the launches are captured into a hipGraph on the first call and replayed afterwards. */
int mi355x_sd_unet_forward(void* handle, void* stream, const float* sample, const float* timestep,
                           const float* encoder_hidden_states, const float* text_embeds, const float* time_ids,
                           const float* in_scale, float* out, int use_graph);
/* The optional inputs of UNet2DConditionModel.forward that change the PROGRAM are chosen at plan time (plan_ex flags), their
 * tensors are handed over per call (forward_ex; a NULL / absent input whose flag was planned is MI355X_SD_ERR_INVALID, as is a
 * non-NULL one that was not planned):
 *   ENC_MASK   encoder_attention_mask [B, L], 1 = attend, 0 = masked: becomes the additive bias (1 - m) * -10000 of every
 *              cross-attention (unet_2d_condition.py:925-927)
 *   SELF_MASK  attention_mask [B, H*W]: the same for every self-attention (:916-923). Like the reference, this only works for a
 *              UNet whose attention levels all see H*W tokens; a level with another token count fails the plan (the reference
 *              fails on the shapes of the add)
 *   CONTROLNET down_block_additional_residuals (one fp32 NCHW tensor per skip connection, in the order the down path produces
 *              them: conv_in first) + mid_block_additional_residual (:1121-1132, 1151-1155), e.g. the outputs of
 *              ControlNetModel; both or neither
 * class_labels, timestep_cond and the IP-Adapter image_embeds: mi355x_sd_unet_set_input below. */
#define MI355X_SD_UNET_ENC_MASK 1
#define MI355X_SD_UNET_SELF_MASK 2
#define MI355X_SD_UNET_CONTROLNET 4
int mi355x_sd_unet_plan_ex(void* handle, int B, int H, int W, int L, int flags, size_t* workspace_bytes);
/* number of skip tensors = length of down_block_additional_residuals, and the [C, H, W] of skip i (i == count: the mid output) */
int mi355x_sd_unet_num_skips(void* handle);
int mi355x_sd_unet_skip_shape(void* handle, int index, int* C, int* H, int* W);
int mi355x_sd_unet_forward_ex(void* handle, void* stream, const float* sample, const float* timestep,
                              const float* encoder_hidden_states, const float* text_embeds, const float* time_ids,
                              const float* in_scale, const float* encoder_attention_mask, const float* attention_mask,
                              const float* const* down_block_additional_residuals, int num_down_residuals,
                              const float* mid_block_additional_residual, float* out, int use_graph);
/* The tensor inputs of UNet2DConditionModel.forward that belong to the CONFIG -- a model built with a class embedding or with
 * time_cond_proj_dim reads them in every forward (unet_2d_condition.py:953-975; embeddings.py:284-285) -- are bound by name, not
 * passed per call: device pointers, read (copied into the plan's static buffers, on the call's stream) by every forward[_ex] call
 * from then on, until replaced; NULL unbinds.
 *   "class_labels"   class_embed_type null + num_class_embeds: int32 [B], the row of the nn.Embedding table per sample;
 *                    "timestep": fp32 [B] (goes through time_proj and a TimestepEmbedding); "projection" / "simple_projection":
 *                    fp32 [B, projection_class_embeddings_input_dim]; "identity": fp32 [B, 4 * block_out_channels[0]].
 *                    A forward call of such a model with nothing bound is MI355X_SD_ERR_INVALID (the reference's ValueError,
 *                    :954-955). class_embeddings_concat=true hands the blocks [emb | class_emb] (:443-449, 972-975); together with
 *                    addition_embed_type it is refused at create, as by the Python planner.
 *   "timestep_cond"  fp32 [B, time_cond_proj_dim] (the LCM guidance-scale embedding); unbound = the reference's None (no term).
 *   "image_embeds"   fp32 [B, encoder_hid_dim] of a model with encoder_hid_dim_type "ip_image_proj" (the CLIP image embedding of the
 *                    IP-Adapter prompt, added_cond_kwargs["image_embeds"], :1054-1061): projected to ip_adapter_num_tokens image
 *                    tokens (ImageProjection, embeddings.py:507-518) whose attention is added to every cross-attention output with
 *                    the IP-Adapter scale (IPAdapterAttnProcessor, attention_processor.py:1816-1900). Unbound at a forward call =
 *                    MI355X_SD_ERR_INVALID (the reference's ValueError).
 * Binding an input the model does not have is MI355X_SD_ERR_INVALID. */
int mi355x_sd_unet_set_input(void* handle, const char* name, const void* device_ptr);
/* IPAdapterAttnProcessor.scale of every cross-attention (the reference's set_ip_adapter_scale, loaders/ip_adapter.py; default 1; 0
 * drops the image-token attention launches). A constant of the planned launches: changing it discards the plan -- plan and bind again. */
int mi355x_sd_unet_set_ip_adapter_scale(void* handle, float scale);
/* ---- seam B2: any exported step program behind one handle (paddlemix_amd/csrc/program_exec.hip) --------------------------------
 * Every model of the path (UNet2DConditionModel, ControlNetModel, SD3Transformer2DModel, the DiT Transformer2DModel, AutoencoderKL
 * decode / encode, the CLIP and T5 text encoders) runs as a static list of the per-op launches below over weights + scratch. The
 * Python planners build that list; `paddlemix_amd.export.export_program(model, plan, path)` writes it to a file (entry-point
 * names, arguments with every device pointer as (region, offset), packed weights, plan-time constants, the named input / output
 * regions), and a host without Python replays it -- the shape of the reference's own deployment path: export offline
 * (PPD/../deploy/sd15/export_model.py:78-90), then run behind a predictor with named inputs
 * (PaddleInferRuntimeModel.__call__, PPD/models/paddleinfer_runtime.py:47-126: get_input_handle / copy_from_cpu / run / copy_to_cpu):
 *   load(path)  host only: parses and type-checks every launch against this library's entry points (refuses another ABI version
 *               or the other 16-bit element build) -> device_bytes -> bind(caller's device buffer, 256-byte aligned: uploads
 *               weights and constants, resolves the pointers) -> io_info(i): name, direction, dtype, shape and DEVICE ADDRESS of
 *               each input / output region (copy inputs there, in the dtype it names) -> run(stream) per call -> destroy.
 * The library allocates no device memory; run() binds the split-K workspace the program was planned with (inside the same
 * buffer), so it reproduces the exporting process bit for bit. Option "use_graph" = 1: the replay becomes one hipGraph. */
#define MI355X_SD_IO_F32 0
#define MI355X_SD_IO_ELEM16 1   /* the build's 16-bit element type (mi355x_sd_elem_dtype) */
#define MI355X_SD_IO_I32 2
#define MI355X_SD_IO_U8 3
int mi355x_sd_program_load(const char* path, void** handle);
int mi355x_sd_program_destroy(void* handle);
int mi355x_sd_program_set_option(void* handle, const char* key, int value);
int mi355x_sd_program_num_launches(void* handle);
int mi355x_sd_program_device_bytes(void* handle, size_t* bytes);
int mi355x_sd_program_bind(void* handle, void* device_buffer, size_t bytes, void* stream);
int mi355x_sd_program_num_io(void* handle);
int mi355x_sd_program_io_info(void* handle, int index, const char** name, int* is_output, int* dtype, int64_t* shape4, int* ndim,
                              size_t* bytes, void** device_ptr);
int mi355x_sd_program_run(void* handle, void* stream);

/* bias[i] = (1 - mask[i]) * -10000: the additive form of a keep-mask (unet_2d_condition.py:921-927) */
int mi355x_sd_mask_to_bias(const float* mask, float* bias, int64_t n, void* stream);

/* flags for mi355x_sd_linear / mi355x_sd_conv3x3 */
#define MI355X_SD_GEGLU 1    /* W/bias rows interleaved [16 value | 16 gate]; writes N/2 columns value*gelu_erf(gate) */
#define MI355X_SD_OUT_F32 2  /* C is fp32 */
#define MI355X_SD_GELU_TANH 8 /* tanh-GELU applied to the final value (FeedForward "gelu-approximate", PPD/models/attention.py:648-649) */
#define MI355X_SD_SILU 4     /* SiLU applied to the final value (TimestepEmbedding.act, PPD/models/embeddings.py:283-295) */
#define MI355X_SD_R_F32 32   /* R is fp32 rows (ldr in fp32 elements): the fp32 residual-stream mode, where the tensors the reference
                              * adds onto (`hidden_states + input_tensor` resnet.py:806, `attn_output + hidden_states` attention.py:430,
                              * 455, 487, transformer_2d.py:467) never round to 16 bits; combine with MI355X_SD_OUT_F32 */
#define MI355X_SD_CONV_KB64 64 /* conv3x3, Cin % 64 == 0: W is packed [O][Cin/64][3][3][64] instead of [O][3][3][Cin] -- the K loop then walks the 9
                              * taps of one 64-channel block back to back, so the eight re-reads of a [pixels x 64 channels] region hit the XCD's
                              * L2 (32 KB per block) instead of coming back through the fabric a full channel sweep later (measured: profiles/HISTORY.md section 5) */
#define MI355X_SD_PAD_BR 16  /* conv3x3, stride 2 only: zero padding is one row / column at the bottom / right instead of all round
                              * (Downsample2D with padding=0: F.pad (0,1,0,1) then an unpadded conv, PPD/models/resnet.py:277-279 --
                              * the VAE encoder's downsamplers, vae.py:113) */

/* C[M,N] = ((A[M,K] . W[N,K]^T) + bias[N] + rowbias[m / rows_per_batch][N] + R[M,N]) * out_scale
 * Replaces LoRACompatibleLinear.forward (PPD/models/lora.py:453-459) and, with conv1x1 weights, the 1x1
 * LoRACompatibleConv (lora.py:364-377); GEGLU = PPD/models/activations.py:101-104 fused into the epilogue;
 * R / out_scale = the residual adds of PPD/models/attention.py:429,459,485 and resnet.py:800-806.
 * bias, rowbias, R may be NULL. */
int mi355x_sd_linear(const void* A, int lda, const void* W, void* C, int ldc, int M, int N, int K,
                     const float* bias, const float* rowbias, int rows_per_batch, int ld_rowbias,
                     const void* R, int ldr, float out_scale, int flags, void* ws, size_t ws_bytes, void* stream);

/* mi355x_sd_linear plus what the SD3 MMDiT blocks (PPD/models/attention.py:164-214, attention_processor.py:916-983) need:
 *   gate  : out = R + gate[m / rows_per_batch][n] * (acc + bias)     (adaLN-Zero gated residuals, attention.py:181-196)
 *   a_/c_ rows_per_batch + batch_stride: row m of A (C) lives at (m / rpb) * batch_stride + (m % rpb) * lda (ldc) -- the
 *   w_scale != NULL: W is OCP fp8 e4m3 [N][K] (one byte per element) with a per-output-channel fp32 scale, W ~ w_scale[n] * q;
 *     every w_scale[n] must be > 0 (clamp an all-zero channel's absmax / 448 to a tiny positive number, as the library's own
 *     quantiser does): the widened path starts the accumulators at bias[n] / w_scale[n];
 *   the kernel widens q to bf16 in registers (exact) and scales the fp32 accumulator (weight-only fp8: BASELINE config 5).
 *   projections of the image and the text tokens read / write one joint [B, S_img + S_txt, .] buffer in place, which is
 *   the fused split + concat of paddlemix/triton_ops/triton_ops.py:1652-1752 (split_concat) folded into the GEMMs. */
int mi355x_sd_linear_ex(const void* A, int lda, int a_rows_per_batch, int64_t a_batch_stride, const void* W,
                        const float* w_scale, void* C,
                        int ldc, int c_rows_per_batch, int64_t c_batch_stride, int M, int N, int K, const float* bias,
                        const float* rowbias, int ld_rowbias, const float* gate, int ld_gate, int rows_per_batch,
                        const void* R, int ldr, float out_scale, int flags, void* ws, size_t ws_bytes, void* stream);

/* LayerNorm folded into the consuming projection (BasicTransformerBlock: norm1 -> attn1.to_q/k/v, norm2 -> attn2.to_q,
 * norm3 -> ff.net.0.proj; PPD/models/attention.py:405-486).  With W' = W diag(gamma) (bf16), w_rowsum[n] = sum_k W'[n][k]
 * and bias' = bias + W beta prepared at load time,
 *     LN(x) W^T + bias  =  rstd[m] * (x W'^T)[m][n] - mean[m] * rstd[m] * w_rowsum[n] + bias'[n],
 * so the normalised activation is never written to or re-read from HBM: mi355x_sd_row_stats reads the rows once and
 * stores row_stats[m] = (rstd, -mean * rstd) (fp32 pairs), mi355x_sd_linear_ln multiplies the RAW rows and applies the
 * correction to its fp32 accumulators before bias / GEGLU / activation flags. */
int mi355x_sd_row_stats(const void* x, int rows, int C, int ldx, float eps, float* row_stats, void* stream);
int mi355x_sd_linear_ln(const void* A, int lda, const float* row_stats, const void* W, const float* w_rowsum, void* C,
                        int ldc, int M, int N, int K, const float* bias, int flags, void* ws, size_t ws_bytes, void* stream);

/* ---- W8A8: fp8 (OCP e4m3) MFMA path of the MMDiT block GEMMs (BASELINE config 5: "fp8 weights ... CDNA4 fp8 MFMA") ----
 * C[M,N] = (A8 . W8^T) * a_scale[m] * w_scale[n] + bias, then optional gate[m / rows_per_batch][n] * (.) + R, optional
 * tanh-GELU; bf16 output. A8 [M,K] / W8 [N,K] are e4m3 bytes, K % 128 == 0 (one v_mfma_scale_f32_16x16x128_f8f6f4 per
 * 16x16 tile and K-tile), lda in elements (= bytes), % 16; a_rows_per_batch / c_rows_per_batch as in mi355x_sd_linear_ex.
 * Scales: value = scale * q with scale = absmax / 448 per A row (mi355x_sd_adaln_f8 / mi355x_sd_quantize_rows) and per
 * output channel of W (quantised at load). The reference has no fp8 inference path; parity is defined against the oracle
 * evaluated on the same quantised operands (tests/test_gpu_sd3.py) and the quantisation error itself is reported. */
int mi355x_sd_linear_f8(const void* A8, int lda, int a_rows_per_batch, int64_t a_batch_stride, const float* a_scale,
                        const void* W8, const float* w_scale, void* C, int ldc, int c_rows_per_batch,
                        int64_t c_batch_stride, int M, int N, int K, const float* bias, const float* gate, int ld_gate,
                        int rows_per_batch, const void* R, int ldr, int flags, void* stream);
/* mi355x_sd_adaln with the e4m3 quantisation fused: y8[row][C] bytes (row stride ldy bytes), y_scale[row]; y_l2
 * (optional) receives the row's L2 norm before quantisation (see mi355x_sd_linear_f8_q). */
int mi355x_sd_adaln_f8(const void* x, int rows, int C, int ldx, const float* scale, const float* shift, int ld_mod,
                       int rows_per_batch, float eps, void* y8, int ldy, float* y_scale, float* y_l2, void* stream);
/* W8A8 GEMM with an e4m3 OUTPUT for the next GEMM (ff.net.0 -> tanh-GELU -> ff.net.2 without a re-quantisation pass):
 * C8[m][n] = e4m3(act(acc * a_scale[m] * w_scale[n] + bias[n]) / c_scale[m]). The output row scale needs no pass over
 * the row: |acc + bias| <= a_l2[m] * w_norm_max + bias_abs_max (Cauchy-Schwarz; w_norm_max = max_n ||W[n]||_2 of the
 * dequantised weights, a_l2 from mi355x_sd_adaln_f8), |gelu(x)| <= |x|, so c_scale[m] = 1.1 * that bound / 448 cannot
 * overflow; e4m3 is a floating-point format, a scale that is ~10x loose costs precision only for elements ~100x below
 * the typical magnitude. ldc in bytes, % 8. */
int mi355x_sd_linear_f8_q(const void* A8, int lda, const float* a_scale, const float* a_l2, const void* W8,
                          const float* w_scale, float w_norm_max, void* C8, int ldc, float* c_scale, int M, int N, int K,
                          const float* bias, float bias_abs_max, int flags, void* stream);
/* bf16 rows -> e4m3 rows + per-row scale (attention output in front of the output projections, GELU output in front
 * of ff.net.2). Source row m lives at (m / x_rows_per_batch) * x_batch_stride + (m % x_rows_per_batch) * ldx
 * (x_rows_per_batch = 0: plain m * ldx); the output rows and scales are compact. */
int mi355x_sd_quantize_rows(const void* x, int64_t rows, int C, int ldx, int x_rows_per_batch, int64_t x_batch_stride,
                            void* y8, int ldy, float* y_scale, void* stream);

/* y = LayerNorm_noaffine(x) * (1 + scale[b]) + shift[b], b = row / rows_per_batch, scale/shift fp32 rows of stride ld_mod.
 * AdaLayerNormZero / AdaLayerNormContinuous (PPD/models/normalization.py:72-86, 190-202) and the Triton op
 * adaptive_layer_norm (paddlemix/triton_ops/triton_ops.py:981-1139). */
int mi355x_sd_adaln(const void* x, int rows, int C, int ldx, const float* scale, const float* shift, int ld_mod,
                    int rows_per_batch, float eps, void* y, int ldy, void* stream);
/* Element type of the modulation (gate / scale / shift) and affine (weight / bias) vectors of the _ex forms below.
 * MI355X_SD_MOD_ELEM = the build's 16-bit element type: what the REFERENCE's ops receive -- gate_msa / scale_mlp / shift_mlp are
 * chunks of a 16-bit linear output and weight / bias 16-bit parameters, all of x.dtype (paddlemix/triton_ops/triton_ops.py:777-786,
 * call sites PPD/models/simplified_sd3.py:62-76); ld_mod % 8 == 0 and 16-byte aligned vectors then. MI355X_SD_MOD_F32: fp32
 * vectors (ld_mod % 4 == 0), what this library's own SD3 / DiT programs keep them in. */
#define MI355X_SD_MOD_F32 0
#define MI355X_SD_MOD_ELEM 1
int mi355x_sd_adaln_ex(const void* x, int rows, int C, int ldx, const void* scale, const void* shift, int ld_mod, int mod_dtype,
                       int rows_per_batch, float eps, void* y, int ldy, void* stream);
/* PatchEmbed.proj operand (PPD/models/embeddings.py:148-155, 209-219): NCHW fp32 -> rows [B*(H/p)*(W/p), C*p*p] bf16,
 * columns ordered (c, py, px) like the flattened conv weight; and the inverse at the output (transformer_sd3.py:349-356):
 * rows [B*h*w, p*p*C] ordered (py, px, c) -> NCHW fp32. */
int mi355x_sd_patchify(const float* x_nchw, int B, int C, int H, int W, int patch, void* out, int ldo, void* stream);
int mi355x_sd_unpatchify(const void* x, int ldx, int B, int C, int H, int W, int patch, float* out_nchw, void* stream);

/* 3x3 convolution, padding 1 (or bottom/right only with MI355X_SD_PAD_BR), stride 1|2, as an implicit GEMM over an NHWC source [B][Hs][Ws][ldx>=Cin];
 * `upsample` = 1 folds F.interpolate(scale_factor=2, mode="nearest") (PPD/models/resnet.py:169-218) into the
 * gather.  W is [Cout][3][3][Cin].  Output rows = B*Ho*Wo with Ho = ((Hs<<upsample) + 2 - 3)/stride + 1
 * (PAD_BR: ((Hs + 1 - 3)/2 + 1).
 * Replaces LoRACompatibleConv.forward (lora.py:364-377) as used by ResnetBlock2D (resnet.py:770,798, temb add
 * :772-784 via rowbias), Downsample2D (:271-294) and Upsample2D. */
int mi355x_sd_conv3x3(const void* X, int ldx, int B, int Hs, int Ws, int Cin, int stride, int upsample,
                      const void* W, void* C, int ldc, int Cout,
                      const float* bias, const float* rowbias, int ld_rowbias,
                      const void* R, int ldr, float out_scale, int flags, void* ws, size_t ws_bytes, void* stream);

/* out = softmax(q k^T * scale + bias) v, layouts q [B,Sq,H,D], k/v [B,Skv,H,D], out [B,Sq,H,D] with explicit
 * batch (bs) and token (ts) strides in elements; bias optional fp32 additive mask addressed
 * b*bias_bs + h*bias_hs + q*bias_qs + kv (0 strides broadcast).  D % 8 == 0, D <= 160.
 * Replaces scaled_dot_product_attention_ (PPD/patches/paddle_patch.py:414-529) and the body of
 * AttnProcessor.__call__ / get_attention_scores (PPD/models/attention_processor.py:673-735, 552-586). */
int mi355x_sd_sdpa(const void* q, const void* k, const void* v, const float* bias, void* out,
                   int B, int H, int Sq, int Skv, int D,
                   int64_t q_bs, int q_ts, int64_t k_bs, int k_ts, int64_t v_bs, int v_ts, int64_t o_bs, int o_ts,
                   int64_t bias_bs, int64_t bias_hs, int64_t bias_qs, float scale, void* stream);

/* out += out_scale * softmax(q k^T * scale + bias) v  -- same layouts as mi355x_sd_sdpa, `out` already holding the result of
 * a first attention over another key set. IPAdapterAttnProcessor.__call__ (PPD/models/attention_processor.py:1819-1901):
 * hidden = attn(q, to_k(text), to_v(text)) + self.scale * attn(q, to_k_ip(image tokens), to_v_ip(image tokens)) (:1871-1886). */
/* mi355x_sd_sdpa with flags. MI355X_SD_SDPA_LOG2: the caller has folded scale * log2(e) into the queries (e.g. into the to_q
 * weights), so q.k IS the base-2 exponent: out = softmax_2(q k^T) v with softmax_2(x) = 2^x / sum 2^x; `scale` is ignored, no mask,
 * D == 64. Saves the multiply-add per score in front of every exponential (profiles/HISTORY.md section 5). */
#define MI355X_SD_SDPA_LOG2 1
int mi355x_sd_sdpa_ex(const void* q, const void* k, const void* v, const float* bias, void* out, int B, int H, int Sq, int Skv, int D,
                      int64_t q_bs, int q_ts, int64_t k_bs, int k_ts, int64_t v_bs, int v_ts, int64_t o_bs, int o_ts,
                      int64_t bias_bs, int64_t bias_hs, int64_t bias_qs, float scale, int flags, void* stream);
int mi355x_sd_sdpa_accum(const void* q, const void* k, const void* v, const float* bias, void* out,
                   int B, int H, int Sq, int Skv, int D,
                   int64_t q_bs, int q_ts, int64_t k_bs, int k_ts, int64_t v_bs, int v_ts, int64_t o_bs, int o_ts,
                   int64_t bias_bs, int64_t bias_hs, int64_t bias_qs, float scale, float out_scale, void* stream);

/* GroupNorm statistics -> scale_shift[B][2][C] fp32 (scale = gamma*rstd, shift = beta - mean*scale); x is
 * [B][HW][ldx>=C].  `workspace` needs mi355x_sd_groupnorm_workspace_floats(B,HW,C) floats.
 * paddle.nn.GroupNorm as used at PPD/models/resnet.py:739,789; transformer_2d.py:359; unet_2d_condition.py:1194. */
int mi355x_sd_groupnorm_workspace_floats(int B, int HW, int C);
int mi355x_sd_groupnorm_stats(const void* x, int B, int HW, int C, int ldx, int groups, float eps,
                              const float* gamma, const float* beta, float* workspace, float* scale_shift,
                              void* stream);
/* GroupNorm (+SiLU when silu != 0) of 16-bit rows in ONE launch: x, y [B][HW][ld >= C]. Only where a (batch, group) chunk fits a
 * block's registers: HW * (C / groups) * 2 bytes <= 96 KiB and C / groups even (mi355x_sd_groupnorm_act_fits returns 1; else
 * MI355X_SD_ERR_UNSUPPORTED -- use the stats + scale_shift_act pair). Same arithmetic as the pair (fp32 sums per thread, double
 * mean / variance, y = x * (gamma * rstd) + (beta - mean * gamma * rstd)); the summation order differs, so results agree with the
 * pair to fp32 rounding of the statistics, not bit for bit. */
int mi355x_sd_groupnorm_act_fits(int HW, int C, int groups);
int mi355x_sd_groupnorm_act(const void* x, int B, int HW, int C, int ldx, int groups, float eps, const float* gamma,
                            const float* beta, int silu, void* y, int ldy, void* stream);
/* y = act(x*scale[b][c] + shift[b][c]), act = SiLU when silu != 0 (the fused GN+SiLU of resnet.py:739-741). */
int mi355x_sd_scale_shift_act(const void* x, int B, int HW, int C, int ldx, const float* scale_shift, int silu,
                              void* y, int ldy, void* stream);
/* Row LayerNorm with optional affine (PPD/models/attention.py:397,442,463). C <= 2560. */
int mi355x_sd_layernorm(const void* x, int rows, int C, int ldx, const float* gamma, const float* beta, float eps,
                        void* y, int ldy, void* stream);
/* The same three with x_f32 != 0: x is fp32 rows (ldx in fp32 elements) -- the norms are where the fp32 residual stream
 * becomes a 16-bit MFMA operand. scale_shift_act_ex can also emit raw16 (NULL: none) = the 16-bit rounding of the raw x
 * rows in the same pass: the operand of the resnet's conv_shortcut GEMM (resnet.py:797-798). */
int mi355x_sd_groupnorm_stats_ex(const void* x, int B, int HW, int C, int ldx, int groups, float eps,
                                 const float* gamma, const float* beta, float* workspace, float* scale_shift,
                                 int x_f32, void* stream);
int mi355x_sd_scale_shift_act_ex(const void* x, int B, int HW, int C, int ldx, const float* scale_shift, int silu,
                                 void* y, int ldy, int x_f32, void* raw16, int ld_raw, void* stream);
int mi355x_sd_layernorm_ex(const void* x, int rows, int C, int ldx, const float* gamma, const float* beta, float eps,
                           void* y, int ldy, int x_f32, void* stream);
/* y (16-bit rows) = x (fp32 rows): operand copies of the fp32 residual stream (inputs of Downsample2D / Upsample2D convs). */
int mi355x_sd_cast_rows(const float* x, int ldx, void* y, int ldy, int64_t rows, int C, void* stream);

/* ---- the reference's fused custom ops, with THEIR signatures (seam B4; paddlemix/triton_ops/triton_ops.py) ----
 * fused_adaLN_scale_residual(x, mha_out, gate_msa, scale_mlp, shift_mlp, weight, bias, epsilon) -> (resi_out, adaLN_out)
 * (triton_ops.py:758-920; unfused definition :842-847): resi_out = mha_out * gate[b] + x, adaLN_out = layer_norm(resi_out, weight,
 * bias, eps) * (1 + scale[b]) + shift[b], b = row / rows_per_batch. x, mha_out, resi_out, adaLN_out: 16-bit rows [rows][ld >= C];
 * gate / scale / shift: fp32 [rows / rows_per_batch][ld_mod >= C]; weight / bias fp32 [C] or NULL. C % 8 == 0, C <= 4096.
 * (fp32 vectors: this library's own convention; the reference passes them in x.dtype -- use the _ex form below for a one-for-one swap) */
int mi355x_sd_fused_adaln_scale_residual(const void* x, int ldx, const void* mha_out, int ld_mha, const float* gate_msa,
                                         const float* scale_mlp, const float* shift_mlp, int ld_mod, int rows_per_batch,
                                         const float* weight, const float* bias, float epsilon, int rows, int C,
                                         void* resi_out, int ld_resi, void* adaln_out, int ld_out, void* stream);
/* The one-for-one form: gate / scale / shift / weight / bias in mod_dtype (MI355X_SD_MOD_ELEM = the tensors exactly as the reference
 * hands them to fused_adaLN_scale_residual: all of x.dtype). Arithmetic in fp32 either way. */
int mi355x_sd_fused_adaln_scale_residual_ex(const void* x, int ldx, const void* mha_out, int ld_mha, const void* gate_msa,
                                            const void* scale_mlp, const void* shift_mlp, int ld_mod, int mod_dtype,
                                            int rows_per_batch, const void* weight, const void* bias, float epsilon, int rows,
                                            int C, void* resi_out, int ld_resi, void* adaln_out, int ld_out, void* stream);
/* split_concat(x [B,S1,3C], y [B,S2,3C]) -> q, k, v, each [B, S1+S2, C]: chunk i of x followed by chunk i of y along the sequence
 * (triton_ops.py:1692-1752); dense 16-bit tensors, C % 8 == 0. */
int mi355x_sd_split_concat(const void* x, const void* y, void* q_out, void* k_out, void* v_out, int B, int S1, int S2, int C,
                           void* stream);

/* get_timestep_embedding (PPD/models/embeddings.py:26-64) in fp32, written as bf16 to
 * out[(i/group)*ldo + (i%group)*dim + j] for i < n, timestep t[i % t_count] (device fp32). */
int mi355x_sd_timestep_embedding(const float* t, int t_count, int n, int dim, int group, int flip_sin_to_cos,
                                 float freq_shift, float scale, float max_period, void* out, int ldo, void* stream);
int mi355x_sd_silu(const void* x, void* y, int64_t n, int in_f32, int out_f32, void* stream);

/* conv_in (PPD/models/unet_2d_condition.py:1064): x NCHW fp32, optional device scalar in_scale (the scheduler's
 * scale_model_input), w [9*Cin][Cout] bf16, y NHWC bf16. */
int mi355x_sd_conv_in3x3(const float* x_nchw, const float* in_scale, const void* w, const float* bias, void* y,
                         int B, int Cin, int H, int W, int Cout, int ldy, void* stream);
/* out_f32 != 0: y is NHWC fp32 (start of the fp32 residual stream) */
int mi355x_sd_conv_in3x3_ex(const float* x_nchw, const float* in_scale, const void* w, const float* bias, void* y,
                            int B, int Cin, int H, int W, int Cout, int ldy, int out_f32, void* stream);
/* conv_out (unet_2d_condition.py:1196): x NHWC bf16, w [Cout<=4][3][3][Cin] bf16, y NCHW fp32. */
int mi355x_sd_conv_out3x3(const void* x, int ldx, const void* w, const float* bias, float* y_nchw,
                          int B, int Cin, int H, int W, int Cout, void* stream);
int mi355x_sd_copy_rows(const void* x, int ldx, void* y, int ldy, int64_t rows, int C, void* stream);

/* ControlNet residual inputs of UNet2DConditionModel.forward (down_block_additional_residuals /
 * mid_block_additional_residual, PPD/models/unet_2d_condition.py:1121-1132, 1151-1155): x[b*HW + p][c] += r[b][c][p] in place
 * on a bf16 NHWC row view (row stride ldx; C % 8 == 0), r NCHW fp32 as the reference passes it. */
int mi355x_sd_add_nchw(void* x, int ldx, const float* r_nchw, int B, int C, int64_t HW, void* stream);
int mi355x_sd_add_nchw_ex(void* x, int ldx, const float* r_nchw, int B, int C, int64_t HW, int x_f32, void* stream);

/* DiagonalGaussianDistribution (PPD/models/vae.py:744-763, built by AutoencoderKL.encode, autoencoder_kl.py:266-283) from
 * the encoder's moments held as fp32 rows [B*HW][ld >= 2L] (channels mean_0..L-1, logvar_0..L-1): writes NCHW fp32
 * mean ( = .mode()), logvar clipped to [-30, 20] and -- sample_nchw != NULL -- (mean + exp(0.5 logvar) * noise) * out_scale
 * ( = .sample() with the caller's noise, times the pipelines' vae.config.scaling_factor; noise NULL: mean * out_scale). */
int mi355x_sd_latent_dist(const float* moments, int ld, int B, int L, int64_t HW, const float* noise_nchw, float out_scale,
                          float* mean_nchw, float* logvar_nchw, float* sample_nchw, void* stream);

/* ---- CLIP text encoder (PPD/transformers/clip/modeling.py) ----
 * CLIPTextEmbeddings.forward (:214-231): out[i][:] = bf16(token_table[ids[i]] + position_table[i % seq_len]); tables bf16
 * [V][D] / [P][D], ids int32 in device memory (range-checked by the caller), D % 8 == 0. position_table may be NULL
 * (token embedding only: T5Stack.embed_tokens, PPD/transformers/t5/modeling.py:981-983). */
int mi355x_sd_embed_tokens(const int32_t* ids, int64_t n_tokens, int seq_len, const void* token_table,
                           const void* position_table, int D, void* out, int ldo, void* stream);
/* y = act(x) on n bf16 elements (n % 8 == 0): kind 0 quick_gelu (x sigmoid(1.702 x), CLIPMLP :338-350 with
 * hidden_act="quick_gelu"), 1 gelu (erf), 2 silu. The rest of CLIPEncoderLayer (:353-400) is mi355x_sd_layernorm,
 * mi355x_sd_linear (bias / residual epilogues) and mi355x_sd_sdpa with the causal mask as its additive bias. */
int mi355x_sd_activation(const void* x, void* y, int64_t n, int kind, void* stream);

/* ---- T5 encoder (the third text encoder of SD3; PPD/transformers/t5/modeling.py) ----
 * T5LayerNorm (:86-108): y = weight * x * rsqrt(mean(x^2) + eps) -- RMS norm, fp32 statistics, no bias; C <= 4096. */
int mi355x_sd_rmsnorm(const void* x, int rows, int C, int ldx, const float* weight, float eps, void* y, int ldy,
                      void* stream);
/* T5DenseGatedActDense (:164-167) on the fused [wi_0 | wi_1] projection x [rows, 2F]: y[r][j] = act(x[r][j]) * x[r][F+j];
 * kind 0 quick_gelu, 1 gelu (erf), 2 silu, 3 gelu_new (tanh approximation; T5 v1.1 "gated-gelu"). F % 8 == 0.
 * T5Attention (:308-424) is mi355x_sd_linear (no bias) + mi355x_sd_sdpa with scale 1.0 and the relative position bias
 * [1, heads, S, S] (compute_bias :293-306) as the additive mask. */
int mi355x_sd_gated_activation(const void* x, int ldx, void* y, int ldy, int64_t rows, int F, int kind, void* stream);

/* ---- AutoencoderKL decoder (PPD/models/autoencoder_kl.py:288-333, PPD/models/vae.py:182-343) ----
 * post_quant_conv (autoencoder_kl.py:121,292-293): 1x1 convolution of a small NCHW fp32 tensor, Cin, Cout <= 16,
 * y[b,co,p] = bias[co] + sum_ci w[co][ci] * bf16(x[b,ci,p] * in_scale); in_scale = 1 / scaling_factor of the calling
 * pipeline (pipeline_stable_diffusion.py:911). w bf16 [Cout][Cin]. */
int mi355x_sd_conv1x1_nchw(const float* x_nchw, float in_scale, const void* w, const float* bias, float* y_nchw,
                           int B, int Cin, int Cout, int64_t HW, void* stream);
/* y[r][0..n) = softmax(x[r][0..n)), x fp32 (scale already applied), y bf16; n % 4 == 0. The VAE mid-block attention
 * (one head of width C = 512: Attention(..., heads = C // C), unet_2d_blocks.py:606-619; get_attention_scores with
 * upcast_softmax, attention_processor.py:552-586) runs as linear(Q K^T, OUT_F32) -> softmax_rows -> linear(P V). */
int mi355x_sd_softmax_rows(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int n, void* stream);
/* out = coef[0]*x + coef[1]*y on fp32 latents, coef in device memory: the linear latent update every
 * epsilon-prediction scheduler step reduces to (PPD/schedulers/scheduling_euler_discrete.py:438-473,
 * scheduling_ddim.py:410-457 with eta = 0). */
int mi355x_sd_axpby(const float* x, const float* y, float* out, const float* coef, int64_t n, void* stream);
/* The same update with the classifier-free-guidance combine folded in (pipeline_stable_diffusion.py:882-891):
 * out = coef[0]*x + coef[1]*(eps_uncond + guidance_scale * (eps_text - eps_uncond)); eps_* = the two batch halves of the
 * UNet output. */
int mi355x_sd_cfg_axpby(const float* x, const float* eps_uncond, const float* eps_text, float* out, const float* coef,
                        float guidance_scale, int64_t n, void* stream);

/* hipGraph capture of a sequence of the calls above issued on `stream` (one denoising step). */
int mi355x_sd_graph_begin(void* stream);
int mi355x_sd_graph_end(void* stream, void** graph_exec);
int mi355x_sd_graph_launch(void* graph_exec, void* stream);
int mi355x_sd_graph_destroy(void* graph_exec);

/* ---- multi-GPU (paddlemix_amd/csrc/comm.hip): what a plain-C host needs for "a batch of independent prompts shards across the
 * GPUs of a node with an RCCL broadcast of the text-encoder / UNet weights over xGMI" (SURVEY.md 8e). One process per GPU. Rank 0 packs
 * the weights (mi355x_sd_unet_finalize_weights / mi355x_sd_program_bind write them into ONE caller-owned device buffer), every rank
 * allocates a buffer of the same size and receives it in place with mi355x_sd_comm_broadcast; each rank then denoises its own prompts
 * with no per-step collective (batch rows never interact), and mi355x_sd_comm_all_gather collects the ranks' latents once at the end.
 * The reference's precedent: the batch-parallel mode of its SD3 pipeline (PPD/pipelines/stable_diffusion_3/
 * pipeline_stable_diffusion_3.py:803-839). `id128`: 128 opaque bytes made by rank 0 (mi355x_sd_comm_unique_id) and carried to the
 * other ranks by the host's own means (file, socket, launcher environment). Collectives are stream-ordered on `stream`; byte counts,
 * no element type. RCCL is loaded with dlopen at the first call: a machine without it still loads the library, these five entry
 * points then return MI355X_SD_ERR_UNSUPPORTED with the reason in mi355x_sd_last_error(). */
int mi355x_sd_comm_unique_id(void* id128);
int mi355x_sd_comm_init(const void* id128, int rank, int world, void** comm);
int mi355x_sd_comm_broadcast(void* comm, void* buf, size_t bytes, int root, void* stream);
int mi355x_sd_comm_all_gather(void* comm, const void* send, void* recv, size_t bytes_per_rank, void* stream);
int mi355x_sd_comm_destroy(void* comm);

/* Test hook: dumps the lane->element maps of the MFMA / LDS-transpose instructions the kernels rely on
 * (out: 64*(4+16+4) floats, see tests/test_gpu_probe.py). */
int mi355x_sd_probe_layouts(float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
