/* muvo_hip.h — C ABI of libmuvo_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the MUVO
 * world-model training step.  Plain pointers and sizes only (no torch types); every tensor is
 * caller-allocated device memory, fp32 contiguous NCHW / NCDHW unless stated; `stream` is a hipStream_t
 * passed as void* (NULL = default stream).  The library keeps no global state besides the last error
 * string.  Every entry returns 0 on success, a negative MUVO_ERR_* otherwise; muvo_last_error() gives
 * the message.  The host side (muvo_amd/ops.py) binds these through ctypes and raises RuntimeError.
 *
 * Each group cites the reference op it replaces (paths relative to the reference repository).
 */
#ifndef MUVO_HIP_H
#define MUVO_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MUVO_OK 0
#define MUVO_ERR_INVALID_ARG (-1)
#define MUVO_ERR_HIP (-2)
/* (no collective error code: the library launches no collectives - gradient exchange is torch.distributed over RCCL, muvo_amd/parallel.py) */

#define MUVO_ACT_NONE 0
#define MUVO_ACT_RELU 1
#define MUVO_ACT_LEAKY 2 /* slope parameter */
#define MUVO_ACT_ELU 3   /* alpha = 1 */
#define MUVO_ACT_TANH 4
#define MUVO_ACT_SIGMOID 5

const char* muvo_last_error(void);
int muvo_abi_version(void);
/* one-launch self test of the MFMA operand/accumulator lane maps; returns 0 when C = A*B is exact */
int muvo_selftest_mfma(void* stream);
/* Deterministic mode (also MUVO_DETERMINISTIC=1 in the environment): reductions that otherwise depend on the arrival order of
 * float / double atomics (split-K partial sums, weight-gradient pixel ranges, bias / statistics / loss partial sums) run with one
 * contributor per address or are added in index order: two runs of the same step give bit-identical results.  A debugging
 * aid for parity work - slower (the affected kernels lose their split parallelism).  The reference has no counterpart
 * (torch.use_deterministic_algorithms plays this role for its cuDNN / ATen kernels). */
int muvo_set_deterministic(int on);
/* Clears the library-owned statistics accumulators of every stream (all-zero between operations by construction: the consuming
   kernel clears what it read).  For the caller's error path: an exception between a statistics pass and its consumer leaves
   partial sums behind that every later normalisation on that stream would add to (muvo_amd/ops.py: reset_accumulators). */
int muvo_reset_accumulators(void);
int muvo_get_deterministic(void);

/* ---- convolution family (conv_gemm.hip) -------------------------------------------------------
 * Replaces nn.Conv2d / nn.Conv3d / nn.ConvTranspose2d forward, data-gradient and weight-gradient:
 * muvo/models/common.py:549-632 (ConvDecoder), :161-202,498-546 (VoxelDecoder1), :102-130 (DecoderDS),
 * :249-367 (1x1 heads), muvo/layers/layers.py:9-66, timm ResNet-18 (muvo/models/mile.py:24,81).
 * Axes are (D,H,W); 2-D problems use D = 1, ksz[0] = 1, stride[0] = 1, pad[0] = 0, dil[0] = 1.
 * Weight layout is PyTorch's: Conv [Cout][Cin][kD][kH][kW], ConvTranspose [Cin][Cout][kD][kH][kW]. */
typedef struct muvo_conv_desc {
  int32_t nd;         /* 2 or 3 */
  int32_t transposed; /* 0 Conv, 1 ConvTranspose */
  int32_t N, Cin, Cout;
  int32_t in_sz[3], out_sz[3];
  int32_t ksz[3], stride[3], pad[3], dil[3];
} muvo_conv_desc;

/* Matrix-pipe arithmetic of the large contractions (process-wide; packed weights depend on it, so repack after a
 * change).  MUVO_CONV_F32: exact fp32-input MFMA (157 TFLOP/s peak).  MUVO_CONV_BF16X3: every fp32 operand is split
 * into bf16 hi + lo and each product formed as hi*hi + hi*lo + lo*hi on bf16 MFMA with fp32 accumulation
 * (3/16 of the fp32-MFMA cost, per-product relative error <= ~1e-5).  Initial value: env MUVO_CONV_MFMA=f32|bf16x3. */
#define MUVO_CONV_F32 0
#define MUVO_CONV_BF16X3 1
#define MUVO_CONV_MODE_DEFAULT MUVO_CONV_BF16X3
/* Products per fp32 product on the split-product implicit-GEMM kernels (MUVO_CONV_BF16X3 mode): 3 = hi*hi + hi*lo + lo*hi
 * (default, fp32-equivalent); 1 = hi*hi only, i.e. plain bf16 operands with fp32 accumulation - the arithmetic of the
 * reference's shipped PRECISION '16-mixed' (muvo/config.py:40) and of BASELINE.json configs[4] ("bf16").  An EXTENSION outside
 * the fp32 1e-3 parity contract: a third of the MFMA work; the voxel 3x3x3 kernels keep three products. */
int muvo_conv_set_products(int n);
int muvo_conv_get_products(void);
int muvo_conv_set_mode(int mode);
int muvo_conv_get_mode(void);
/* Which convolutions use the split-product kernels in MUVO_CONV_BF16X3 mode.  gflop_per_item >= 0: every phase with
 * at least this much work per batch item (GFLOP, 2*MAC); smaller ones stay on exact fp32 MFMA (env
 * MUVO_BF16X3_MIN_GFLOP sets the same threshold at start-up).  Negative (the initial state without the env variable):
 * the built-in policy fitted to per-layer timings on MI355X - an operation needs >= 0.1 GFLOP and >= 256 result pixels
 * per batch item and >= 16 reduction channels (weight gradients: >= 0.05 GFLOP per item and more than one tap). */
int muvo_conv_set_bf16x3_min_gflop(double gflop_per_item);
/* nn.Linear on token-major activations [rows][features] with thousands of rows (the transformer encoder:
 * muvo/models/mile.py:96-101, 558-561 - nn.TransformerEncoderLayer in_proj / out_proj / linear1 / linear2) on the bf16x3
 * implicit-GEMM kernels, as a 1x1 convolution over a one-row image of `rows` pixels.  Needs in_f % 8 == 0, out_f % 16 == 0,
 * both >= 64.  pack: w [out_f][in_f] -> the two packed operands (sizes from pack_floats); split: x -> bf16 hi/lo planes
 * (workspace_bytes(rows, features)); forward: y = act(x W^T + b) from the planes of x; dgrad: dx = dz W from the planes
 * of dz; wgrad: dw += dz^T x from both (scratch: out_f * in_f floats, all-zero on entry, left all-zero). */
int muvo_linear_bf16x3_pack_floats(int in_f, int out_f, int64_t* fwd_floats, int64_t* dgrad_floats);
int muvo_linear_bf16x3_pack(int in_f, int out_f, const float* w, float* wp_fwd, float* wp_dgrad, void* stream);
int64_t muvo_linear_bf16x3_workspace_bytes(int64_t rows, int features);
int muvo_linear_bf16x3_split(const float* x, int64_t rows, int features, void* ws, void* stream);
int muvo_linear_bf16x3_forward(int64_t rows, int in_f, int out_f, const void* ws_x, const float* wp_fwd, const float* bias,
                               float* y, int act, float slope, void* stream);
int muvo_linear_bf16x3_dgrad(int64_t rows, int in_f, int out_f, const void* ws_dz, const float* wp_dgrad, float* dx,
                             void* stream);
int muvo_linear_bf16x3_wgrad(int64_t rows, int in_f, int out_f, const void* ws_x, const void* ws_dz, float* scratch,
                             float* dw, void* stream);
/* Batched weight packing: all packed copies go stale with every optimizer step; instead of one or more small launches per
 * layer, list the layers once in a host array of muvo_pack_table_item_bytes()-sized entries (add appends the phases of a
 * layer and advances *n_items / *n_blocks; it returns 1 and appends nothing for shapes served by the voxel / head kernels,
 * which keep using muvo_conv_pack_weights), copy the array to the device, and call run after each optimizer step: one
 * launch refreshes every listed copy.  Source and destination pointers are captured at add time. */
int64_t muvo_pack_table_item_bytes(void);
int muvo_conv_pack_table_add(void* host_items, int capacity, int* n_items, int64_t* n_blocks, const muvo_conv_desc* d,
                             const float* w, float* wp_fwd, float* wp_dgrad);
int muvo_linear_bf16x3_pack_table_add(void* host_items, int capacity, int* n_items, int64_t* n_blocks, int in_f, int out_f,
                                      const float* w, float* wp_fwd, float* wp_dgrad);
int muvo_pack_table_run(const void* dev_items, int n_items, int64_t n_blocks, void* stream);
/* sizes (in floats) of the K-major packed weight buffers used by forward/wgrad and by dgrad */
int muvo_conv_pack_sizes(const muvo_conv_desc* d, int64_t* fwd_floats, int64_t* dgrad_floats);
/* repack w into wp_fwd and/or wp_dgrad (either may be NULL) */
int muvo_conv_pack_weights(const muvo_conv_desc* d, const float* w, float* wp_fwd, float* wp_dgrad, void* stream);
/* bytes of workspace that forward (op 0: ws), dgrad (op 1: ws) and wgrad (op 2: ws_x, op 3: ws_dy) need for this shape
 * in the current mode (0 = none; the bf16x3 kernels read channels-last bf16 hi/lo copies of their activation operands,
 * which the call writes there first).  The copy of x is the same for op 0 and op 2, the copy of dy for op 1 and op 3. */
int64_t muvo_conv_workspace_bytes(const muvo_conv_desc* d, int op);
/* kernel family that serves this shape in the current mode (for profiling/roofline attribution): 0 exact-fp32 implicit
 * GEMM, 1 bf16x3 implicit GEMM, 2 4x4x1-MFMA small-channel Conv3d, 3 float4 VALU heads, 4 bf16x3 small-channel Conv3d;
 * op 0 fwd, 1 dgrad, 2 wgrad */
int muvo_conv_kernel_family(const muvo_conv_desc* d, int op);
/* family 1 only: 1 = the launch uses the eight-wave ping-pong tiles (256x128 / 128x256), 0 = the four-wave small tiles */
int muvo_conv_kernel_variant(const muvo_conv_desc* d, int op);
/* y = act(conv(x, w) + bias); bias may be NULL; ws may be NULL when muvo_conv_workspace_bytes(d, 0) == 0 */
int muvo_conv_forward(const muvo_conv_desc* d, const float* x, const float* wp_fwd, const float* bias, float* y, int act,
                      float slope, void* ws, void* stream);
/* dx = conv_data_grad(dy, w) (dy already multiplied by act'(y) by the caller); ws_valid != 0: ws already holds the split
 * planes of dy (written by muvo_conv_prepare_dy), dy itself is then not read by the bf16x3 path */
int muvo_conv_dgrad(const muvo_conv_desc* d, const float* dy, const float* wp_dgrad, float* dx, void* ws, int ws_valid,
                    void* stream);
/* muvo_conv_forward that also accumulates, per (n, output channel), the sum and the sum of squares of the ACTIVATED output into
 * moments[N][Cout][2] (doubles, all-zero on entry): the statistics of the instance norm that follows the convolution
 * (ConvInstanceNorm3d, common.py:190-202), taken from the epilogue registers instead of a pass over the output tensor.  Only
 * the bf16x3 voxel kernels (muvo_conv_kernel_family(d, 0) == 4) do this: ask muvo_conv_forward_moments_supported first.
 * muvo_adain_fwd_moments consumes such a buffer (and leaves it all-zero). */
/* muvo_conv_forward + the 1x1 head on its output (ConvDecoder stage + RGBHead / LidarReHead, muvo/models/common.py:608-632,
 * 287-303): logits (N, CO, out spatial) = head_b + head_w (CO, Cout) . y per pixel, formed in the convolution's epilogue from
 * the activated values while they are stored - the head's forward makes no pass over y.  Ask _supported first (eight-wave
 * bf16x3 tiles, no split-K, Cout % 64 == 0, CO <= 4, CO * Cout <= 1024, y below 2 GB). */
int muvo_conv_forward_head_supported(const muvo_conv_desc* d, int CO);
int muvo_conv_forward_head(const muvo_conv_desc* d, const float* x, const float* wp_fwd, const float* bias, float* y, int act,
                           float slope, void* ws, const float* head_w, const float* head_b, int CO, float* logits, void* stream);
int muvo_conv_forward_moments_supported(const muvo_conv_desc* d);
int muvo_conv_forward_moments(const muvo_conv_desc* d, const float* x, const float* wp_fwd, const float* bias, float* y, int act,
                              float slope, double* moments, void* stream);
int muvo_adain_fwd_moments(const float* x, const float* style, float* y, float* save_mean, float* save_rstd, double* moments,
                           int N, int C, int64_t S, float eps, void* stream);
/* AdaIN folded into the consumer (muvo/models/common.py:190-202: conv -> AdaptiveInstanceNorm3d -> next conv inside a
 * DecoderBlock3d).  muvo_adain_affine: statistics from `moments` (as muvo_adain_fwd_moments; cleared) -> save_mean / save_rstd
 * (N*C, for muvo_adain_bwd) and aff[N][C][2] = (style_scale * rstd, style_shift - style_scale * rstd * mean); no pass over x.
 * muvo_conv_forward_affine / muvo_conv_wgrad_affine: the convolution / its weight gradient on scale * x + shift per (n, input
 * channel) with zero padding applied after the map (bf16x3 voxel kernels with 8 / 16 input channels:
 * muvo_conv_affine_supported); moments (may be NULL) as in muvo_conv_forward_moments; dw / dbias are accumulated. */
int muvo_adain_affine(const float* style, double* moments, float* save_mean, float* save_rstd, float* aff, int N, int C, int64_t S,
                      float eps, void* stream);
int muvo_conv_affine_supported(const muvo_conv_desc* d);
int muvo_conv_forward_affine(const muvo_conv_desc* d, const float* x, const float* aff, const float* wp_fwd, const float* bias,
                             float* y, int act, float slope, double* moments, void* stream);
int muvo_conv_wgrad_affine(const muvo_conv_desc* d, const float* x, const float* aff, const float* dy, float* dw, float* dbias,
                           void* stream);
/* Grouped Linear: L <= 16 layers y_l = x W_l^T + b_l on the same input x (M <= 24 rows, K features, K % 4 == 0) in one launch
 * per pass — the style projections of all AdaptiveInstanceNorm layers of a decoder (muvo/models/common.py:205-246,227-246:
 * `self.latent_affine(style)` per layer on the same latent).  W[l]: (N[l], K) row-major, b[l]: (N[l]) or NULL, Y[l]: (M, N[l]).
 * bwd: dx (M, K) is OVERWRITTEN with sum_l dY_l W_l (NULL: skipped); dW[l] / db[l] are accumulated (NULL entries / NULL
 * arrays: skipped).  The pointer arrays are host arrays. */
int muvo_grouped_linear_fwd(const float* x, int M, int K, int L, const float* const* W, const float* const* b, float* const* Y,
                            const int* N, void* stream);
int muvo_grouped_linear_bwd(const float* x, int M, int K, int L, const float* const* W, float* const* dY, float* dx,
                            float* const* dW, float* const* db, const int* N, void* stream);
/* Clock probe of the dominant kernel class (eight-wave bf16x3 implicit-GEMM tiles): shader clock (MHz) and wall time per
 * K step (32 deep, 24 MFMA 32x32x16 per wave) that workgroup 0 of the most recent launch measured over its K loop.  The
 * kernels run power-limited well below the 2.4 GHz the dense MFMA peak is quoted at; bench.py reports both. */
int muvo_bf3_loop_clock(double* shader_mhz, double* us_per_k_step);
/* Last stage of VoxelDecoder1 fused (common.py:541-545 + VoxelSemHead :354-367): AdaIN of the last convolution's output x
 * (N,C,S) followed by the 1x1x1 class head (head_w (CO,C), head_b (CO)) -> logits (N,CO,S).  The normalised tensor is never
 * written; backward recomputes it and forms dy = head_w^T dlogits on the fly.  `moments`: the (sum, sum of squares) buffer of
 * muvo_conv_forward_moments (cleared).  muvo_adain_head_bwd: dx (N,C,S) (incl. the LeakyReLU derivative of the producing
 * convolution, as muvo_adain_bwd), dstyle (N,2C) overwritten, dhead_w / dhead_b accumulated; ws: 2*N*C doubles. */
int muvo_adain_head_supported(int C, int CO, int64_t S);
int muvo_adain_head_fwd(const float* x, const float* style, float* save_mean, float* save_rstd, double* moments, const float* head_w,
                        const float* head_b, float* logits, int N, int C, int CO, int64_t S, float eps, void* stream);
int muvo_adain_head_bwd(const float* x, const float* style, const float* save_mean, const float* save_rstd, const float* head_w,
                        const float* dlogits, float* dx, float* dstyle, float* dhead_w, float* dhead_b, double* ws, int N, int C,
                        int CO, int64_t S, int act, float slope, void* stream);
/* dx += conv_data_grad(dy, w) for the 1x1 output heads (muvo_conv_kernel_family(d, 1) == 3; RGBHead / LidarReHead / VoxelSemHead,
 * common.py:274-303,354-367): the head hangs off a decoder trunk, dx already holds the gradient that came back through the
 * trunk, so no separate add pass over the feature map is needed */
int muvo_conv_dgrad_accumulate(const muvo_conv_desc* d, const float* dy, const float* wp_dgrad, float* dx, void* stream);
/* backward preamble when muvo_conv_kernel_family(d, 1) == muvo_conv_kernel_family(d, 2) == 1: ws_dy <- split planes of
 * dy * act'(y) (y may be NULL with MUVO_ACT_NONE), dbias += its per-channel sums (dbias may be NULL) */
int muvo_conv_prepare_dy(const muvo_conv_desc* d, const float* y, const float* dy, int act, float slope, void* ws_dy,
                         float* dbias, void* stream);
/* The same preamble for a layer whose output also feeds a 1x1 head with CO <= 4 produced channels (RGBHead / LidarReHead behind the
 * transposed convolutions of ConvDecoder, common.py:608-632): planes of (dy + W_head^T dhead) * act'(y).  dy: gradient from the
 * trunk, NULL at the last stage (its output feeds the head only); dhead (N, CO, S); head_w (CO, Cout).  The head's data gradient
 * (a pass over all Cout channels of the feature map) is never written.  dhead_w (CO, Cout) / dhead_b (CO), optional and only with
 * an activation (y is then read by the pass anyway): the head's weight / bias gradient sum_pixels dhead * y, accumulated. */
int muvo_conv_prepare_dy_head_supported(const muvo_conv_desc* d, int CO);
int muvo_conv_prepare_dy_head(const muvo_conv_desc* d, const float* y, const float* dy, const float* dhead, const float* head_w,
                              int CO, int act, float slope, void* ws_dy, float* dbias, float* dhead_w, float* dhead_b, void* stream);
/* The operand layout of the bf16x3 kernels on its own: x (N,C,S) fp32 -> ws = [bf16 hi plane | bf16 lo plane] channels-last
 * (N,S,roundup(C,8)) (+ 16 zero bytes + scratch); optionally multiplied by act'(y) first and with per-channel sums added to
 * dbias (the backward preamble above without a descriptor).  ws: muvo_split_planes_bytes(N, C, S) bytes. */
int64_t muvo_split_planes_bytes(int N, int C, int64_t S);
int muvo_split_planes(const float* x, void* ws, int N, int C, int64_t S, const float* y, int act, float slope, float* dbias,
                      void* stream);
/* dw += conv_weight_grad(x, dy) (PyTorch layout); dbias += sum(dy) if non-NULL.
 * dwp_scratch: fwd_floats floats of workspace that must be ALL-ZERO on entry and is left all-zero on return (the split-K
 * partial sums accumulate in it with float atomics; the unpack pass clears what it reads, so no memset is needed per call).  ws_x / ws_dy: see muvo_conv_workspace_bytes (may be NULL
 * when 0 bytes); flags bit 0 / bit 1: ws_x / ws_dy already hold the copies written by muvo_conv_forward(x) /
 * muvo_conv_dgrad(dy) for the same tensors, so wgrad does not rewrite them. */
int muvo_conv_wgrad(const muvo_conv_desc* d, const float* x, const float* dy, float* dwp_scratch, float* dw, float* dbias,
                    void* ws_x, void* ws_dy, int flags, void* stream);

/* db[m] += sum_{n,s} dy[n][m][s] (bias gradient of any NCHW-like tensor) */
int muvo_bias_grad_nchw(const float* dy, float* db, int N, int M, int64_t S, void* stream);

/* ---- strided batched GEMM (gemm.hip) ------------------------------------------------------------
 * Replaces nn.Linear fwd/bwd (mile.py:151-161, transition.py, common.py:53-68,205-246), the matmuls of
 * nn.MultiheadAttention (mile.py:96-101) and ConvTranspose2d on a 1x1 input (common.py:578-581).
 * C[b1][b2][m][n] = act(alpha * sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn] + bias[n / bias_div])
 * mode 1: C += alpha * A*B with float atomics (split-K; bias/act must be unset). */
typedef struct muvo_gemm_desc {
  int32_t M, N, K;
  int64_t sam, sak, sbk, sbn, scm;
  int32_t B1, B2;
  int64_t a_b1, a_b2, b_b1, b_b2, c_b1, c_b2;
  float alpha;
  int32_t bias_div;
  int32_t act;
  float slope;
  int32_t mode;
} muvo_gemm_desc;
int muvo_gemm(const muvo_gemm_desc* d, const float* A, const float* B, float* C, const float* bias, void* stream);

/* ---- normalisation (norm.hip) --------------------------------------------------------------------
 * Train-mode BatchNorm2d (batch statistics over N*S, running stats momentum update, unbiased running
 * var) fused with ReLU and a residual add: res_mode 0 none, 1 add before ReLU (BasicBlock,
 * layers.py:48-66), 2 add after ReLU (DecoderDS, common.py:128).  ws: 2*C doubles of workspace. */
int muvo_bn_train_fwd(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                      float* save_mean, float* save_rstd, float* running_mean, float* running_var, double* ws, int N, int C,
                      int64_t S, float eps, float momentum, int res_mode, int relu, void* stream);
/* mask_mode 0: no ReLU; 1: ReLU mask from y > 0; 2: ReLU mask from bn(x) > 0 (residual added after the ReLU).
 * dgamma/dbeta are accumulated (+=). dres (may be NULL) receives the masked gradient for the residual branch. */
int muvo_bn_train_bwd(const float* x, const float* y, const float* dy, const float* gamma, const float* beta,
                      const float* save_mean, const float* save_rstd, float* dx, float* dres, float* dgamma, float* dbeta,
                      double* ws, int N, int C, int64_t S, int mask_mode, void* stream);
/* AdaptiveInstanceNorm3d (common.py:227-246): y = style[:, :C] * (x-mean)/sqrt(var+eps) + style[:, C:]
 * x_batch_stride = 0 broadcasts one (C,S) tensor over the batch (VoxelDecoder1.constant_tensor). ws: 2*N*C doubles. */
int muvo_adain_fwd(const float* x, const float* style, float* y, float* save_mean, float* save_rstd, double* ws, int N,
                   int C, int64_t S, int64_t x_batch_stride, float eps, void* stream);
/* act/slope: when x is the output of an activation (conv + LeakyReLU in ConvInstanceNorm3d, common.py:190-202) the
 * returned dx is already multiplied by that activation's derivative (MUVO_ACT_NONE: plain AdaIN gradient) */
int muvo_adain_bwd(const float* x, const float* style, const float* dy, const float* save_mean, const float* save_rstd,
                   float* dx, float* dstyle, double* ws, int N, int C, int64_t S, int64_t x_batch_stride, int act,
                   float slope, void* stream);
/* post-LN transformer sub-layer tail: z = x + dropout(a); y = LayerNorm(z) (nn.TransformerEncoderLayer, mile.py:96-101) */
int muvo_add_dropout_layernorm_fwd(const float* x, const float* a, const float* gamma, const float* beta, float* y,
                                   float* z, float* mean, float* rstd, int rows, int E, float eps, float p, uint64_t seed,
                                   void* stream);
int muvo_add_dropout_layernorm_bwd(const float* dy, const float* z, const float* mean, const float* rstd,
                                   const float* gamma, float* dx, float* da, float* dgamma, float* dbeta, int rows, int E,
                                   float p, uint64_t seed, void* stream);

/* ---- elementwise / layout / pooling / resampling (elementwise.hip) ------------------------------- */
int muvo_act_fwd(const float* x, float* y, int64_t n, int act, float slope, void* stream);
int muvo_act_bwd(const float* y, const float* dy, float* dx, int64_t n, int act, float slope, void* stream);
int muvo_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, void* stream);
int muvo_axpby(const float* a, const float* b, float* out, int64_t n, float alpha, float beta, void* stream);
int muvo_copy2d(const float* src, float* dst, int64_t rows, int64_t cols, int64_t ld_src, int64_t ld_dst, int accumulate,
                void* stream);
int muvo_colsum_acc(const float* x, float* out, int64_t rows, int64_t cols, int64_t ld, void* stream);
int muvo_batchsum(const float* x, float* out, int N, int64_t inner, int accumulate, void* stream);
/* tokens[(l0+l)][n][c] = x[n][c][l] + pos[c][l] + temb[c*temb_stride]  (mile.py:542-557) and its inverse (:560-561) */
int muvo_nchw_to_tokens(const float* x, const float* pos, const float* temb, int temb_stride, float* tokens, int N, int C,
                        int L, int l0, void* stream);
int muvo_tokens_to_nchw(const float* tokens, float* x, int N, int C, int L, int l0, void* stream);
/* F.max_pool2d 3x3 s2 p1 (ResNet stem) / 2x2 s2 (DecoderDS, common.py:127-128); idx: one byte per output = position a*k + b of
 * the (first) maximum inside its window, consumed by the backward pass */
int muvo_maxpool2d_fwd(const float* x, float* y, uint8_t* idx, int64_t NC, int H, int W, int OH, int OW, int k, int s, int p,
                       void* stream);
int muvo_maxpool2d_bwd(const float* dy, const uint8_t* idx, float* dx, int64_t NC, int H, int W, int OH, int OW, int k, int s,
                       int p, void* stream);
int muvo_avgpool_fwd(const float* x, float* y, int64_t G, int64_t S, void* stream);
int muvo_avgpool_bwd(const float* dy, float* dx, int64_t G, int64_t S, void* stream);
/* F.interpolate(scale_factor=2, mode='trilinear', align_corners=False) (common.py:169) */
int muvo_upsample3d_x2_fwd(const float* x, float* y, int64_t NC, int D, int H, int W, void* stream);
int muvo_upsample3d_x2_bwd(const float* dy, float* dx, int64_t NC, int D, int H, int W, void* stream);
/* PreProcess (muvo/models/preprocess.py:102-225): u8 -> /255 -> crop -> (label, ImageNet-normalised) */
int muvo_preprocess_image(const uint8_t* img, float* label, float* norm, int64_t NC, int C, int H, int W, int top, int left,
                          int CH, int CW, const float* mean3, const float* std3, void* stream);
int muvo_preprocess_route(const uint8_t* img, float* norm, int64_t NC, int C, int H, int W, int OH, int OW,
                          const float* mean3, const float* std3, void* stream);
/* Fused multi-head self-attention core of nn.TransformerEncoderLayer (muvo/models/mile.py:96-101,558), csrc/attention.hip:
 * qkv (L, N, 3*H*DH) packed in-projection (q | k | v along the last axis, heads contiguous inside each), out (L, N, H*DH) =
 * dropout(softmax(q k^T / sqrt(DH))) v per (n, head), lse (N*H, L) row log-sum-exp saved for the backward pass; K and V of a
 * head live in LDS, the L x L matrices never reach HBM.  p / seed: attention-probability dropout, mask index
 * ((n*H + h)*L + query)*L + key (same as muvo_softmax_dropout_*).  muvo_attention_supported: DH in {16,32,48,64}, L <= 384
 * and 2 * roundup(L,16) * (DH+4) * 4 bytes (+ 8 * roundup(L,16)) <= 160 KB of LDS; otherwise use the unfused entry points.
 * muvo_attention_bwd writes all of dqkv (dq, dk, dv) — no atomics, deterministic. */
int muvo_attention_supported(int L, int DH);
int muvo_attention_fwd(const float* qkv, float* out, float* lse, int L, int N, int H, int DH, float p, uint64_t seed,
                       void* stream);
int muvo_attention_bwd(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, int L, int N, int H,
                       int DH, float p, uint64_t seed, void* stream);
/* The whole recurrent state-space model (muvo/models/transition.py:76-173) as two persistent kernels (csrc/rssm.hip): one
 * workgroup per CU walks the T time steps, 4 grid barriers per step; T <= 64, H/S/E/A multiples of 4; B <= 64 sequences, run as
 * consecutive launches over slabs of 4 (sequences are independent).  The grid is clamped to what the occupancy calculator says
 * is co-resident; a barrier spin that still times out (2 s) raises the sticky word barrier_word[1] and ends the kernel.
 * weights[18] (PyTorch [out][in] layouts): pre_gru_net.0.{weight,bias}, recurrent_model.{weight_ih,weight_hh,bias_ih,bias_hh},
 *   prior_action_module.0.{weight,bias}, posterior_action_module.0.{weight,bias}, prior.module.0.{weight,bias},
 *   prior.module.2.{weight,bias}, posterior.module.0.{weight,bias}, posterior.module.2.{weight,bias}.
 * emb (B,T,E), act (B,T,AD), noise (B,T,2,S) [prior, posterior draws]; use_prior_mask bit t: step t+1 continues from the PRIOR
 *   sample of step t (transition.py:118-124), else from the posterior sample.
 * out7 (B,T,.): hidden state h (H) — shared by the prior and posterior dicts —, prior mu, sigma, sample, posterior mu, sigma,
 *   sample (S).  keep12 (B,T,.), written for the backward pass: h_prev (H), z_prev (S), a_prev (AD), u (H), gi (3H), gh (3H),
 *   x_prior (H+A), x_post (H+E+A), y1_prior (H+A), y1_post (H+E+A), mls_prior (2S), mls_post (2S).
 * muvo_rssm_backward: kept5 = {h_prev, gi, gh, mls_prior, mls_post}; upstream7 = gradients of out7 (NULL = none);
 *   grads10 (B,T,.) written: d_emb (E), dmls_prior, dmls_post (2S), dy1_prior (H+A), dy1_post (H+E+A), dgi, dgh (3H), du (H),
 *   dla_prior, dla_post (A) — the weight gradients are dW = dY^T X over the B*T rows (caller: one skinny GEMM per weight):
 *   W_post2: dmls_post x y1_post; W_post0: dy1_post x x_post; W_prior2: dmls_prior x y1_prior; W_prior0: dy1_prior x x_prior;
 *   W_ih: dgi x u; W_hh: dgh x h_prev; W_pre: du x z_prev; action modules: dla x a_prev; biases: column sums of the dY.
 *   wt_scratch: muvo_rssm_transposed_floats() floats (the transposed weights are rebuilt by every call); scratch:
 *   muvo_rssm_scratch_floats() floats; barrier_word: 8 bytes of device memory, zero before the first call: [0] the
 *   barrier counter (reset by every launch), [1] the sticky time-out flag (the caller polls it: ops.rssm_check). */
int muvo_rssm_supported(int B, int T, int H, int S, int E, int A, int AD);
int64_t muvo_rssm_transposed_floats(int H, int S, int E, int A);
int64_t muvo_rssm_scratch_floats(int B, int H, int S, int E, int A);
int muvo_rssm_forward(int B, int T, int H, int S, int E, int A, int AD, const float* const* weights, const float* emb,
                      const float* act, const float* noise, uint64_t use_prior_mask, float* const* out7, float* const* keep12,
                      uint32_t* barrier_word, float min_std, void* stream);
int muvo_rssm_backward(int B, int T, int H, int S, int E, int A, int AD, const float* const* weights, float* wt_scratch,
                       const float* noise, uint64_t use_prior_mask, const float* const* kept5, const float* const* upstream7,
                       float* const* grads10, float* scratch, uint32_t* barrier_word, void* stream);
/* Training-time augmentation inside PreProcess.forward (muvo/models/preprocess.py:45-48,213-214; PixelAugmentation :295-333,
 * RouteAugmentation :336-367; torchvision 0.15.2 tensor algorithms).  The random draws are explicit inputs.
 * muvo_pixel_augment: img (F,3,H,W) in [0,1] (the cropped image = rgb_label_1) is augmented IN PLACE, norm (F,3,H,W) receives
 *   the ImageNet-normalised result; frames whose row says "nothing" are left untouched in both.  params: F rows of
 *   MUVO_PIXAUG_STRIDE floats: [0] 0 none / 1 gaussian blur 5x5 / 2 sharpen, [1] sigma | sharpness factor, [2] colour jitter
 *   on (0/1), [3..6] order of the colour ops (0 brightness, 1 contrast, 2 saturation, 3 hue), [7..10] their factors.
 *   tmp: F*3*H*W floats, gray_sum: F doubles (scratch).
 * muvo_preprocess_route_aug: muvo_preprocess_route + RouteAugmentation of sample b (all S frames alike); params (may be
 *   NULL): B rows of MUVO_ROUTEAUG_STRIDE floats: [0] 0 none / 1 drop / 2 end of route / 3 affine, [1] rows zeroed (mode 2),
 *   [2..7] torchvision's inverse affine matrix (_get_inverse_affine_matrix, center (0,0)). */
#define MUVO_PIXAUG_STRIDE 16
#define MUVO_ROUTEAUG_STRIDE 8
int muvo_pixel_augment(float* img, float* norm, float* tmp, const float* params, double* gray_sum, int64_t frames, int H, int W,
                       const float* mean3, const float* std3, void* stream);
int muvo_preprocess_route_aug(const uint8_t* img, float* out, const float* params, int B, int S, int C, int H, int W, int OH,
                              int OW, const float* mean3, const float* std3, void* stream);
int muvo_divide_scalar(const float* x, float* y, int64_t n, float divisor, void* stream);
int muvo_resize_bilinear(const float* x, float* y, int64_t NC, int H, int W, int OH, int OW, void* stream);
int muvo_resize_nearest_f32(const float* x, float* y, int64_t NC, int D, int H, int W, int OD, int OH, int OW, void* stream);
int muvo_resize_nearest_u8(const uint8_t* x, uint8_t* y, int64_t NC, int D, int H, int W, int OD, int OH, int OW,
                           void* stream);
/* attention probabilities: P = softmax(S), Pd = dropout(P) (functional MHA dropout) */
int muvo_softmax_dropout_fwd(const float* S, float* P, float* Pd, int64_t rows, int cols, float p, uint64_t seed, void* stream);
int muvo_softmax_dropout_bwd(const float* P, const float* dPd, float* dS, int64_t rows, int cols, float p, uint64_t seed,
                             void* stream);
/* nn.GRUCell pointwise part and RepresentationModel sampling (muvo/models/transition.py:18-24,49-52,176-181) */
int muvo_gru_fwd(const float* gi, const float* gh, const float* h, float* hnew, int B, int H, void* stream);
int muvo_gru_bwd(const float* gi, const float* gh, const float* h, const float* dhnew, float* dgi, float* dgh, float* dh,
                 int B, int H, void* stream);
int muvo_rssm_sample_fwd(const float* mu_logsigma, const float* eps, int64_t eps_ld, float* mu, float* sigma, float* sample,
                         int B, int S, float min_std, void* stream);
int muvo_rssm_sample_bwd(const float* mu_logsigma, const float* eps, int64_t eps_ld, const float* dmu, const float* dsigma,
                         const float* dsample, float* dmls, int B, int S, void* stream);

/* ---- losses + optimiser (losses.hip): muvo/losses.py:53-287, muvo/trainer.py:251-390,1022-1060 ------ */
/* SpatialRegressionLoss over channels [c0,c1) of (F,Ct,HW) tensors; loss = weight * masked mean; stats2: 2 doubles */
int muvo_spatial_loss_fwd(const float* pred, const float* target, int64_t F, int Ct, int64_t HW, int c0, int c1, int norm,
                          float ignore, float weight, double* stats2, float* loss, void* stream);
int muvo_spatial_loss_bwd(const float* pred, const float* target, float* dpred, int64_t F, int Ct, int64_t HW, int c0, int c1,
                          int norm, float ignore, float weight, const double* stats2, const float* gout, void* stream);
/* the same with an explicit pixel mask (F, HW) uint8 instead of `target[:, c0] != ignore` (instance_mask argument of
 * SpatialRegressionLoss.forward, muvo/losses.py:87-90: LOSSES.RGB_INSTANCE, muvo/trainer.py:303-321); mask NULL = the plain form */
int muvo_spatial_loss_masked_fwd(const float* pred, const float* target, const uint8_t* mask, int64_t F, int Ct, int64_t HW, int c0,
                                 int c1, int norm, float ignore, float weight, double* stats2, float* loss, void* stream);
int muvo_spatial_loss_masked_bwd(const float* pred, const float* target, const uint8_t* mask, float* dpred, int64_t F, int Ct,
                                 int64_t HW, int c0, int c1, int norm, float ignore, float weight, const double* stats2,
                                 const float* gout, void* stream);
/* VoxelLoss (CE mean) + SemScalLoss + GeoScalLoss in one pass; loss3 = {ce, sem_scal, geo_scal} * weight */
int muvo_voxel_loss_stats_doubles(int C);
int muvo_voxel_loss_coef_floats(int C);
int muvo_voxel_loss_fwd(const float* logits, const uint8_t* target, int64_t F, int C, int64_t V, const float* class_w,
                        float weight, double* stats, float* coef, float* loss3, void* stream);
int muvo_voxel_loss_bwd(const float* logits, const uint8_t* target, float* dlogits, int64_t F, int C, int64_t V,
                        const float* class_w, float weight, const float* coef, const float* gout3, void* stream);
/* per-pixel class-weighted cross entropy of SegmentationLoss (muvo/losses.py:22-37, F.cross_entropy(reduction='none')):
 * logits (N,C,HW), target (N,HW) bytes, class_w C floats or NULL, loss / gloss (N,HW); the top-k / mean reduction stays with
 * the caller (losses.py:44-50) */
int muvo_seg_ce_fwd(const float* logits, const uint8_t* target, const float* class_w, float* loss, int64_t N, int C, int64_t HW,
                    void* stream);
int muvo_seg_ce_bwd(const float* logits, const uint8_t* target, const float* class_w, const float* gloss, float* dlogits, int64_t N,
                    int C, int64_t HW, void* stream);
int muvo_l1_rows_fwd(const float* p, const float* t, int64_t rows, int cols, float weight, float* loss, void* stream);
int muvo_l1_rows_bwd(const float* p, const float* t, float* dp, int64_t rows, int cols, float weight, const float* gout,
                     void* stream);
int muvo_kl_loss_fwd(const float* prior_mu, const float* prior_sigma, const float* post_mu, const float* post_sigma, int B,
                     int T, int S, float weight, float* loss, void* stream);
int muvo_kl_loss_bwd(const float* prior_mu, const float* prior_sigma, const float* post_mu, const float* post_sigma,
                     float* d_prior_mu, float* d_prior_sigma, float* d_post_mu, float* d_post_sigma, int B, int T, int S,
                     float weight, float alpha, const float* gout, void* stream);
/* torch.optim.AdamW step over a flat parameter segment; grad_scale multiplies g (1/world_size after a sum all-reduce) */
int muvo_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int step, float grad_scale, void* stream);

/* ---- evaluation metrics of the validation path (muvo/metrics.py via muvo/trainer.py:426-490) ----------------------------
 * All accumulators are caller-zeroed device buffers the kernels ADD into (several batches may share them).
 * muvo_ssim_frames: sums[n] += sum over (c, y, x) of the SSIM map of frame n (valid 11x11 Gaussian window given as 121
 *   floats, losses.py:304-314); mean = sums / (C (H-10) (W-10)).  Replaces SSIMLoss._ssim, losses.py:316-339.
 * muvo_sqdiff_frames: sums[n] += sum (pred - target)^2 over the L elements of frame n (PSNRMetric.psnr, metrics.py:305-309).
 * muvo_chamfer_sums: sums[2n] += sum_i min_j |a_i - b_j|, sums[2n+1] += sum_j min_i |a_i - b_j| for point sets a (N,P,3),
 *   b (N,Q,3) (CDMetric.add_batch, metrics.py:243-249).
 * muvo_ssc_counts: prediction = argmax over the C logits (N,C,V); counts[0..2] += completion tp/fp/fn, counts[3+3j..] +=
 *   tp/fp/fn of class j over voxels with label != 255 (trainer.py:482-490, SSCMetrics.add_batch, metrics.py:77-100). */
int muvo_ssim_frames(const float* pred, const float* target, const float* window, double* sums, int N, int C, int H, int W,
                     float c1, float c2, void* stream);
/* SSIM as a training loss (LOSSES.SSIM: trainer.py:312-318, SSIMLoss losses.py:292-348): muvo_ssim_maps = muvo_ssim_frames plus
 * the derivative maps dA, dB, dC (N*C*(H-10)*(W-10) floats each) of every SSIM value w.r.t. the prediction's window mean,
 * E[p^2] and E[p t]; muvo_ssim_bwd: dpred = scale * adjoint window pass (scale = upstream gradient / number of SSIM values) */
int muvo_ssim_maps(const float* pred, const float* target, const float* window, double* sums, float* dA, float* dB, float* dC, int N,
                   int C, int H, int W, float c1, float c2, void* stream);
int muvo_ssim_bwd(const float* pred, const float* target, const float* window, const float* dA, const float* dB, const float* dC,
                  float* dpred, int N, int C, int H, int W, float scale, void* stream);
int muvo_sqdiff_frames(const float* pred, const float* target, double* sums, int N, int64_t L, void* stream);
int muvo_chamfer_sums(const float* a, const float* b, double* sums, int N, int P, int Q, void* stream);
int muvo_ssc_counts(const float* logits, const uint8_t* label, uint64_t* counts, int64_t F, int C, int64_t V, void* stream);

/* ---- BEV lifting: FrustumPooling (muvo/models/frustum_pooling.py:67-217) as called from Mile.encode (mile.py:506-522) ----
 * muvo_frustum_cells: cells[b][d][h][w] = (iz*ny + iy)*nx + ix of the frustum point, -1 outside the grid (get_geometry
 *   :108-128 + the index part of voxel_pooling :139-158).  combine = R K^-1 per frame (3x3 row major), trans = camera position,
 *   xs/ys/ds = pixel columns, rows and depth bin centres of the frustum (:92-106); sx, ox, sy, oy = BEV intrinsics
 *   (geometry_utils.py:8-19), bz/dz = first cell centre and cell size along "up" (gen_dx_bx :10-21).
 * muvo_frustum_pool_fwd: out[b][c*nz + iz][iy][ix] = sum over the lifted points of the cell of depth[b][d][p] * feat[b][c][p]
 *   (outer product mile.py:519 + voxel_pooling :130-182).  mask: (B,D,HW) bytes, non-zero = lift (NULL = all, mile.py:511-518).
 *   acc: B*ncell*C floats of scratch (channels-last accumulator).  Valid for nz = 1 or any nz (out is (B, C, ncell) with the
 *   up index inside the cell number; the caller reorders for nz > 1).
 * muvo_frustum_pool_bwd: gradients of the above w.r.t. feat and depth (QuickCumsum.backward :56-64 chained through the outer
 *   product); g_cl: B*ncell*C floats of scratch.  Deterministic (no atomics).
 * muvo_depth_expectation: e[b][p] = sum_d ds[d] depth[b][d][p] (get_depth_map :211-214). */
int muvo_frustum_cells(const float* combine, const float* trans, const float* xs, const float* ys, const float* ds, int32_t* cells,
                       int B, int D, int H, int W, float sx, float ox, float sy, float oy, float bz, float dz, int nx, int ny,
                       int nz, void* stream);
int muvo_frustum_pool_fwd(const float* feat, const float* depth, const uint8_t* mask, const int32_t* cells, float* acc, float* out,
                          int B, int C, int D, int64_t HW, int ncell, void* stream);
int muvo_frustum_pool_bwd(const float* feat, const float* depth, const uint8_t* mask, const int32_t* cells, const float* gout,
                          float* g_cl, float* dfeat, float* ddepth, int B, int C, int D, int64_t HW, int ncell, void* stream);
int muvo_depth_expectation(const float* depth, const float* ds, float* e, int B, int D, int64_t HW, void* stream);
/* adjoint of muvo_resize_bilinear (backward of F.interpolate(..., 'bilinear', align_corners=False) in `Decoder`, common.py:96) */
int muvo_resize_bilinear_bwd(const float* dy, float* dx, int64_t NC, int H, int W, int OH, int OW, void* stream);
/* softmax over the channel dimension of (B, C, HW): depth distribution of the mono depth head (mile.py:509) */
int muvo_softmax_channel_fwd(const float* x, float* y, int B, int C, int64_t HW, void* stream);
int muvo_softmax_channel_bwd(const float* y, const float* dy, float* dx, int B, int C, int64_t HW, void* stream);

/* ---- input pipeline on the device: per-frame work of CarlaDataset.load_single_element_time_t (muvo/data/dataset.py:275-327) ----
 * muvo_range_projection: raw lidar sweep (P,3) float32 in the sensor frame + CARLA object tags -> range_view_pcd_xyzd (4,H,W)
 *   float32 (x, y, z in the ego frame, depth; empty pixels 0,0,0,-1) and optionally the label image (H,W): convert_coor_lidar
 *   (data/data_preprocessing.py:119-122), label remap and ego-box masking (dataset.py:281-290), PointCloud.do_range_projection
 *   (muvo/utils/geometry_utils.py:176-213; float64 geometry, the closest point of a pixel wins, exact ties: lowest index).
 *   remap: 256-entry table; ego_dim: EGO_VEHICLE_DIMENSION (constants.py:8); scratch: 12*H*W bytes.
 * muvo_voxel_grid: sparse voxel rows (Q,4) int64 (x, y, z, tag) -> dense uint8 grid (dataset.py:316-327: tag 255 -> 0, remap,
 *   later rows win); scratch: 4*X*Y*Z bytes. */
int muvo_range_projection(const float* points_xyz, const uint8_t* obj_tag, const uint8_t* remap, int64_t P, const double* lidar_pos,
                          const double* ego_dim, double fov_down_deg, double fov_up_deg, int H, int W, void* scratch,
                          float* xyzd, uint8_t* seg, void* stream);
int muvo_voxel_grid(const int64_t* rows, int64_t Q, const uint8_t* remap, int X, int Y, int Z, uint32_t* scratch, uint8_t* voxels,
                    void* stream);
/* convert_instance_mask_to_center_and_offset_label (muvo/utils/instance_utils.py:4-35, called from
 * PreProcess.prepare_bev_labels, preprocess.py:68-100): instance ids (F,H,W) uint8 (0 = background) -> centre heat map
 * center (F,H,W) = max over the instances of the frame of exp(-d^2 / sigma^2) around the rounded centroid, and offset (F,2,H,W)
 * = centroid - pixel (rows, then columns) on instance pixels, `ignore` elsewhere.  scratch: F*256*3 doubles. */
int muvo_instance_labels(const uint8_t* instance, int64_t F, int H, int W, float sigma, float ignore, double* scratch, float* center,
                         float* offset, void* stream);

/* ---- measurement aid for the data-parallel path on ONE GPU (muvo_amd/parallel.py: SegmentedGradReducer(fake_peers=G)) ----
 * Lightning's implicit DDP (train.py:93-98) all-reduces the gradients; on a one-GPU box a one-rank RCCL all-reduce is a no-op, so
 * nothing ever ran beside the big-LDS convolution tiles and the persistent recurrent kernels.  muvo_fake_allreduce stands in for
 * the collective's LOCAL footprint: `workgroups` resident workgroups of 256 threads (RCCL's channels) walk buf (n floats, n % 4 == 0,
 * 16-byte aligned), read every element twice (own copy + "received" copy: two uncached loads), write 0.5 * (a + b) - bit-identical
 * to the input, so parity is unchanged - and sleep `sleep` x 64 clocks per 16-KB round, which sets the rate (calibrated by the
 * caller to the xGMI ring's ~1 ms per 100 MB).  The loop has a fixed trip count: every wave terminates. */
int muvo_fake_allreduce(float* buf, int64_t n, int workgroups, int sleep, void* stream);

/* antialiased linear resize of (NC, H, W) float planes: EVAL.RESOLUTION of PreProcess.forward (muvo/models/preprocess.py:209-210,
 * functional_resize_batch :252-273: torchvision 0.15.2 `resize(image, size, antialias=True)`, i.e. ATen's _upsample_bilinear2d_aa -
 * per axis the triangle filter of width in/out around scale * (i + 0.5), weights normalised, horizontal pass first).  ynorm (may be
 * NULL): (y - mean[c]) / std[c] with c = plane index % C (the ImageNet normalisation that follows, preprocess.py:217; host arrays). */
int muvo_resize_bilinear_aa(const float* x, float* y, float* ynorm, const float* mean, const float* std, int64_t NC, int C, int H,
                            int W, int OH, int OW, void* stream);

/* ---- BatchNorm that writes its consumer's operand format (round 4; muvo/layers/layers.py:9-66, muvo/models/common.py:102-130, timm
 * ResNet-18: conv -> BatchNorm2d (train mode) -> ReLU -> conv) ----
 * muvo_bn_train_fwd_planes = muvo_bn_train_fwd that ALSO stores the channels-last bf16 hi / lo planes of its result into `planes`
 * (muvo_split_planes_bytes(N, C, S) bytes: what the bf16x3 convolution kernels read, otherwise made by a separate split pass over
 * y); y may be NULL when the planes are the only thing the consumers need.  muvo_bn_train_bwd_planes = muvo_bn_train_bwd whose dx
 * goes out as planes (dx may be NULL): the operand of the data- and weight-gradient kernels of the convolution that produced x
 * (muvo_conv_dgrad with ws_valid, muvo_conv_wgrad with flag 2) - muvo_conv_prepare_dy is not needed.  Same arithmetic per element
 * as the unfused kernels (bit-identical results).  muvo_conv_forward_planes: muvo_conv_forward with ws_valid (x may be NULL then). */
int muvo_bn_train_fwd_planes(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                             float* save_mean, float* save_rstd, float* running_mean, float* running_var, int N, int C,
                             int64_t S, float eps, float momentum, int res_mode, int relu, void* planes, void* stream);
int muvo_bn_train_bwd_planes(const float* x, const float* y, const float* dy, const float* gamma, const float* beta,
                             const float* save_mean, const float* save_rstd, float* dx, float* dres, float* dgamma,
                             float* dbeta, int N, int C, int64_t S, int mask_mode, void* planes, void* stream);
int muvo_conv_forward_planes(const muvo_conv_desc* d, const float* x, const float* wp_fwd, const float* bias, float* y, int act,
                             float slope, void* ws, int ws_valid, void* stream);

/* ---- ResNet-18 stem (timm resnet18 conv1: Conv2d(3 | 4, 64, 7, stride 2, padding 3, bias=False); muvo/models/mile.py:24,81) as a direct
 * convolution on bf16x3 split products (csrc/conv_stem.hip): the input patch of an output tile in LDS, operands built from it without
 * a gather.  w / dw: the parameter and its gradient in PyTorch's layout (64, Cin, 7, 7) - no packed copy; dw is accumulated into
 * (float atomics: not offered in the deterministic mode).  bias may be NULL; relu != 0 applies max(0, .) in the epilogue.
 * muvo_stem_conv_supported: 1 when the descriptor is such a stem (even input height / width) and MUVO_STEM_KERNEL != 0. */
int muvo_stem_conv_supported(const muvo_conv_desc* d);
int muvo_stem_conv_forward(const muvo_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int relu, void* stream);
int muvo_stem_conv_wgrad(const muvo_conv_desc* d, const float* x, const float* dy, float* dw, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MUVO_HIP_H */
