/*
 * pixelpick_hip.h — C ABI of libpixelpick_hip.so (MI355X / gfx950 native hot path of PixelPick).
 *
 * The reference (NoelShin/PixelPick) has no FFI layer: its hot path is reached through the Python
 * call surface of query.py and networks/.  This header is the boundary a drop-in replacement binds
 * instead (ctypes stub in INTEGRATION.md).  Every entry point cites the reference lines it replaces
 * (paths relative to the reference repo root).
 *
 * Conventions
 *   - plain pointers + sizes; no torch types.  All pointers are DEVICE pointers unless stated.
 *   - the caller owns every buffer including the workspace; the library never allocates, frees or
 *     synchronises the device (all work is enqueued on `stream`, graph-capturable).
 *   - return 0 = ok, negative = error (PP_ERR_*); message via pp_last_error() (thread local).
 *   - outputs are fully overwritten.
 */
#ifndef PIXELPICK_HIP_H
#define PIXELPICK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pp_stream_t; /* hipStream_t */

enum {
    PP_OK = 0,
    PP_ERR_BAD_ARG = -1,      /* null pointer, bad shape/stride, unknown enum */
    PP_ERR_BAD_K = -2,        /* k < 1 or k > H*W */
    PP_ERR_WORKSPACE = -3,    /* workspace too small */
    PP_ERR_UNSUPPORTED = -4,  /* shape outside the supported range (e.g. an image of 2^31 pixels or more) */
    PP_ERR_LAUNCH = -5        /* hipLaunch failure (message holds hipGetErrorString) */
};

/* query.py:229-239 UncertaintySampler strategies.  (`random`, query.py:242-244, is host RNG.) */
enum { PP_ACQ_ENTROPY = 0, PP_ACQ_LEAST_CONFIDENCE = 1, PP_ACQ_MARGIN = 2 };
/* OR-ed into `strategy` at any acquisition entry point: score in the reference's OPERATION ORDER - p_c = exp(x_c - m) / S, then
 * sum(-p_c log p_c) with libm-accurate exp / log, each product and sum rounded separately (query.py:190,230) - instead of the default
 * algebraic form (entropy = log S + sum e_c (m - x_c) / S, v_exp_f32 exponentials; identical NaN behaviour).  Both forms are held to the
 * oracle's picks at full size; the reference-order form runs at about half the rate (0.40 of HBM).  Per call, no process state. */
#define PP_ACQ_REFERENCE_ORDER 0x100

/* Class counts up to this are scored from registers (one read of the logits); wider heads take the streamed scorers (acq_stream_kernel:
 * the class vector is read two or three times, the later passes from L2) - the reference softmaxes whatever width the model emits
 * (query.py:190), so no class count is rejected. */
#define PP_ACQ_MAX_CLASSES 64

int pp_version(void);
const char* pp_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Acquisition: query.py:190-204 + query.py:33-69 for a batch of B images in ONE pass over logits.
 *
 *   logits   f32, element (b,c,h,w) at logits[b*sB + c*sC + h*sH + w*sW] (NCHW or NHWC strides);
 *            replaces F.softmax(model(x)["pred"][:, :, :h, :w], dim=1)               query.py:190
 *   strategy PP_ACQ_*; score = UncertaintySampler(prob)                               query.py:192,229-239
 *   exclude  u8 [B,H,W] or NULL; non-zero = already queried or void; such pixels get score
 *            0.0 (entropy, least-confidence) or 1.0 (margin)                          query.py:195-201
 *   k        pixels returned per image: uc_map.flatten().topk(k, largest = strategy in
 *            {entropy, least_confidence}).indices                                     query.py:36,57-61
 *   out_idx  i32 [B,k] flat index h*W+w, value-sorted; ties -> lower index first; NaN scores
 *            (0*log 0, query.py:230) sort first for largest, last for smallest.
 *   out_val  f32 [B,k] or NULL — the scores of out_idx.
 *   out_map  f32 [B,H,W] or NULL — the score map after exclusion (what query.py calls uc_map).  With k > 48 and no out_map the
 *   map is never formed (flat fp32 class planes, C = 11 / 19 / 21, k <= H*W / 8, H*W >= 16384): a sampled per-image threshold, the
 *   scorer writes only the candidates beyond it, a list select picks from them; images the sample misleads are redone exactly.
 *   k larger than the number of un-excluded pixels is NOT an error (as in the reference, excluded
 *   pixels are then returned, lowest index first).
 *   (SURVEY.md 8(b)'s sketch carries a `sorted` flag after k.  It is deliberately absent: the reference's torch.topk runs with its
 *   default sorted=True and its callers rely on that order (query.py:50-54 takes the first n of the value-sorted candidates), and the
 *   rank-merge that joins the per-block candidate lists yields the value-sorted order as a by-product - an unsorted mode would
 *   neither match a reference call nor save a launch.)
 */
size_t pp_acq_workspace_bytes(int64_t B, int64_t C, int64_t H, int64_t W, int64_t k);

int pp_acq_score_topk(const float* logits, int64_t B, int64_t C, int64_t H, int64_t W,
                      int64_t sB, int64_t sC, int64_t sH, int64_t sW,
                      const uint8_t* exclude, int strategy, int64_t k,
                      int32_t* out_idx, float* out_val, float* out_map,
                      void* workspace, size_t ws_bytes, pp_stream_t stream);

/* Score map only (query.py:190-201; also model.py:241-260 Model._query).  exclude may be NULL. */
int pp_acq_score_map(const float* logits, int64_t B, int64_t C, int64_t H, int64_t W,
                     int64_t sB, int64_t sC, int64_t sH, int64_t sW,
                     const uint8_t* exclude, int strategy, float* out_map, pp_stream_t stream);

/* MC-dropout accumulation (query.py:181-187 `uc_map += uc_map_; prob += prob_`, then / mc_n_steps): over the T passes in
 * logits [T,C,H,W] (element strides sT,sC,sH,sW), prob_out [C,H,W] (+)= scale * sum_t softmax(logits[t]) and
 * uc_out [H,W] (+)= scale * sum_t score(softmax(logits[t])) with the strategy's formula on the probabilities; either
 * output may be NULL; accumulate == 0 overwrites.  No [T,C,H,W] probability tensor exists. */
int pp_acq_softmax_sum(const float* logits, int64_t T, int64_t C, int64_t H, int64_t W, int64_t sT, int64_t sC, int64_t sH,
                       int64_t sW, float* prob_out, float* uc_out, int strategy, float scale, int accumulate, pp_stream_t stream);

/* query.py:246-247 UncertaintySampler.__call__(prob): prob f32 [B,C,H,W] (already softmaxed, any
 * strides) -> out_map f32 [B,H,W]; formulas of query.py:229-239 verbatim (no exclusion). */
int pp_uncertainty_from_prob(const float* prob, int64_t B, int64_t C, int64_t H, int64_t W,
                             int64_t sB, int64_t sC, int64_t sH, int64_t sW,
                             int strategy, float* out_map, pp_stream_t stream);

/* query.py:57-61 on a GIVEN score map: scores f32 [B,N] contiguous -> out_idx i32 [B,k] (+ out_val).
 * largest != 0 selects the k largest.  Same ordering policy as pp_acq_score_topk. */
size_t pp_topk_workspace_bytes(int64_t B, int64_t N, int64_t k);

int pp_topk_select(const float* scores, int64_t B, int64_t N, int64_t k, int largest,
                   int32_t* out_idx, float* out_val,
                   void* workspace, size_t ws_bytes, pp_stream_t stream);

/* Acquisition straight from the LOW-resolution classifier output.  Replaces, in one launch
 * and without materialising the full-resolution logits,
 *   deeplab.py:55-56   pred = F.interpolate(pred, size=inputs.shape[2:], mode='bilinear', align_corners=True)
 *   query.py:190       prob = F.softmax(model(x)["pred"][:, :, :h, :w], dim=1)      ([:h,:w] = the VOC crop, :171-174)
 *   query.py:229-239, 195-201, 57-61   score, exclusion, top-k   (as pp_acq_score_topk)
 *   low      f32 [B,h,w,ldx] channels-last classifier output (what SegmentHead's 1x1 conv writes), C valid channels
 *   H, W     the size the reference interpolates to (the network input size); align_corners as F.interpolate's flag
 *   Hc, Wc   the crop scored and indexed: pixel (Y,X), Y < Hc <= H, X < Wc <= W, has flat index Y*Wc + X
 *   exclude  u8 [B,Hc,Wc] or NULL;  out_idx i32 [B,k], out_val f32 [B,k] or NULL, out_map f32 [B,Hc,Wc] or NULL
 *   k == 0   writes out_map only (out_idx / workspace may be NULL).
 * The interpolated logits are bit-identical to pp_bilinear_fwd's, so the result equals pp_bilinear_fwd followed by
 * pp_acq_score_topk bit for bit (tested). */
size_t pp_acq_lowres_workspace_bytes(int64_t B, int64_t C, int64_t Hc, int64_t Wc, int64_t k);

int pp_acq_lowres_score_topk(const float* low, int64_t ldx, int64_t B, int64_t C, int64_t h, int64_t w,
                             int64_t H, int64_t W, int align_corners, int64_t Hc, int64_t Wc,
                             const uint8_t* exclude, int strategy, int64_t k,
                             int32_t* out_idx, float* out_val, float* out_map,
                             void* workspace, size_t ws_bytes, pp_stream_t stream);

/* The strategy's score (no exclusion) at n listed pixels of the interpolated map: pixel i is image img_idx[i] (i32),
 * flat index pix_idx[i] = Y*Wc + X (i32) -> out[i].  QueryStats' "entropy at the queried pixels" (query.py:262-266)
 * without the full softmax map. */
int pp_acq_lowres_score_at(const float* low, int64_t ldx, int64_t B, int64_t C, int64_t h, int64_t w,
                           int64_t H, int64_t W, int align_corners, int64_t Hc, int64_t Wc, int strategy,
                           const int32_t* img_idx, const int32_t* pix_idx, int64_t n, float* out,
                           pp_stream_t stream);


/* =============================================================================================
 * Network layers (DeepLabv3+-MobileNetV2 / FPN-ResNet50 forward + backward), NHWC fp32.
 *
 * Activations are [B,H,W,C] with an explicit pixel stride `ld` (elements): a channel slice of a wider
 * tensor is addressed by offsetting the pointer, which is how the reference's torch.cat tensors
 * (aspp.py:73, deeplab.py:50) are produced and consumed without a copy.  Dense-conv weights are HWIO
 * [kh][kw][Cin][Cout]; depthwise weights [3][3][C].  All reductions are deterministic (two-stage, no
 * float atomics).  Workspaces are caller-owned.
 * ============================================================================================= */

/* nn.Conv2d forward (dense, groups=1) as an implicit GEMM on the fp32 MFMA pipe.
 * networks/mobilenet_v2.py:9,42,48,56; aspp.py:9-10,55,58; deeplab.py:24; decoders.py:107,111,116;
 * backbones/resnet_models.py Bottleneck convs; decoders.py:25-28,92 (bias != NULL).
 * y[b,oh,ow,n] = bias[n] + sum x[b, oh*stride - pad + th*dil, ow*stride - pad + tw*dil, c] * w[th,tw,c,n] */
/* `workspace` (optional, may be NULL / 0): scratch for the split-K form used when the output has too few tiles to
 * fill the 256 CUs (partial sums [splits][M][Cout], reduced in fixed order: deterministic).  Size it with
 * pp_conv2d_fwd_workspace_bytes (0 = this shape never splits); with less, the single-pass form runs. */
size_t pp_conv2d_fwd_workspace_bytes(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil);
int pp_conv2d_fwd(const float* x, int64_t ldx, int B, int H, int W, int Cin, const float* w, const float* bias,
                  int kh, int kw, int stride, int pad, int dil, float* y, int64_t ldy, int Cout, void* workspace,
                  size_t ws_bytes, pp_stream_t stream);

/* ---- training BatchNorm split over its neighbours (SURVEY.md 7 hard part (b); mobilenet_v2.py:42-56: pw -> BN -> ReLU6 -> dw ->
 * BN -> ReLU6 -> pw).  The producer's epilogue delivers partial column statistics (pp_conv2d_fwd_stats, pp_dwconv3x3_fwd_fused),
 * pp_bn_finalize_partials turns them into mean / invstd / running statistics and the folded per-channel scale / shift, and the
 * CONSUMER applies act(fma(x, scale, shift)) where it loads its input (pp_dwconv3x3_fwd_fused, pp_conv2d_fwd_affine_in,
 * pp_dwconv3x3_bwd_weight_affine_in): the normalised tensor is never written.  Arithmetic = pp_scale_shift_act's, so a consumer of
 * the raw tensor + (scale, shift, act) computes what it would compute from the materialised tensor, bit for bit.
 *   pp_bn_finalize_partials            stats [rows][2][C] (sum, sum of squares) over M rows -> mean, invstd, scale = gamma*invstd,
 *                                      shift = beta - mean*scale, running-stat update (nn.BatchNorm2d training semantics)
 *   pp_dwconv3x3_fwd_stats_rows        partial rows pp_dwconv3x3_fwd_fused writes for this shape
 *   pp_dwconv3x3_fwd_fused             depthwise 3x3 with optional input affine (in_scale may be NULL) and optional output statistics
 *   pp_dwconv3x3_bwd_weight_affine_in  pp_dwconv3x3_bwd_weight on x = act(fma(x_raw, in_scale, in_shift))
 *   pp_conv2d_fwd_accepts_affine_in    1 when pp_conv2d_fwd_affine_in has a kernel for this shape (stride-1 pad-0 pointwise layers:
 *                                      the in-block split-K kernel, or the 128x32-tiled kernel for <= 32 output channels)
 *   pp_conv2d_fwd_affine_in            pp_conv2d_fwd on x = act(fma(x_raw, in_scale, in_shift)); fails (no fallback) on other shapes */
int pp_bn_finalize_partials(const float* stats, int64_t rows, int64_t M, int C, const float* gamma, const float* beta, float eps,
                            float momentum, float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                            float* shift, pp_stream_t stream);
int64_t pp_dwconv3x3_fwd_stats_rows(int B, int H, int W, int C, int stride, int pad, int dil);
int pp_dwconv3x3_fwd_fused(const float* x, int64_t ldx, int B, int H, int W, int C, const float* w, int stride, int pad, int dil,
                           const float* in_scale, const float* in_shift, int in_act, float* y, int64_t ldy, float* stats,
                           size_t stats_floats, pp_stream_t stream);
int pp_dwconv3x3_bwd_weight_affine_in(const float* x_raw, int64_t ldx, int B, int H, int W, int C, const float* in_scale,
                                      const float* in_shift, int in_act, const float* dy, int64_t lddy, int stride, int pad, int dil,
                                      float* dw, void* workspace, size_t ws_bytes, pp_stream_t stream);
int pp_conv2d_fwd_accepts_affine_in(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil);
int pp_conv2d_fwd_affine_in(const float* x_raw, int64_t ldx, int B, int H, int W, int Cin, const float* in_scale, const float* in_shift,
                            int in_act, const float* w, const float* bias, int kh, int kw, int stride, int pad, int dil, float* y,
                            int64_t ldy, int Cout, void* workspace, size_t ws_bytes, pp_stream_t stream);

/* Inference form of Conv2d -> BatchNorm2d(eval) [-> + residual] -> ReLU/ReLU6 in ONE launch (mobilenet_v2.py:7-12,
 * 36-57; aspp.py:9-20; decoders.py:107-113; resnet_models.py:71-92): the epilogue folds scale = gamma/sqrt(running_var+eps),
 * shift = beta - running_mean*scale with the arithmetic of pp_bn_eval_affine + pp_scale_shift_act (bit-identical results).
 * gamma == NULL: no BatchNorm (bias / residual / activation only).  act: 0 none, 1 ReLU, 2 ReLU6. */
int pp_conv2d_fwd_bn_act(const float* x, int64_t ldx, int B, int H, int W, int Cin, const float* w, const float* bias,
                         int kh, int kw, int stride, int pad, int dil, const float* gamma, const float* beta,
                         const float* running_mean, const float* running_var, float eps, const float* residual,
                         int64_t ldr, int act, float* y, int64_t ldy, int Cout, void* workspace, size_t ws_bytes,
                         pp_stream_t stream);

/* dL/dx of the above (what autograd computes at model.py:121); any stride (the stride-2 Bottleneck convs of
 * backbones/resnet_models.py:63-64,142-144 gather dY rows where (row + pad - tap*dil) is divisible by the stride).
 * B,H,W,Cin describe the conv INPUT for the workspace query.  accumulate != 0: dx += result (the tensor already holds
 * the gradient of another consumer of x, e.g. the residual branch of mobilenet_v2.py:62-63). */
size_t pp_conv2d_bwd_data_workspace_bytes(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil);
int pp_conv2d_bwd_data(const float* dy, int64_t lddy, int B, int Ho, int Wo, int Cout, const float* w, int kh, int kw,
                       int stride, int pad, int dil, float* dx, int64_t lddx, int H, int W, int Cin, int accumulate,
                       void* workspace, size_t ws_bytes, pp_stream_t stream);

/* dL/dw (HWIO) and optionally dL/dbias [Cout]; workspace holds the split-M partial sums. */
size_t pp_conv2d_bwd_weight_workspace_bytes(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil);
int pp_conv2d_bwd_weight(const float* x, int64_t ldx, int B, int H, int W, int Cin, const float* dy, int64_t lddy,
                         int Cout, int kh, int kw, int stride, int pad, int dil, float* dw, float* dbias,
                         void* workspace, size_t ws_bytes, pp_stream_t stream);

/* bf16x3 operand planes shared between the calls of one layer.  The MFMA-bound convolutions (layers the planner puts on the bf16x3 kernels) read their
 * activation operand as three chunk-major bf16 planes; by default every call splits its operand into its workspace.  The
 * forward's operand x is also the weight gradient's, and the backward-data's operand dy is the weight gradient's other one:
 * a caller may split a tensor ONCE with pp_x3_split and pass the planes to every *_pre call that reads it (NULL = split
 * inside, as the plain entry points do; planes are ignored by calls that do not run a bf16x3 kernel).
 * pp_conv2d_x3_planes_bytes(which, ...): 0 if the call (which = 0 forward, 1 backward-data, 2 weight gradient) would not use
 * planes, else the size of its activation planes (forward / weight gradient: of x; backward-data: of dy).
 * which = 3 / 4: the forward / backward-data call reads WEIGHT planes (pp_x3_split_weights, that direction's
 * layout) when it gets them through *_pre2 - asked by callers that keep the weight planes of a step - else 0. */
/* Backward-data of up to four convolutions that read ONE input, as one launch (+ its split-K reduce): the ASPP branches of
 * /root/reference/networks/aspp.py:49-57,64-67 - x1..x4 = aspp1..4(x): a 1x1 and three dilated 3x3 convolutions, stride 1, "same"
 * padding d*(k-1)/2 - whose input gradient torch forms as four conv-backward results added up.  dy [B,H,W,>= nb*Cout] holds branch b's
 * output gradient in channels [b*Cout, (b+1)*Cout); w_b is HWIO [k_b,k_b,Cin,Cout]; dx [B,H,W,Cin] (+)= the sum over the branches.
 * The workspace query returns 0 where the merged form is not offered (the caller then issues pp_conv2d_bwd_data per branch). */
size_t pp_conv2d_bwd_data_multi_workspace_bytes(int B, int H, int W, int Cin, int Cout, int nb, int k0, int d0, int k1, int d1, int k2, int d2,
                                                int k3, int d3);
int pp_conv2d_bwd_data_multi(const float* dy, int64_t lddy, int B, int H, int W, int Cout, int nb, const float* w0, int k0, int d0,
                             const float* w1, int k1, int d1, const float* w2, int k2, int d2, const float* w3, int k3, int d3, float* dx,
                             int64_t lddx, int Cin, int accumulate, void* workspace, size_t ws_bytes, pp_stream_t stream);

size_t pp_x3_planes_bytes(int64_t rows, int C);
int pp_x3_split(const float* x, int64_t ldx, int64_t rows, int C, void* planes, size_t planes_bytes, pp_stream_t stream);
size_t pp_conv2d_x3_planes_bytes(int which, int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil);
int pp_conv2d_fwd_pre(const float* x, int64_t ldx, int B, int H, int W, int Cin, const float* w, const float* bias,
                      int kh, int kw, int stride, int pad, int dil, float* y, int64_t ldy, int Cout, void* workspace,
                      size_t ws_bytes, const void* x_planes, pp_stream_t stream);
int pp_conv2d_bwd_data_pre(const float* dy, int64_t lddy, int B, int Ho, int Wo, int Cout, const float* w, int kh, int kw,
                           int stride, int pad, int dil, float* dx, int64_t lddx, int H, int W, int Cin, int accumulate,
                           void* workspace, size_t ws_bytes, const void* dy_planes, pp_stream_t stream);
/* The WEIGHTS' planes as well (round 5): they change once per optimiser step, so a training loop may split them at the start of the step on
 * another stream - transpose = 1: the forward's layout, 0: the backward-data's - and pass them to the *_pre2 forms (NULL = split inside;
 * ignored by calls that do not run a bf16x3 kernel).  /root/reference/networks/decoders.py:107-114 (the SegmentHead convolutions). */
size_t pp_x3_weight_planes_bytes(int kh_kw, int Cin, int Cout, int transpose);
int pp_x3_split_weights(const float* w, int kh_kw, int Cin, int Cout, int transpose, void* planes, size_t planes_bytes, pp_stream_t stream);
int pp_conv2d_fwd_pre2(const float* x, int64_t ldx, int B, int H, int W, int Cin, const float* w, const float* bias,
                       int kh, int kw, int stride, int pad, int dil, float* y, int64_t ldy, int Cout, void* workspace,
                       size_t ws_bytes, const void* x_planes, const void* w_planes, pp_stream_t stream);
int pp_conv2d_bwd_data_pre2(const float* dy, int64_t lddy, int B, int Ho, int Wo, int Cout, const float* w, int kh, int kw,
                            int stride, int pad, int dil, float* dx, int64_t lddx, int H, int W, int Cin, int accumulate,
                            void* workspace, size_t ws_bytes, const void* dy_planes, const void* w_planes, pp_stream_t stream);
int pp_conv2d_bwd_weight_pre(const float* x, int64_t ldx, int B, int H, int W, int Cin, const float* dy, int64_t lddy,
                             int Cout, int kh, int kw, int stride, int pad, int dil, float* dw, float* dbias,
                             void* workspace, size_t ws_bytes, const void* x_planes, const void* dy_planes, pp_stream_t stream);

/* Convolution + training BatchNorm (+ residual, activation) in ONE launch (mobilenet_v2.py:42-43,48-49,56-57: conv -> BatchNorm2d
 * [-> ReLU6]): the blocks of the convolution exchange their column sums the way the blocks of pp_bn_train_fwd_fused do (same
 * fine-grained `xchg` / `sync` areas, same epoch protocol) and normalise the tile they still hold in registers.  conv_out
 * receives the raw convolution (the BatchNorm backward's input), y the normalised output; mean / invstd / running statistics as
 * pp_bn_train_fwd_fused.  Only for shapes whose whole grid is co-resident: ask pp_conv2d_fwd_bn_train_ok (1 / 0) first - the call
 * refuses other shapes instead of falling back. */
int pp_conv2d_fwd_bn_train_ok(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil);
size_t pp_conv2d_fwd_bn_train_xchg_bytes(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil);
int pp_conv2d_fwd_bn_train(const float* x, int64_t ldx, int B, int H, int W, int Cin, const float* w, int kh, int kw, int stride, int pad,
                           int dil, float* conv_out, int64_t ldc, const float* gamma, const float* beta, float eps, float momentum,
                           float* running_mean, float* running_var, float* mean, float* invstd, const float* residual, int64_t ldr,
                           int act, float* y, int64_t ldy, int Cout, void* xchg, size_t xchg_bytes, int32_t* sync, size_t sync_ints,
                           pp_stream_t stream);

/* Backward-data convolution + the BatchNorm backward of the layer in FRONT of it, in one launch (the mirror of
 * pp_conv2d_fwd_bn_train; conv -> BatchNorm -> activation -> THIS convolution, the activated tensor having no other consumer:
 * mobilenet_v2.py:52-56, depthwise BatchNorm / ReLU6 -> project convolution).  dy is the gradient of this convolution's output;
 * bn_x / mean / invstd / gamma / beta / act describe the BatchNorm whose output this convolution read; dx_bn receives the gradient of
 * the BatchNorm's INPUT (what pp_bn_bwd_fused would have written), dgamma / dbeta its parameter gradients.  The gradient of the
 * BatchNorm's output is never written.  Same xchg / sync areas as the single-launch BatchNorm; ask *_ok first.
 * A BatchNorm output with more consumers (mobilenet_v2.py:63-66: the block output feeds the next block's expand convolution AND
 * its residual add) is handled when THIS convolution's backward is the last of them to run: grad_in (or NULL) = the gradient the
 * output already holds from the others, added to the tile first (the `accumulate` of a plain backward-data); dres (or NULL) receives
 * the gradient of the BatchNorm's own residual input (the masked total gradient, what pp_bn_bwd_fused writes to its dres). */
int pp_conv2d_bwd_data_bn_bwd_ok(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil);
size_t pp_conv2d_bwd_data_bn_bwd_xchg_bytes(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil);
int pp_conv2d_bwd_data_bn_bwd(const float* dy, int64_t lddy, int B, int Ho, int Wo, int Cout, const float* w, int kh, int kw, int stride,
                              int pad, int dil, int H, int W, int Cin, const float* bn_x, int64_t ldbx, const float* mean,
                              const float* invstd, const float* gamma, const float* beta, int act, float* dgamma, float* dbeta,
                              float* dx_bn, int64_t lddx, const float* grad_in, int64_t ldgi, float* dres, int64_t lddr,
                              void* xchg, size_t xchg_bytes, int32_t* sync, size_t sync_ints, pp_stream_t stream);

/* Deferred reduces.  A weight gradient is a partial-sum kernel ([slices][...] in the workspace) followed by a small
 * fixed-order reduce; a backward pass has ~60 of them (model.py:121 loss.backward()).  The *_partials forms run only the
 * first kernel and describe the reduce in *job (kind 0: nothing left to do - the call completed the gradient itself, e.g.
 * when a bias gradient is asked for); pp_wgrad_reduce_batch then runs any number of jobs in one launch per 64 jobs, with
 * the arithmetic of the single-layer kernels (bit-identical).  The caller keeps every job's workspace untouched until
 * the batch has been enqueued on the same stream. */
typedef struct pp_reduce_job {
    const float* part;   /* partial sums */
    float* dst;          /* the gradient tensor */
    int64_t cn;          /* elements per tap (kinds 1, 2) / outputs (kind 3) */
    int32_t splits;      /* partial slices */
    int32_t ntaps;       /* live taps (kinds 1, 2) */
    int32_t kind;        /* 0 none, 1 float4 sequential, 2 32-lane tree, 3 32-lane fp64 (depthwise) */
    uint8_t widx[12];    /* weight-tensor tap index of every live tap */
} pp_reduce_job;
int pp_conv2d_bwd_weight_partials(const float* x, int64_t ldx, int B, int H, int W, int Cin, const float* dy, int64_t lddy,
                                  int Cout, int kh, int kw, int stride, int pad, int dil, float* dw, float* dbias,
                                  void* workspace, size_t ws_bytes, pp_reduce_job* job, pp_stream_t stream);
int pp_dwconv3x3_bwd_weight_partials(const float* x, int64_t ldx, int B, int H, int W, int C, const float* dy, int64_t lddy,
                                     int stride, int pad, int dil, float* dw, void* workspace, size_t ws_bytes, pp_reduce_job* job,
                                     pp_stream_t stream);
int pp_wgrad_reduce_batch(const pp_reduce_job* jobs, int n, pp_stream_t stream);

/* Column-reduction workspace shared by pp_bn_train_fwd, pp_bn_bwd and pp_dwconv3x3_bwd_weight. */
size_t pp_colreduce_workspace_bytes(int64_t M, int C);

/* nn.BatchNorm2d, training mode (mobilenet_v2.py:10,39,49,57; aspp.py:11,56,59; deeplab.py:25;
 * decoders.py:108,112): batch mean / biased variance over the M = B*H*W rows, running-stat update
 * (momentum, unbiased variance), and the per-channel affine  scale = gamma*invstd,
 * shift = beta - mean*scale  to be applied by pp_scale_shift_act.  running_* may be NULL. */
int pp_bn_train_fwd(const float* x, int64_t ldx, int64_t M, int C, const float* gamma, const float* beta, float eps,
                    float momentum, float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                    float* shift, void* workspace, size_t ws_bytes, pp_stream_t stream);

/* The same BatchNorm2d training forward INCLUDING the apply (+ residual, + activation) in ONE launch
 * (column strips x row chunks; blocks exchange their partial sums inside the launch; see nn_ops.hip).
 * `workspace` (pp_bn_fused_workspace_bytes, 8-byte aligned) holds one 64-bit {launch tag, value} word per partial and
 * `sync` (pp_bn_fused_sync_ints ints) the launch epoch and a done-counter.  BOTH MUST be FINE-GRAINED device memory
 * (hipExtMallocWithFlags(hipDeviceMallocFinegrained)), zero-filled ONCE before first use and then left alone: the
 * blocks of a launch sit on different XCDs whose L2s are not coherent for ordinary allocations (stale partials and
 * run-to-run differences were observed with ordinary memory; pixelpick_amd/engine.py `_bn_exchange` allocates the area
 * once per device).  They must not be shared by launches that can run concurrently.  Results are deterministic. */
size_t pp_bn_fused_workspace_bytes(int64_t M, int C);
/* 1 when a [M, C] map is small enough for the single-launch training BatchNorm kernels to keep every thread's rows in
 * registers between their two passes (no second read of x / dy); then the depthwise convolution in front of a BatchNorm is
 * also computed inside that launch (pp_dwconv3x3_bn_train_fwd_fused) at no extra traffic.  mobilenet_v2.py:38-39,52-53. */
int pp_bn_fused_rows_cached(int64_t M, int C);
size_t pp_bn_fused_sync_ints(int C);
int pp_bn_train_fwd_fused(const float* x, int64_t ldx, int64_t M, int C, const float* gamma, const float* beta, float eps,
                          float momentum, float* running_mean, float* running_var, float* mean, float* invstd,
                          const float* residual, int64_t ldr, int act, float drop_p, uint64_t drop_seed,
                          const uint64_t* drop_seed_dev, float* y, int64_t ldy, void* workspace,
                          size_t ws_bytes, int32_t* sync, size_t sync_ints, pp_stream_t stream);
/* drop_p > 0: the nn.Dropout(p) that follows BN -> ReLU (aspp.py:60-61, decoders.py:108-114) is applied in the same pass
 * with pp_dropout's mask stream (same seed / seed_dev -> same mask), act must be 0 or 1. */

/* Training-mode BatchNorm whose statistics come from the PRODUCER of x (round 2): pp_conv2d_fwd_stats writes, next to the
 * convolution output, per-wave (or, behind a split-K convolution, per reduce-block) column sums and sums of squares
 * stats[rows][2][Cout] (rows = pp_conv2d_fwd_stats_rows(shape); 0 = this shape delivers none), and
 * pp_bn_train_fwd_partials reduces them (fixed order, fp64) in every block and applies scale/shift (+ residual, activation,
 * dropout) in one pass over x: no statistics pass, no exchange between blocks, no co-residency requirement
 * (mobilenet_v2.py:42-43,56-57 conv -> BatchNorm pairs and every other dense conv -> BatchNorm pair of the networks). */
int64_t pp_conv2d_fwd_stats_rows(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil);
int pp_conv2d_fwd_stats(const float* x, int64_t ldx, int B, int H, int W, int Cin, const float* w, const float* bias,
                        int kh, int kw, int stride, int pad, int dil, float* y, int64_t ldy, int Cout, void* workspace,
                        size_t ws_bytes, float* stats, size_t stats_floats, pp_stream_t stream);
int pp_bn_train_fwd_partials(const float* x, int64_t ldx, int64_t M, int C, const float* stats, int64_t stat_rows, const float* gamma,
                             const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* mean,
                             float* invstd, const float* residual, int64_t ldr, int act, float drop_p, uint64_t drop_seed,
                             const uint64_t* drop_seed_dev, float* y, int64_t ldy, pp_stream_t stream);

/* Depthwise 3x3 convolution + training-mode BatchNorm (+ residual) + activation in ONE launch (mobilenet_v2.py:38-40, 52-54):
 * the single-launch BatchNorm computes the convolution in its statistics pass, writes it to x_out [B,Ho,Wo,C] (BatchNorm's
 * input, needed by the backward pass) and applies the normalisation from there.  Bit-identical to pp_dwconv3x3_fwd followed
 * by pp_bn_train_fwd_fused.  workspace / sync as pp_bn_train_fwd_fused (fine-grained memory). */
int pp_dwconv3x3_bn_train_fwd_fused(const float* in, int64_t ld_in, int B, int H, int W, int C, const float* w, int stride, int pad,
                                    int dil, float* x_out, int64_t ldx, const float* gamma, const float* beta, float eps,
                                    float momentum, float* running_mean, float* running_var, float* mean, float* invstd,
                                    const float* residual, int64_t ldr, int act, float* y, int64_t ldy, void* workspace,
                                    size_t ws_bytes, int32_t* sync, size_t sync_ints, pp_stream_t stream);

/* Single-launch form of pp_bn_bwd (same arguments + sync).  grad_scale = 1/(1-p) when a dropout was fused into the
 * forward (its mask is recovered from y_act == 0), else 1.  y_act == NULL with beta != NULL (no residual, no dropout):
 * the ReLU/ReLU6 mask is recomputed from x with the forward's own fma (bit-identical), y_act is not read. */
int pp_bn_bwd_fused(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* y_act, int64_t ldya, int act,
                    int64_t M, int C, const float* mean, const float* invstd, const float* gamma, float* dgamma,
                    float* dbeta, float* dx, int64_t lddx, float* dres, int64_t lddr, float grad_scale, const float* beta,
                    void* workspace, size_t ws_bytes, int32_t* sync, size_t sync_ints, pp_stream_t stream);

/* Sparse output gradients.  The loss of a sparsely labelled batch (model.py:113-119: 20 labelled pixels per image, ignore_index
 * elsewhere) has a gradient that is zero in all but a few hundred of the 32768 low-resolution rows.  pp_row_flags marks the non-zero
 * rows of a [M, C] gradient (flags[r] = 1 when any entry of row r is non-zero); pp_conv1x1_bwd_data_sparse is the backward-data of a
 * pointwise convolution (w: HWIO of a 1x1 kernel = [Cin][Cout]) that writes zeros for the unflagged rows of dx and computes the flagged
 * ones; pp_bn_bwd_fused_sparse is pp_bn_bwd_fused told that the unflagged rows of dy are exact zeros: its statistics pass visits only
 * the flagged rows (in the dense kernel's order: dgamma / dbeta / dx are bit-equal to pp_bn_bwd_fused on the same dy) and its dx pass
 * never reads dy elsewhere.  A row cache variant of the kernel (small maps) ignores the flags. */
int pp_row_flags(const float* dy, int64_t lddy, int64_t M, int C, unsigned char* flags, pp_stream_t stream);
/* The weight (and, dbias != NULL, bias) gradient of that pointwise convolution over the flagged rows only: dw [Cin][Cout] (HWIO of a 1x1
 * kernel) = sum_r x[r][:]^T dy[r][:], rows in ascending order inside fixed 512-row blocks, blocks in order (deterministic; equal to the
 * dense gradient up to the order of the fp32 additions).  Cin * Cout + Cout <= 8192.  decoders.py:64 / :120 classifiers behind
 * model.py:113-119's sparse labels. */
size_t pp_conv1x1_bwd_weight_sparse_workspace_bytes(int64_t M, int Cin, int Cout);
int pp_conv1x1_bwd_weight_sparse(const float* x, int64_t ldx, int64_t M, int Cin, const float* dy, int64_t lddy, int Cout,
                                 const unsigned char* row_flags, float* dw, float* dbias, void* workspace, size_t ws_bytes, pp_stream_t stream);
int pp_conv1x1_bwd_data_sparse(const float* dy, int64_t lddy, int64_t M, int Cout, const float* w, int Cin, const unsigned char* row_flags,
                             float* dx, int64_t lddx, pp_stream_t stream);
int pp_bn_bwd_fused_sparse(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* y_act, int64_t ldya, int act,
                         int64_t M, int C, const float* mean, const float* invstd, const float* gamma, float* dgamma,
                         float* dbeta, float* dx, int64_t lddx, float* dres, int64_t lddr, float grad_scale, const float* beta,
                         void* workspace, size_t ws_bytes, int32_t* sync, size_t sync_ints, const unsigned char* row_flags,
                         pp_stream_t stream);

/* nn.BatchNorm2d, eval mode: scale/shift from the running statistics. */
int pp_bn_eval_affine(int C, const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                      float eps, float* scale, float* shift, pp_stream_t stream);

/* y = act(x*scale + shift [+ residual]);  act 0 none, 1 ReLU, 2 ReLU6.  BN apply + nn.ReLU/ReLU6
 * (+ the residual add of mobilenet_v2.py:62-63 / resnet Bottleneck) in one pass. */
int pp_scale_shift_act(const float* x, int64_t ldx, int64_t M, int C, const float* scale, const float* shift,
                       const float* residual, int64_t ldr, int act, float* y, int64_t ldy, pp_stream_t stream);

/* Backward of [BN(train) -> (+residual) -> act]: g = dy * act'(y_act); dgamma, dbeta, dx; dres = g
 * (NULL if no residual).  y_act is the activation OUTPUT (needed when act != 0). */
int pp_bn_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* y_act, int64_t ldya, int act,
              int64_t M, int C, const float* mean, const float* invstd, const float* gamma, float* dgamma, float* dbeta,
              float* dx, int64_t lddx, float* dres, int64_t lddr, void* workspace, size_t ws_bytes, pp_stream_t stream);

/* Depthwise 3x3 convolution (groups = C), mobilenet_v2.py:38,52; generic stride / padding / dilation.
 * H, W are always the INPUT spatial size. */
int pp_dwconv3x3_fwd(const float* x, int64_t ldx, int B, int H, int W, int C, const float* w, int stride, int pad, int dil,
                     float* y, int64_t ldy, pp_stream_t stream);

/* Inference form of the depthwise 3x3 -> BatchNorm2d(eval) -> ReLU6 (mobilenet_v2.py:38-40,52-54) in one launch;
 * same epilogue contract as pp_conv2d_fwd_bn_act. */
int pp_dwconv3x3_fwd_bn_act(const float* x, int64_t ldx, int B, int H, int W, int C, const float* w, int stride, int pad,
                            int dil, const float* gamma, const float* beta, const float* running_mean,
                            const float* running_var, float eps, const float* residual, int64_t ldr, int act, float* y,
                            int64_t ldy, pp_stream_t stream);
int pp_dwconv3x3_bwd_data(const float* dy, int64_t lddy, int B, int H, int W, int C, const float* w, int stride, int pad,
                          int dil, float* dx, int64_t lddx, pp_stream_t stream);
int pp_dwconv3x3_bwd_weight(const float* x, int64_t ldx, int B, int H, int W, int C, const float* dy, int64_t lddy,
                            int stride, int pad, int dil, float* dw, void* workspace, size_t ws_bytes, pp_stream_t stream);

/* nn.GroupNorm(G, C) [+ nn.ReLU] (decoders.py:92-94), x [B,P=H*W,C]: statistics per (image, group) over P*(C/G)
 * elements, biased variance, eps inside the sqrt.  mean/rstd [B*G] are outputs of fwd and inputs of bwd; y (the
 * activation output) supplies the ReLU mask in bwd. */
size_t pp_groupnorm_workspace_bytes(int B, int64_t P, int C);
int pp_groupnorm_relu_fwd(const float* x, int64_t ldx, int B, int64_t P, int C, int G, const float* gamma, const float* beta,
                          float eps, int relu, float* y, int64_t ldy, float* mean, float* rstd, void* workspace,
                          size_t ws_bytes, pp_stream_t stream);
int pp_groupnorm_relu_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* y, int64_t ldy, int B,
                          int64_t P, int C, int G, const float* mean, const float* rstd, const float* gamma, float* dgamma,
                          float* dbeta, float* dx, int64_t lddx, void* workspace, size_t ws_bytes, pp_stream_t stream);

/* nn.MaxPool2d(ksize, stride, pad) (resnet_models.py:121), torch's first-maximum rule; argmax u8 [B,Ho,Wo,C] holds the
 * winning tap (kh*ksize + kw) and drives the deterministic gather in bwd.  H, W are the INPUT size in both calls. */
int pp_maxpool2d_fwd(const float* x, int64_t ldx, int B, int H, int W, int C, int ksize, int stride, int pad, float* y,
                     int64_t ldy, unsigned char* argmax, pp_stream_t stream);
int pp_maxpool2d_bwd(const float* dy, int64_t lddy, const unsigned char* argmax, int B, int H, int W, int C, int ksize, int stride,
                     int pad, float* dx, int64_t lddx, pp_stream_t stream);

/* fixed_padding (mobilenet_v2.py:15-21): zero-pad [B,H,W,C] to [B,Hp,Wp,C]; and its adjoint
 * y = crop(xp) (+ add), used for the gradient of the padded block input (+ the residual gradient). */
int pp_pad2d(const float* x, int64_t ldx, int B, int H, int W, int C, int pad_top, int pad_left, int Hp, int Wp, float* y,
             int64_t ldy, pp_stream_t stream);
int pp_crop2d_add(const float* xp, int64_t ldxp, int B, int Hp, int Wp, int C, int pad_top, int pad_left, const float* add,
                  int64_t ldadd, float* y, int64_t ldy, int H, int W, pp_stream_t stream);

/* F.interpolate(mode="bilinear") with torch's index arithmetic (deeplab.py:49,55; aspp.py:70;
 * decoders.py:82,101).  align_corners as given; scale_h/scale_w > 0 reproduce the `scale_factor=`
 * call form (decoders.py:101), 0 uses in/out.  out_nchw: y is [B,C,Ho,Wo] contiguous (the model's
 * "pred", deeplab.py:55-56).  Backward is a deterministic gather; dy_nchw likewise. */
int pp_bilinear_fwd(const float* x, int64_t ldx, int B, int H, int W, int C, float* y, int64_t ldy, int Ho, int Wo,
                    int align_corners, float scale_h, float scale_w, int out_nchw, pp_stream_t stream);
/* workspace (optional, may be NULL): B*Ho*W*C floats let up-sampling factors >= 3 run as two separable gathers. */
size_t pp_bilinear_bwd_workspace_bytes(int B, int Ho, int W, int C);
int pp_bilinear_bwd(const float* dy, int64_t lddy, int B, int Ho, int Wo, int C, float* dx, int64_t lddx, int H, int W,
                    int align_corners, float scale_h, float scale_w, int dy_nchw, void* workspace, size_t ws_bytes,
                    pp_stream_t stream);

/* out[b][c] = mul * sum_p x[b][p][c]  (nn.AdaptiveAvgPool2d(1), aspp.py:54, with mul = 1/P; also the
 * adjoint of the broadcast below) and y[b][p][c] = mul * v[b][c] (the 1x1 -> HxW bilinear broadcast of
 * aspp.py:70, and the adjoint of the pool). */
int pp_image_colsum(const float* x, int64_t ldx, int B, int64_t P, int C, float mul, float* out, int64_t ldo, pp_stream_t stream);
int pp_image_broadcast(const float* v, int64_t ldv, int B, int64_t P, int C, float mul, float* y, int64_t ldy, pp_stream_t stream);

/* nn.Dropout (aspp.py:61; decoders.py:110,114): y = x * keep / (1-p), keep from a counter-based hash of
 * (seed, element index) — the backward pass calls this again on dy with the same seed. */
/* seed_dev (NULL or a device u64): added (hashed) to `seed` when the kernel RUNS, so a launch captured in a hipGraph
 * draws a fresh mask on every replay once the host bumps that word. */
int pp_dropout(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t M, int C, float p, uint64_t seed,
               const uint64_t* seed_dev, pp_stream_t stream);

/* ---- training-time augmentation on the device (datasets/base_dataset.py:48-141) -----------------
 * Images are HWC uint8 RGB (what PIL hands the reference), label maps / query masks [H,W] uint8.  The host builds the
 * resampling tables (pixelpick_amd/augment.py restates libImaging's precompute_coeffs / nearest index rules and torch's
 * nearest rule); the kernels evaluate them bit-exactly: clip8((2^21 + sum px*k) >> 22) per pass.
 *   pp_aug_resample_h   TF.resize(x, BILINEAR), horizontal pass: src [H,W,3] -> dst [H,Wout,3]; bounds [Wout,2] (first
 *                       source column, count), kk [Wout,ksize] 22-bit fixed-point coefficients.
 *   pp_aug_vcrop        vertical pass evaluated only where TF.pad(fill) -> TF.crop(start, size) -> TF.hflip look:
 *                       tmp [h_in, w_rs, 3] -> out [ch,cw,3]; rows/columns beyond (h_rs, w_rs) take the fill colour.
 *   pp_aug_labels       label map through PIL-NEAREST tables (ty, tx), query mask through torch-nearest tables (qy, qx),
 *                       same pad (ignore_index / 0) / crop / flip; y_out int64 [ch,cw], q_out uint8 0/1 (base_dataset.py:114).
 *   pp_aug_jitter       in place; op 0 brightness, 1 contrast, 2 saturation (PIL ImageEnhance blends), 3 hue (PIL RGB->HSV,
 *                       uint8 wrap-around shift, HSV->RGB), 4 RandomGrayscale (convert("L") x3); contrast needs 8 scratch bytes.
 *   pp_aug_blur         cv2.GaussianBlur(img, (ks,ks), sigma) with a host-built float kernel; scratch [H*W*3] floats (the float32
 *                       separable filter cv2 ran for 8-bit images before 3.4.2).
 *   pp_aug_blur_q8      the same call as OpenCV >= 3.4.2 / 4.x computes it for 8-bit images (base_dataset.py:208 today): host-built
 *                       8.8 fixed-point taps summing to 256 (device array), exact integer row pass into scratch [H*W*3] uint16,
 *                       16.16 column pass, round half up; BORDER_REFLECT_101.
 *   pp_aug_to_tensor    TF.normalize(TF.to_tensor(x), mean, std): HWC uint8 -> CHW float32; mean3 / std3 are HOST arrays. */
int pp_aug_resample_h(const uint8_t* src, int H, int W, const int32_t* bounds, const int32_t* kk, int ksize, int Wout, uint8_t* dst,
                      pp_stream_t stream);
int pp_aug_vcrop(const uint8_t* tmp, const int32_t* bounds, const int32_t* kk, int ksize, int h_rs, int w_rs, int start_h, int start_w,
                 int ch, int cw, int flip, int fill_r, int fill_g, int fill_b, uint8_t* out, pp_stream_t stream);
int pp_aug_labels(const uint8_t* y, const uint8_t* q, int W, const int32_t* ty, const int32_t* tx, const int32_t* qy, const int32_t* qx,
                  int h_rs, int w_rs, int start_h, int start_w, int ch, int cw, int flip, int ignore_index, int64_t* y_out,
                  uint8_t* q_out, pp_stream_t stream);
int pp_aug_jitter(uint8_t* img, int64_t n_pixels, int op, float factor, unsigned long long* scratch_sum, pp_stream_t stream);
int pp_aug_blur(uint8_t* img, int H, int W, const float* kernel, int ks, float* scratch, pp_stream_t stream);
int pp_aug_blur_q8(uint8_t* img, int H, int W, const uint16_t* kernel_q8, int ks, uint16_t* scratch, pp_stream_t stream);
int pp_aug_to_tensor(const uint8_t* img, int64_t n_pixels, const float* mean3, const float* std3, float* out, pp_stream_t stream);

/* nn.Dropout2d (mobilenet_v2.py:114-115 on the high-level features in MC-dropout TRAINING, :127,133-134 on the low-level
 * features): x, y [B,P,C] channels-last; one keep/drop draw per (sample, channel) from the hash of (seed, b*C + c).  The
 * backward pass calls it again on dy with the same seed. */
int pp_dropout2d(const float* x, int64_t ldx, float* y, int64_t ldy, int B, int64_t P, int C, float p, uint64_t seed,
                 const uint64_t* seed_dev, pp_stream_t stream);

/* F.cross_entropy(logits, target, ignore_index) (model.py:116) on NCHW logits (plane stride 1 within a
 * class: element (b,c,pix) at b*sB + c*sC + pix): *loss = mean over labelled pixels, *count = their
 * number; dlogits (NULL to skip) = grad_out * (softmax - onehot)/count on labelled pixels, 0 elsewhere,
 * [B,C,HW] contiguous.  grad_out NULL means 1. */
size_t pp_sparse_ce_workspace_bytes(void);
int pp_sparse_ce_fwd_bwd(const float* logits, int B, int C, int64_t HW, int64_t sB, int64_t sC, const int64_t* target,
                         int ignore_index, float* loss, float* count, const float* grad_out, float* dlogits, void* workspace,
                         size_t ws_bytes, pp_stream_t stream);

/* The same loss and its gradient taken straight from the LOW-resolution classifier output, for a model whose last op is
 * deeplab.py:55-56  F.interpolate(low, size=(H,W), mode='bilinear', align_corners):
 *   *loss = F.cross_entropy(F.interpolate(low), target, ignore_index)  (model.py:116),  *count = labelled pixels,
 *   dlow (NULL to skip) f32 [B,h,w,lddx] = grad_out * d loss / d low  (what autograd hands to the classifier conv).
 * low f32 [B,h,w,ldx] channels-last, C valid channels; target i64 [B,H,W].  The labels are scanned once, the class vector
 * is interpolated only at labelled pixels (80 of 524 288 in the BASELINE step) and the backward gathers per low-res
 * pixel in a fixed order (no atomics: bitwise reproducible).  Neither the [B,C,H,W] logits nor their gradient exist. */
size_t pp_sparse_ce_lowres_workspace_bytes(void);
int pp_sparse_ce_lowres_fwd_bwd(const float* low, int64_t ldx, int B, int C, int h, int w, int H, int W, int align_corners,
                                const int64_t* target, int ignore_index, float* loss, float* count, const float* grad_out,
                                float* dlow, int64_t lddx, void* workspace, size_t ws_bytes, pp_stream_t stream);

/* Step metrics on the device (model.py:124-125,194-196 + utils/metrics.py:168-177): hist[t*C + argmax_c logits] += 1
 * for every pixel whose target t is in [0, C) (ignore_index >= C is skipped like RunningScore._fast_hist).  hist is
 * an int64 [C,C] accumulator the caller zeroes / reads (C*C*8 bytes D2H instead of two full maps). */
int pp_confusion_matrix_update(const float* logits, int B, int C, int64_t HW, int64_t sB, int64_t sC, const int64_t* target,
                               int64_t* hist, pp_stream_t stream);

/* torch.optim.Adam step on flat buffers (utils/utils.py:125-141): elements [0,n_split) use lr_a (the
 * backbone/encoder group at lr/10), the rest lr_b; L2 weight decay; `step` is 1-based; grads are
 * multiplied by grad_scale first (1/world_size after the gradient all-reduce). */
/* hyper_dev (NULL or device f32[4] = {lr_a, lr_b, 1-beta1^step, sqrt(1-beta2^step)}): when given, these four
 * replace the by-value arguments at RUN time, so the launch can be replayed from a hipGraph with a moving step. */
int pp_adam_step_flat(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t n_split,
                      float lr_a, float lr_b, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                      float grad_scale, const float* hyper_dev, pp_stream_t stream);

/* torch.optim.SGD(momentum, weight_decay) on flat buffers with the same two-segment learning rate: the optimiser the
 * reference builds for voc and for optimizer_type "SGD" (utils/utils.py:208-270: lr 1e-3 backbone/encoder, 1e-2 the
 * rest, momentum 0.9, weight decay 5e-4 / 1e-4).  step == 1 initialises the momentum buffer with the gradient, as
 * torch does.  hyper_dev (optional): [lr_a, lr_b] read at run time. */
int pp_sgd_step_flat(float* params, const float* grads, float* momentum_buf, int64_t n, int64_t n_split, float lr_a, float lr_b,
                     float momentum, float weight_decay, int64_t step, float grad_scale, const float* hyper_dev,
                     pp_stream_t stream);

/* y = a + b on [M,C] matrices with pixel strides (gradient accumulation for tensors with several
 * consumers: the residual input of mobilenet_v2.py:62, the ASPP input of aspp.py:64-69). */
int pp_add2d(const float* a, int64_t lda, const float* b, int64_t ldb, float* y, int64_t ldy, int64_t M, int C, pp_stream_t stream);

/* [B,C,H,W] -> [B,H,W,C] (the network input; deeplab.py:43 receives NCHW). */
int pp_nchw_to_nhwc(const float* x, int B, int C, int64_t HW, float* y, int64_t ldy, pp_stream_t stream);
/* [B,H,W,C] (pixel stride ldx) -> [B,C,H,W] contiguous: the FPN model's "pred"/"emb" outputs (decoders.py:77). */
int pp_nhwc_to_nchw(const float* x, int64_t ldx, int B, int C, int64_t HW, float* y, pp_stream_t stream);

/* Blocks of the single-launch BatchNorm kernels that fit on the device at once (occupancy x CUs; 0 = no device).  A launch
 * uses at most half of it, so that two spin-waiting launches (second stream / second process) are always co-resident. */
int pp_bn_fused_capacity(void);
/* Measurement yardsticks (stateless launches; bench.py reports the product kernels against them).
 * pp_yardstick_stream_read: a kernel that only reads `bytes` of x (float4 per lane, `blocks` blocks of 256 threads, 0 = 256;
 * blocks < 0: -blocks blocks, non-temporal loads) - acq_kernel's bandwidth is quoted against what this reaches on the same buffer.
 * pp_yardstick_mfma_stream: conv_x3_kernel's MFMA stream from registers only; data_kind 0 zeros, 1 near-constant, 2 random operands. */
int pp_yardstick_stream_read(const void* x, size_t bytes, int blocks, float* sink, pp_stream_t stream);
int pp_yardstick_mfma_stream(int data_kind, int iters, float* sink, pp_stream_t stream);

/* The A/B and ablation switches of earlier rounds (process-global planner state, `pp_debug_*`) are NOT part of this library: they exist
 * only in the test build, libpixelpick_hip_knobs.so (same sources + -DPP_DEBUG_KNOBS, see pixelpick_hip_knobs.h); `nm -D` of
 * libpixelpick_hip.so shows none of them, and its planners run on their compiled-in defaults. */
#ifdef PP_DEBUG_KNOBS
#include "pixelpick_hip_knobs.h"
#endif

/* ---------------------------------------------------------------------------------------------
 * Launch plans: the train step of model.py:101-122 (model.train(); forward; cross_entropy; backward; optimizer.step()) as ONE
 * foreign call.  The reference re-runs that Python per iteration (model.py:106 loop); with static shapes every C-ABI call of
 * a step repeats with identical arguments, so the host records them once and this executor re-issues them (csrc/plan.hip).
 * A recorded call goes through its entry point again - same planning, kernels, queues and results as the eager call.
 *
 *   pp_plan_add_call       fn = address of one of this library's enqueuing entry points; slots = its arguments, one 8-byte
 *                          slot each (integers sign-extended, pointers as addresses, float in the low 4 bytes);
 *                          PP_ERR_UNSUPPORTED for any other address, PP_ERR_BAD_ARG when n_slots is not its arity
 *                          (pp_plan_entry_args(fn), -1 = not an entry point).
 *   pp_plan_add_event_record / _stream_wait    hipEventRecord(event, stream) / hipStreamWaitEvent(stream, event):
 *                          the fork of the weight-gradient queue; caller-owned hipEvent_t.
 *   pp_plan_add_join       `waiting` waits for everything enqueued on `waited_for` so far (plan-owned event).
 *   pp_plan_add_host_break pp_plan_replay returns here so that the caller can act (the gradient all-reduce of a data-parallel
 *                          step, trainer.py) and resume.
 *   pp_plan_replay         issues ops [from, ...) until the end or the next host break; *next = index to resume from
 *                          (== pp_plan_size: finished).  Returns the failing entry point's code (then *next = its index).
 * The plan stores addresses only: every buffer, stream and event it names must outlive it. */
typedef void* pp_plan_t;
pp_plan_t pp_plan_create(void);
void pp_plan_destroy(pp_plan_t plan);
int64_t pp_plan_size(pp_plan_t plan);
int pp_plan_entry_args(const void* fn);
int pp_plan_add_call(pp_plan_t plan, const void* fn, const uint64_t* slots, int n_slots);
int pp_plan_add_event_record(pp_plan_t plan, void* event, pp_stream_t stream);
int pp_plan_add_stream_wait(pp_plan_t plan, pp_stream_t stream, void* event);
int pp_plan_add_join(pp_plan_t plan, pp_stream_t waiting, pp_stream_t waited_for);
int pp_plan_add_host_break(pp_plan_t plan);
int pp_plan_replay(pp_plan_t plan, int64_t from, int64_t* next);

/* Data-parallel training (trainer.py: RCCL all-reduce of the flat gradient under the backward pass): `cus` compute units are set
 * aside for the communication kernel that stays resident meanwhile.  Launches whose blocks wait for each other (the single-launch
 * BatchNorm kernels, convolution + BatchNorm in one launch) size themselves against occupancy x (CUs - cus); shapes that no longer
 * fit take their multi-launch form.  Default 0, or PIXELPICK_COMM_CU_RESERVE at load time.  Changes the launch plans: set it before
 * the first step. */
void pp_set_comm_cu_reserve(int cus);
int pp_get_comm_cu_reserve(void);
/* Profiling hook for bench.py: `starts`/`stops` are HOST arrays of n caller-created hipEvent_t.  The
 * i-th launch of a dominant kernel (acq_kernel, conv_igemm_kernel) after this call records starts[i] / stops[i] on its
 * stream immediately before / after the launch.  Pass (NULL, NULL, 0) to switch off.  The arrays must
 * stay alive until then.  Process-global and meant for ONE measuring thread (the only state besides pp_set_comm_cu_reserve that a call
 * can leave behind in this library). */
void pp_set_kernel_events(void** starts, void** stops, int n);

#ifdef __cplusplus
}
#endif
#endif /* PIXELPICK_HIP_H */
