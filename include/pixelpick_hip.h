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
    PP_ERR_UNSUPPORTED = -4,  /* shape outside the supported range (e.g. C > PP_ACQ_MAX_CLASSES) */
    PP_ERR_LAUNCH = -5        /* hipLaunch failure (message holds hipGetErrorString) */
};

/* query.py:229-239 UncertaintySampler strategies.  (`random`, query.py:242-244, is host RNG.) */
enum { PP_ACQ_ENTROPY = 0, PP_ACQ_LEAST_CONFIDENCE = 1, PP_ACQ_MARGIN = 2 };

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
 *   out_map  f32 [B,H,W] or NULL — the score map after exclusion (what query.py calls uc_map).
 *   k larger than the number of un-excluded pixels is NOT an error (as in the reference, excluded
 *   pixels are then returned, lowest index first).
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

/* Debug/bench knobs.
 * reduce mode: 0 = threshold-prefiltered per-wave top-k with DPP reductions (default), 1 = same with
 *              ds_bpermute (__shfl) reductions, 2 = plain k-round extraction loop (no prefilter).
 * exact formula: 0 = default scorer (entropy = log S + sum e_c (m - x_c) / S, v_exp_f32 exponentials;
 *              identical NaN behaviour), 1 = the reference's operation order p_c = exp(x_c - m) / S,
 *              sum(-p_c log p_c) with libm-accurate exp/log (query.py:190,230). */
void pp_debug_set_reduce_mode(int mode);
void pp_debug_set_exact_formula(int on);
/* Tuning knob for the C == 19 flat path: occupancy bound (2/3/4 waves per SIMD, 0 = default) and
 * pixels per thread (4/8/16, 0 = automatic). */
void pp_debug_set_acq_tuning(int occ, int ppt);

/* Profiling hook for bench.py: `starts`/`stops` are HOST arrays of n caller-created hipEvent_t.  The
 * i-th launch of the dominant acquisition kernel after this call records starts[i] / stops[i] on its
 * stream immediately before / after the launch.  Pass (NULL, NULL, 0) to switch off.  The arrays must
 * stay alive until then. */
void pp_debug_set_kernel_events(void** starts, void** stops, int n);

#ifdef __cplusplus
}
#endif
#endif /* PIXELPICK_HIP_H */
