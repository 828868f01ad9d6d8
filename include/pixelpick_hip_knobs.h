/*
 * pixelpick_hip_knobs.h - the TEST BUILD's planner switches (libpixelpick_hip_knobs.so = the library's sources + -DPP_DEBUG_KNOBS).
 * Process-global state for A/B measurements and for forcing a kernel form in parity tests; none of it is exported by the release library
 * ("holds no global mutable state").  Included by pixelpick_hip.h when PP_DEBUG_KNOBS is defined.
 */
#ifndef PIXELPICK_HIP_KNOBS_H
#define PIXELPICK_HIP_KNOBS_H

/* Knobs.
 * reduce mode: 0 = threshold-prefiltered per-wave top-k with DPP reductions (default), 1 = same with
 *              ds_bpermute (__shfl) reductions, 2 = plain k-round extraction loop (no prefilter).
 *              bit 8: large-k selection through the one-block radix select, bit 9: no quantised-histogram select,
 *              bit 10: the score histogram of the large-k selection in its own pass over the map instead of inside the scorer launch (A/B).
 *              bit 11: large k without a caller's map through the map-writing scorer + map select instead of the sampled threshold +
 *              candidate emission + list select (A/B); bits 12-17: the sample's aim in sixteenths of k (0 = 40: 2.5 k pixels pass);
 *              bits 18-19: 128 / 64 / 256 / 512 sample locations per image; bits 20-21: 16 / 8 / 4 / 32 pixels per location.
 * exact formula: 0 = default scorer (entropy = log S + sum e_c (m - x_c) / S, v_exp_f32 exponentials;
 *              identical NaN behaviour), 1 = the reference's operation order p_c = exp(x_c - m) / S,
 *              sum(-p_c log p_c) with libm-accurate exp/log (query.py:190,230). */
void pp_debug_set_reduce_mode(int mode);
void pp_debug_set_exact_formula(int on);
/* Tuning knob for the C == 19 flat path: occupancy bound (2/3/4 waves per SIMD, 0 = default) and
 * pixels per thread (4/8, 0 = automatic).  occ 8 / 9: synchronous NHWC kernel / generic strided path; occ 10: the streamed scorers
 * (class vector read from memory in passes) at ANY class count - the tests compare them bit for bit with the register kernels. */
void pp_debug_set_acq_tuning(int occ, int ppt);
/* Dense conv kernel A/B knobs: low 2 bits 0 = 128x128 large tile (default; measured fastest), 2 = 128x64 tiles;
 * bit 2 = linear instead of XCD-aware tile order; bit 3 = conditional (non-vector) loads; bits 4/5 = cap the large
 * tile at 2 / 1 blocks per CU; bit 6 split-K off; bit 7 64-deep K step of the 64x64 tiles off; bit 8 LDS-DMA kernel of the
 * 128-row tiles off (bit 15: backward-data only for the 128x64-tiled layers; bit 22: forward only); bits 9-14 weight-gradient / ragged-tile / K-order variants;
 * bit 12 32-deep K step for the 128x128 tiles; bits 16/17 TIMING-ONLY ablation (skips the split-K reduce: wrong results);
 * bit 18 LDS-DMA kernel of the 64x64 tiles off (bit 19: forward only); bit 20 LDS-DMA weight-gradient kernel of the
 * 128-wide tiles off; bit 21 LDS-DMA weight-gradient kernel for the 64x64 tiles on.
 * Findings: profiles/r01_conv_ablation.txt. */
void pp_debug_set_dw_variant(int v);   /* bit 0: one-output-per-thread depthwise kernels (A/B); bit 8: separable bilinear backward off;
                                        * bits 13-15 / 16-17: column-block width / least rows per thread of the depthwise weight gradient (0 = by map size) */
void pp_debug_set_splitk(int v);       /* tiles_threshold | target_blocks << 10 | min_k_steps << 20 | min_steps_per_slice << 26 */
void pp_debug_set_wgrad_target(int blocks);   /* split-M target of the weight-gradient kernels (default 1024) */
void pp_debug_set_bn_target(int blocks);   /* strips x row chunks of the single-launch BatchNorm (default 384, <= 1024) */
void pp_debug_set_bn_bytes_per_block(int bytes);   /* large maps: one block per this many bytes (default 0 = off: measured neutral); -1: register-cached variants off */
/* Debug: pp_bn_train_fwd_fused writes per-block wall-clock stamps (100 MHz; [blocks][8]: entry, statistics pass done, block
 * reduction done, partial published, strip combined, rows written) into this device buffer; NULL (default) = off. */
void pp_debug_set_bn_probe(void* device_buffer);
void pp_debug_set_conv_thresholds(int v);   /* big_tile_min | wgrad_rows_min << 12 (defaults 384 / 128) */
void pp_debug_conv_plan(int64_t M, int Cn, int Ck, int ntaps, int* out4);   /* tile rows, tile cols, tiles, split-K slices */
void pp_debug_set_conv_rows(int bits);      /* whole-row VALU kernels of the narrow pointwise layers: bit 0 off, bit 1 forward rows kernel only from 65536 rows (A/B) */
void pp_debug_set_conv_bn_fuse(int bits);   /* fused conv + BatchNorm launches offered: bit 0 tiled fwd, 1 split-K fwd, 2 bwd 64x64, 3 bwd split-K / 128x32 (default 15; A/B) */
void pp_debug_set_x3_variant(int v);   /* experiment forms of conv_x3_kernel<256,128> (ring depth, priority, DMA placement, timing ablations); 0 = product */
void pp_debug_set_gemm_pw(int v);   /* pointwise GEMM kernel (gemm_pw.hip): low 4 bits 0 off, 1 rule (default), 2..7 force tile form 0..5; bits 4..: least rows (default 4096) */
void pp_debug_set_x3f(int v);   /* in-kernel activation split (conv_x3f.hip, an experiment that lives in the test build only - measured slower): bit 0 ON; bits 1-3 least GFLOP of a non-classic layer {1, 0.5, 2, 4, 8, 0.25, 0}; bits 4-5 least tiles {128, 64, 192, 256}; bits 6-7 least K {256, 128, 512, 16} (A/B) */
void pp_debug_set_x3(int on);   /* large-tile conv layers: 1 = bf16x3-split MFMA kernel (default), 0 = fp32 MFMA kernels (A/B, parity) */
void pp_debug_set_conv_variant(int v);


/* Test stand-in for such a resident kernel: `blocks` (<= 256) blocks that each take a whole CU's LDS and spin until *stop != 0 (a
 * host-visible int) or max_ticks of the 100 MHz clock (<= 60 s) have passed; *started counts the blocks that got a CU. */
int pp_debug_occupy_cus(int blocks, const int* stop, uint64_t max_ticks, uint64_t* started, pp_stream_t stream);


#endif /* PIXELPICK_HIP_KNOBS_H */
