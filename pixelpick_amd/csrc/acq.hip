// acq.hip — fused acquisition for PixelPick on gfx950 (MI355X).
//
// Replaces, for a batch of images and in ONE read of the logits:
//   query.py:190      prob = F.softmax(model(x)["pred"][:, :, :h, :w], dim=1)
//   query.py:229-239  UncertaintySampler._entropy / _least_confidence / _margin_sampling
//   query.py:195-201  uc_map[mask] = fill ; uc_map[mask_void] = fill
//   query.py:57-61    uc_map.flatten().topk(k, largest).indices
//
// Bandwidth-bound integer/float streaming work: no MFMA.  Each lane keeps the C class logits of its
// pixels in registers (softmax needs no second memory pass), planes are read as 16 B/lane coalesced
// loads, per-wave top-k candidates are extracted with DPP reductions and merged by a tiny second
// kernel, so the logits are read exactly once and nothing full-size is written unless asked for.
#include <algorithm>
#include <cmath>

#include "pp_common.h"
#include <type_traits>

namespace pp {

static int g_reduce_mode = 0;
static int g_exact_formula = 0;   // 1: reference operation order (19 exp + 19 div + 19 log per pixel) as the process default (test build's knob)
// The same choice PER CALL: `strategy | PP_ACQ_REFERENCE_ORDER` at an entry point.  The flag lives in a thread-local for the duration of
// that call (the planners below read it where they used to read the global), so concurrent callers on other threads / streams do not see it.
static thread_local int t_exact_call = 0;
static inline int exact_formula() { return t_exact_call | g_exact_formula; }
struct ExactScope {
    int keep;
    explicit ExactScope(int& strategy) : keep(t_exact_call)
    {
        if (strategy >= 0 && (strategy & PP_ACQ_REFERENCE_ORDER)) { t_exact_call = 1; strategy &= ~PP_ACQ_REFERENCE_ORDER; }
    }
    ~ExactScope() { t_exact_call = keep; }
};
static int g_tune_occ = 0;        // tuning knobs (C == 19 flat path only): waves/SIMD bound, pixels per thread
static int g_tune_ppt = 0;
static int g_acq_strat_spec = 1;   // strategy-specialised scorers of the three dataset class counts (pp_debug_set_acq_tuning bit 10 of `occ`: off)
static int g_nt_off = 0;          // A/B: ordinary instead of non-temporal logit loads in the strategy-specialised scorers (pp_debug_set_acq_tuning bit 11 of `occ`)
static int g_tune_xcd = 0;        // 0: by plane size, 1: never, 2: always (pp_debug_set_acq_tuning bits 8-9 of `occ`)

typedef float f32x4_nt __attribute__((ext_vector_type(4)));
typedef float f32x2_nt __attribute__((ext_vector_type(2)));      // (native vector type: __builtin_nontemporal_load does not take HIP's float4 struct)

constexpr int kBlock = 256;
constexpr int kSmallKMax = 48;        // fused per-wave extraction up to this k (measured: 0.74/0.70/0.62 of HBM at k=20/32/48, 0.37 at 64); beyond: map + radix select
constexpr int kMergeItems = 16;       // merge kernel: candidates per thread
constexpr int kMergeChunk = kBlock * kMergeItems;
constexpr int kLargeThreads = 1024;
constexpr int kLargeLdsMaxP = 16384;  // 128 KiB of u64 sort keys in LDS

struct AcqParams {
    const float* logits;
    const uint8_t* exclude;
    float* out_map;      // [B,N] or null
    uint64_t* cand;      // [B, waves_per_image, k] or null
    int64_t sB, sC, sH, sW;
    int C, W;
    int64_t N;           // H*W
    int blocks_per_image;
    int k;
    int strategy;
    int reduce_mode;
    int from_prob;       // 1: input already holds probabilities (UncertaintySampler.__call__, query.py:246-247)
    uint32_t* qhist = nullptr;   // HIST kernels: [B][kQBins] bin counts of the scores written to out_map (zeroed by the host), qscale bins per unit score
    float qscale = 0.0f;
    int xcd_per = 0;     // > 0: XCD-contiguous block order (acq_kernel only): hardware block b works on logical block
    int nb = 0;          //      (b % 8) * xcd_per + b / 8 of nb, so that the blocks one XCD holds walk ONE contiguous eighth of the launch
    // EMIT kernels (large-k selection without the score map): every wave appends the (key, index) words of its pixels at or beyond the image's
    // sampled threshold key to its OWN fixed segment of the image's list and stores how many it wrote - no atomics, no block-level exchange
    const uint32_t* tkey = nullptr;   // [B] threshold order key (acq_sample_thr_kernel)
    uint64_t* elist = nullptr;        // [B][eimg]; wave (blk, w) owns entries [(blk * 4 + w) * PPT * 64, +PPT * 64)
    uint32_t* ecnt = nullptr;         // [B][nseg]
    int64_t eimg = 0;
    int nseg = 0;
    int segst = 0;                    // entries from one segment to the next: PPT * 64 + kSegPad (not a multiple of 4 KB: see emit_list_entries)
    int sample_locs = 0, sample_gpl = 0;   // acq_sample_thr_kernel: locations per image, 4-pixel groups per location
};

// ---- per-pixel score ----------------------------------------------------------------------------
// Same operation order as the reference: p_c = exp(x_c - m) / S ; then the strategy's formula.
template <int CMAX, bool EXACT>
__device__ __forceinline__ float pixel_score(const float (&x)[CMAX], int C, int strategy, int from_prob)
{
    // (-p) * log p and the running sum are rounded separately, as torch and the oracle do (hipcc contracts a * b + c by default, and
    // whether it does depends on how the loop was unrolled: the HIP __fmul_rn / __fadd_rn are plain operators and do not stop it)
#pragma clang fp contract(off)
    if (from_prob) {  // x is prob: the UncertaintySampler formulas verbatim
        if (strategy == PP_ACQ_ENTROPY) {
            float acc = 0.0f;
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (EXACT || c < C) acc += (-x[c]) * logf(x[c]);
            return acc;
        }
        float t1 = -INFINITY, t2 = -INFINITY;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (EXACT || c < C) {
                t2 = fmaxf(t2, fminf(t1, x[c]));
                t1 = fmaxf(t1, x[c]);
            }
        return strategy == PP_ACQ_LEAST_CONFIDENCE ? 1.0f - t1 : fabsf(t1 - t2);
    }
    float m = x[0];
#pragma unroll
    for (int c = 1; c < CMAX; ++c)
        if (EXACT || c < C) m = fmaxf(m, x[c]);
    float e[CMAX];
    float S = 0.0f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
        if (EXACT || c < C) {
            e[c] = expf(x[c] - m);
            S += e[c];
        }
    if (strategy == PP_ACQ_ENTROPY) {  // query.py:230  (-p*log p).sum(dim=1); 0*log 0 -> NaN as reference
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (EXACT || c < C) {
                float p = e[c] / S;
                acc += (-p) * logf(p);
            }
        return acc;
    } else if (strategy == PP_ACQ_LEAST_CONFIDENCE) {  // query.py:234  1 - max_c p ; max e == exp(0) == 1
        return 1.0f - 1.0f / S;
    } else {  // query.py:238-239  |top1 - top2| of p ; division is monotone so top-2 of e give top-2 of p
        float t1 = -INFINITY, t2 = -INFINITY;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (EXACT || c < C) {
                t2 = fmaxf(t2, fminf(t1, e[c]));
                t1 = fmaxf(t1, e[c]);
            }
        return fabsf(t1 / S - t2 / S);
    }
}

// Default scorer: algebraically identical, ~5x fewer VALU slots.  With d_c = x_c - m, e_c = exp(d_c):
//   entropy = -sum p_c log p_c = log S + (sum e_c (m - x_c)) / S      (both terms >= 0: no cancellation)
//   least-confidence = 1 - 1/S ;  margin = |1/S - exp(x_(2) - m)/S|
// e_c uses v_exp_f32 on d*log2(e): absolute error <= ~1e-7 on every term (terms are <= 1), the same
// class as the ulp differences between libm implementations.  The reference's 0*log 0 = NaN behaviour
// (query.py:230) is kept exactly: NaN iff the smallest p_c = exp(x_min - m)/S rounds to 0.
__device__ __forceinline__ float fast_exp(float d) { return __builtin_amdgcn_exp2f(d * 1.44269504088896340736f); }

// STRAT >= 0: the strategy is a compile-time constant - the chains the other two strategies need (second maximum: margin only;
// minimum and the e * d sum: entropy only) are not computed at all (fewer VALU slots and registers: C = 21 fits three waves per SIMD).
template <int CMAX, bool EXACT, int STRAT = -1>
__device__ __forceinline__ float pixel_score_fast(const float (&x)[CMAX], int C, int strategy_rt)
{
    const int strategy = STRAT >= 0 ? STRAT : strategy_rt;
    constexpr bool kX2 = STRAT < 0 || STRAT == PP_ACQ_MARGIN, kEnt = STRAT < 0 || STRAT == PP_ACQ_ENTROPY;
    float m = x[0], x2 = -INFINITY, xmin = x[0];
#pragma unroll
    for (int c = 1; c < CMAX; ++c)
        if (EXACT || c < C) {
            if (kX2) x2 = fmaxf(x2, fminf(m, x[c]));
            m = fmaxf(m, x[c]);
            if (kEnt) xmin = fminf(xmin, x[c]);
        }
    float S = 0.0f, T = 0.0f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
        if (EXACT || c < C) {
            const float d = x[c] - m;
            const float e = fast_exp(d);
            S += e;
            if (kEnt) T = fmaf(e, -d, T);
        }
    if (strategy == PP_ACQ_ENTROPY) {
        float ent = logf(S) + T / S;
        if (xmin - m < -87.0f) {                    // rare: possible underflow of the smallest probability
            if (expf(xmin - m) / S == 0.0f) ent = __uint_as_float(0x7FC00000u);
        }
        return ent;
    } else if (strategy == PP_ACQ_LEAST_CONFIDENCE) {
        return 1.0f - 1.0f / S;
    } else {
        return fabsf(1.0f / S - expf(x2 - m) / S);
    }
}

// ---- per-wave top-k extraction -------------------------------------------------------------------
// Each lane holds PPT (key, ~idx) pairs; k rounds of {lane-local max, two DPP wave reductions,
// knock out the winner}.  Winner order == global order (key desc, index asc).  Lane 0 stores.
template <int PPT>
__device__ __forceinline__ void wave_extract_topk(uint32_t (&kh)[PPT], uint32_t (&kl)[PPT], int k,
                                                  uint64_t* dst, int mode)
{
    const int lane = threadIdx.x & (kWave - 1);
    for (int r = 0; r < k; ++r) {
        uint32_t lh = 0;
#pragma unroll
        for (int j = 0; j < PPT; ++j) lh = kh[j] > lh ? kh[j] : lh;
        uint32_t ll = 0;
#pragma unroll
        for (int j = 0; j < PPT; ++j) ll = (kh[j] == lh && kl[j] > ll) ? kl[j] : ll;
        const uint32_t mh = wave_umax(lh, mode);
        const uint32_t ml = wave_umax(lh == mh ? ll : 0u, mode);
        if (lane == 0) dst[r] = mh == 0u ? 0ull : (((uint64_t)mh << 32) | ml);
        if (mh == 0u) {  // exhausted (wave-uniform)
            for (int q = r + 1; q < k; ++q)
                if (lane == 0) dst[q] = 0ull;
            break;
        }
#pragma unroll
        for (int j = 0; j < PPT; ++j)
            if (kh[j] == mh && kl[j] == ml) kh[j] = 0u;
    }
}

// Threshold-prefiltered variant (same result).  tau = k-th largest of the 64 lane-local maxima (ballot
// bit search, mostly SALU); the wave's top-k all have key >= tau, and typically only ~k..2k keys survive.
// Survivors go to a per-wave LDS list, each is ranked against the list (broadcast reads) and written
// straight to its sorted slot.  Falls back to the exact loop when ties blow the list up (e.g. a wave
// that sees only excluded pixels).
constexpr int kSurvCap = 256;   // 4 survivors per lane; k up to 64 expects ~k..2k survivors

template <int PPT>
__device__ __forceinline__ void wave_extract_topk_prefilter(uint32_t (&kh)[PPT], uint32_t (&kl)[PPT], int k,
                                                            uint64_t* dst, int mode, volatile uint64_t* sbuf,
                                                            volatile uint32_t* scnt)
{
    const int lane = threadIdx.x & (kWave - 1);
    uint32_t lm = 0;
#pragma unroll
    for (int j = 0; j < PPT; ++j) lm = kh[j] > lm ? kh[j] : lm;
    uint32_t tau = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t t = tau | (1u << bit);
        if (__popcll(__ballot(lm >= t)) >= k) tau = t;
    }
    if (tau == 0u) tau = 1u;  // fewer than k lanes hold a valid key: every valid key survives
    if (lane == 0) *scnt = 0u;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < PPT; ++j)
        if (kh[j] >= tau) {
            const uint32_t pos = atomicAdd(const_cast<uint32_t*>(scnt), 1u);
            if (pos < (uint32_t)kSurvCap) sbuf[pos] = ((uint64_t)kh[j] << 32) | kl[j];
        }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t total = *scnt;
    if (total > (uint32_t)kSurvCap) {  // wave-uniform
        wave_extract_topk<PPT>(kh, kl, k, dst, mode);
        return;
    }
    constexpr int SPL = kSurvCap / kWave;          // survivors per lane
    uint64_t sv[SPL];
    int rk[SPL];
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        sv[j] = (uint32_t)(lane + j * kWave) < total ? sbuf[lane + j * kWave] : 0ull;
        rk[j] = 0;
    }
    if (total <= (uint32_t)kWave) {                 // common case (k ~ 20): one survivor per lane at most
        for (uint32_t i = 0; i < total; ++i) rk[0] += sbuf[i] > sv[0];
    } else {
        for (uint32_t i = 0; i < total; ++i) {
            const uint64_t v = sbuf[i];
#pragma unroll
            for (int j = 0; j < SPL; ++j) rk[j] += v > sv[j];
        }
    }
#pragma unroll
    for (int j = 0; j < SPL; ++j)
        if (sv[j] != 0ull && rk[j] < k) dst[rk[j]] = sv[j];
    for (int r = (int)total + lane; r < k; r += kWave) dst[r] = 0ull;
}

// ---- per-BLOCK candidate list --------------------------------------------------------------------------
// Every wave extracts its own sorted top-k into LDS; wave 0 then merges the block's NW lists by rank (keys are unique - the pixel
// index sits in the low word - so "how many of the NW*k candidates are greater" IS the slot) and ONE list of k leaves the block.
// A quarter of the candidate traffic of per-wave lists, and at 256x512 (64 blocks per image -> 1280 candidates) the merge below needs
// one level instead of two.  s_top: NW * kSmallKMax keys.  dst == nullptr: the merged list stays in s_top[0..k) (cand_merge_kernel).
template <int PPT>
__device__ __forceinline__ void block_emit_topk(uint32_t (&kh)[PPT], uint32_t (&kl)[PPT], int k, uint64_t* dst, int mode,
                                                uint64_t (*s_surv)[kSurvCap], uint32_t* s_cnt, uint64_t* s_top)
{
    constexpr int NW = kBlock / kWave;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
    if (mode == 2) wave_extract_topk<PPT>(kh, kl, k, s_top + wave * k, 0);
    else wave_extract_topk_prefilter<PPT>(kh, kl, k, s_top + wave * k, mode, s_surv[wave], &s_cnt[wave]);
    __syncthreads();
    if (wave != 0) return;
    const int n = NW * k;                                  // <= 192
    constexpr int J = (NW * kSmallKMax + kWave - 1) / kWave;
    uint64_t mine[J];
    int rk[J];
    int nvalid = 0;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int i = lane + j * kWave;
        mine[j] = i < n ? s_top[i] : 0ull;
        rk[j] = 0;
        nvalid += __popcll(__ballot(mine[j] != 0ull));
    }
    if (n <= kWave) {
        for (int i = 0; i < n; ++i) rk[0] += s_top[i] > mine[0];
    } else {
        for (int i = 0; i < n; ++i) {
            const uint64_t v = s_top[i];
#pragma unroll
            for (int j = 0; j < J; ++j) rk[j] += v > mine[j];
        }
    }
    __builtin_amdgcn_wave_barrier();                       // dst may be s_top itself: every lane has finished reading
#pragma unroll
    for (int j = 0; j < J; ++j)
        if (mine[j] != 0ull && rk[j] < k) dst[rk[j]] = mine[j];
    for (int r = nvalid + lane; r < k; r += kWave) dst[r] = 0ull;
}

// ---- quantised score bins (the large-k selection's threshold search, see select_qhist_kernel) ----------------------------------
constexpr int kQBins = 1024;
__device__ __forceinline__ uint32_t qbin(float s, bool lg, float scale)
{
    if (s != s) return lg ? (uint32_t)(kQBins - 1) : 0u;             // NaN: first for largest, last for smallest (order_key's policy)
    const float t = s * scale;
    const int qi = t >= (float)(kQBins - 1) ? kQBins - 1 : (t > 0.0f ? (int)t : 0);
    return (uint32_t)(lg ? qi : kQBins - 1 - qi);
}

// ---- main kernel ---------------------------------------------------------------------------------
// VEC == 4: planes are flat & 16-B aligned (sW == 1, sH == W, N % 4 == 0): float4 per class plane.
// VEC == 1: arbitrary element strides (NHWC views, cropped views), one pixel per load.
// A block covers kBlock*VEC*G consecutive pixels of ONE image.
// MATH: 0 = default scorer, 1 = reference operation order, 2 = input already holds probabilities.
// HIST (map-writing launches of the large-k selection, VEC == 4): the block also counts its scores into the image's kQBins-bin histogram
// (LDS bins in the survivor lists' storage - no candidates are extracted in that mode - then one global atomic per non-empty bin): what
// select_qhist_kernel did in a second pass over the map.
template <int CMAX, bool EXACT, int VEC, int G, int MATH, int OCC = 3, int STRAT = -1, bool HIST = false, bool NT = true, bool EMIT = false>
__global__ __launch_bounds__(kBlock, OCC) void acq_kernel(AcqParams p)
{
    constexpr int PPT = VEC * G;
    static_assert(!HIST || VEC == 4, "histogram epilogue: the flat float4 form");
    static_assert(!EMIT || (VEC == 4 && !HIST), "candidate emission: the flat float4 form");
    __shared__ uint64_t s_surv[kBlock / kWave][kSurvCap];
    static_assert(sizeof(uint64_t) * (kBlock / kWave) * kSurvCap >= sizeof(uint32_t) * kQBins, "the bins live in the survivor lists");
    uint32_t* lh = reinterpret_cast<uint32_t*>(&s_surv[0][0]);
    if constexpr (HIST) {
        for (int i = threadIdx.x; i < kQBins; i += kBlock) lh[i] = 0u;
        __syncthreads();
    }
    __shared__ uint32_t s_cnt[kBlock / kWave];
    __shared__ uint64_t s_top[(kBlock / kWave) * kSmallKMax];
    int lb = blockIdx.x;
    if (p.xcd_per) {
        // Large class planes (>= 4 MB: 1024 x 2048 logits): a block's C streams lie a whole plane apart, and with blocks dealt
        // round-robin to the eight XCDs every XCD walks every plane end to end.  Giving each XCD one contiguous eighth of the
        // launch lifts a pure read of this layout from 0.65 to 0.74 of 8 TB/s (tools/probe/plane_read.hip, profiles/r04_acq_layout.txt).
        lb = (int)(blockIdx.x & 7u) * p.xcd_per + (int)(blockIdx.x >> 3);
        if (lb >= p.nb) return;
    }
    const int img = lb / p.blocks_per_image;
    const int blk = lb - img * p.blocks_per_image;
    const int tid = threadIdx.x;
    const bool largest = p.strategy != PP_ACQ_MARGIN;
    const float fill = largest ? 0.0f : 1.0f;
    const float* base = p.logits + (int64_t)img * p.sB;
    const uint8_t* excl = p.exclude ? p.exclude + (int64_t)img * p.N : nullptr;
    float* omap = p.out_map ? p.out_map + (int64_t)img * p.N : nullptr;

    uint32_t kh[PPT], kl[PPT];

#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int64_t pix0 = ((int64_t)blk * G + g) * (kBlock * VEC) + (int64_t)tid * VEC;
        float s[VEC];
        if (pix0 < p.N) {
            if constexpr (VEC == 4) {
                float x[4][CMAX];
#pragma unroll
                for (int c = 0; c < CMAX; ++c)
                    if (EXACT || c < p.C) {
                        if constexpr (NT) {
                            // the logits are read exactly once: non-temporal loads (no L2 / MALL allocation for a 2.6 GB stream that
                            // nobody reads again) - measured 0.420-0.425 -> 0.392-0.396 ms per launch at B=256 x 256x512x19, i.e. 0.76 ->
                            // 0.82 of 8 TB/s; pp_debug_set_acq_tuning bit 11 selects the ordinary loads for A/B
                            const f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(base + (int64_t)c * p.sC + pix0));
                            x[0][c] = v.x; x[1][c] = v.y; x[2][c] = v.z; x[3][c] = v.w;
                        } else {
                        const float4 v = *reinterpret_cast<const float4*>(base + (int64_t)c * p.sC + pix0);
                        x[0][c] = v.x; x[1][c] = v.y; x[2][c] = v.z; x[3][c] = v.w;
                        }
                    }
                uint32_t ex = excl ? *reinterpret_cast<const uint32_t*>(excl + pix0) : 0u;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    if constexpr (MATH == 0) s[v] = pixel_score_fast<CMAX, EXACT, STRAT>(x[v], p.C, p.strategy);
                    else s[v] = pixel_score<CMAX, EXACT>(x[v], p.C, p.strategy, MATH == 2);
                    if ((ex >> (8 * v)) & 0xFFu) s[v] = fill;
                    if constexpr (HIST) atomicAdd(&lh[qbin(s[v], largest, p.qscale)], 1u);
                    __builtin_amdgcn_sched_barrier(0);  // one pixel's temporaries at a time (VGPR budget)
                }
                if (omap) *reinterpret_cast<float4*>(omap + pix0) = make_float4(s[0], s[1], s[2], s[3]);
            } else {
                const int64_t h = pix0 / p.W, w = pix0 - h * p.W;
                const float* px = base + h * p.sH + w * p.sW;
                float x[CMAX];
#pragma unroll
                for (int c = 0; c < CMAX; ++c)
                    if (EXACT || c < p.C) x[c] = NT ? __builtin_nontemporal_load(px + (int64_t)c * p.sC) : px[(int64_t)c * p.sC];
                if constexpr (MATH == 0) s[0] = pixel_score_fast<CMAX, EXACT>(x, p.C, p.strategy);
                else s[0] = pixel_score<CMAX, EXACT>(x, p.C, p.strategy, MATH == 2);
                if (excl && excl[pix0]) s[0] = fill;
                if (omap) omap[pix0] = s[0];
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                kh[g * VEC + v] = order_key(s[v], largest);
                kl[g * VEC + v] = 0xFFFFFFFFu - (uint32_t)(pix0 + v);
            }
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) { kh[g * VEC + v] = 0u; kl[g * VEC + v] = 0u; }
        }
        // keep one group's class vector live at a time (occupancy hides the HBM latency, not hoisted loads)
        __builtin_amdgcn_sched_barrier(0);
    }

    if constexpr (EMIT) {
        // keys at or beyond the image's threshold key form an upper set of the (key, index) order: if the image's lists hold >= k words
        // the k picks are among them (topk_lsel_kernel checks that).  Ballot order inside the segment; the select ranks exactly.
        const uint32_t tk = p.tkey[img];
        const int wave = tid >> 6, lane = tid & (kWave - 1);
        const int seg = blk * (kBlock / kWave) + wave;
        uint64_t* dst = p.elist + (int64_t)img * p.eimg + (int64_t)seg * p.segst;
        const uint64_t below = (1ull << lane) - 1ull;
        uint32_t n = 0;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const bool sv = kh[j] >= tk;                 // (pixels past the end of the image carry key 0, tk >= 1)
            const uint64_t bal = __ballot(sv);
            if (sv) dst[n + (uint32_t)__popcll(bal & below)] = ((uint64_t)kh[j] << 32) | kl[j];
            n += (uint32_t)__popcll(bal);
        }
        if (lane == 0) p.ecnt[(int64_t)img * p.nseg + seg] = n;
        return;
    }
    if constexpr (HIST) {
        __syncthreads();
        uint32_t* Hg = p.qhist + (int64_t)img * kQBins;
        for (int d = threadIdx.x; d < kQBins; d += kBlock) {
            const uint32_t c = lh[d];
            if (c) atomicAdd(&Hg[d], c);
        }
        return;
    }
    if (p.cand)
        block_emit_topk<PPT>(kh, kl, p.k, p.cand + ((int64_t)img * p.blocks_per_image + blk) * p.k, p.reduce_mode, s_surv, s_cnt, s_top);
}


// ---- dense NHWC (channels-last) input ----------------------------------------------------------------
// Pixel-major storage: one pixel's C logits are contiguous (76 B at C=19).  A per-lane strided read touches 38
// cache lines per wave-instruction; instead the block streams its pixel range as flat coalesced float4 loads
// into LDS and every lane then reads its own pixel's class vector back (stride C floats: conflict-free for odd C).
// Measured (B=256 x 256x512x19, entropy, k=20): 4.02 TB/s vs 3.17 TB/s for the per-lane strided path and
// 5.98 TB/s for NCHW planes; issuing the next group's loads into registers ahead of the compute was slower (2.6).
template <int CMAX, bool EXACT, int G, int MATH>
__global__ __launch_bounds__(kBlock, 3) void acq_nhwc_kernel(AcqParams p)
{
    __shared__ __attribute__((aligned(16))) float s_x[kBlock * CMAX];
    __shared__ uint64_t s_surv[kBlock / kWave][kSurvCap];
    __shared__ uint32_t s_cnt[kBlock / kWave];
    __shared__ uint64_t s_top[(kBlock / kWave) * kSmallKMax];
    const int img = blockIdx.x / p.blocks_per_image;
    const int blk = blockIdx.x - img * p.blocks_per_image;
    const int tid = threadIdx.x;
    const bool largest = p.strategy != PP_ACQ_MARGIN;
    const float fill = largest ? 0.0f : 1.0f;
    const int C = EXACT ? CMAX : p.C;
    const float* base = p.logits + (int64_t)img * p.sB;
    const uint8_t* excl = p.exclude ? p.exclude + (int64_t)img * p.N : nullptr;
    float* omap = p.out_map ? p.out_map + (int64_t)img * p.N : nullptr;
    uint32_t kh[G], kl[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int64_t pix_blk = ((int64_t)blk * G + g) * kBlock;          // first pixel of this group
        const int64_t npix = p.N - pix_blk < kBlock ? p.N - pix_blk : kBlock;
        if (npix > 0) {
            const int64_t nfl = npix * C;                                 // floats to stage (multiple of 4: N % 4 == 0)
            const float4* src = reinterpret_cast<const float4*>(base + pix_blk * C);
            for (int64_t i = tid; i < nfl / 4; i += kBlock)         // (read once: non-temporal, as acq_kernel)
                reinterpret_cast<f32x4_nt*>(s_x)[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(src) + i);
        }
        __syncthreads();
        const int64_t pix = pix_blk + tid;
        if (pix < p.N) {
            float x[CMAX];
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (EXACT || c < C) x[c] = s_x[tid * C + c];
            float sc;
            if constexpr (MATH == 0) sc = pixel_score_fast<CMAX, EXACT>(x, p.C, p.strategy);
            else sc = pixel_score<CMAX, EXACT>(x, p.C, p.strategy, MATH == 2);
            if (excl && excl[pix]) sc = fill;
            if (omap) omap[pix] = sc;
            kh[g] = order_key(sc, largest);
            kl[g] = 0xFFFFFFFFu - (uint32_t)pix;
        } else {
            kh[g] = 0u; kl[g] = 0u;
        }
        __syncthreads();
    }
    if (p.cand)
        block_emit_topk<G>(kh, kl, p.k, p.cand + ((int64_t)img * p.blocks_per_image + blk) * p.k, p.reduce_mode, s_surv, s_cnt, s_top);
}

// ---- dense NHWC, asynchronous variant ------------------------------------------------------------------
// Same pixel -> wave mapping and the same results as acq_nhwc_kernel, but every WAVE streams its own 64-pixel slices
// into a private, double-buffered LDS slot with LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction, no
// staging registers, no ds_write pass) and keeps the NEXT slice in flight while it scores the current one: the only
// synchronisation is the issuing wave's counted `s_waitcnt vmcnt(N)` (MI355X_MICROARCH.md item 7) - no barrier at all.
// The DMA pieces are written in inline asm so that hipcc's own s_waitcnt bookkeeping (which would drain vmcnt(0) in
// front of the first LDS read) does not see them.
__device__ __forceinline__ void glds16(const float* gsrc, uint32_t lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds1(const uint8_t* gsrc, uint32_t lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_ubyte %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int CMAX, int G>
__global__ __launch_bounds__(kBlock, (CMAX > 19 ? 2 : 3)) void acq_nhwc_dma_kernel(AcqParams p)
{
    constexpr int NW = kBlock / kWave;
    constexpr int NCH = (CMAX * 256 + 1023) / 1024;      // 1-KiB DMA pieces per 64-pixel slice (5 at C = 19)
    constexpr int SLOT = NCH * 256;                      // floats per slot (the last piece may overhang the slice)
    __shared__ __attribute__((aligned(1024))) float s_x[NW][2][SLOT];
    __shared__ __attribute__((aligned(256))) uint32_t s_e[NW][2][kWave];   // sub-dword LDS-DMA lands one DWORD per lane
    __shared__ uint64_t s_surv[NW][kSurvCap];
    __shared__ uint32_t s_cnt[NW];
    __shared__ uint64_t s_top[NW * kSmallKMax];
    const int img = blockIdx.x / p.blocks_per_image;
    const int blk = blockIdx.x - img * p.blocks_per_image;
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool largest = p.strategy != PP_ACQ_MARGIN;
    const float fill = largest ? 0.0f : 1.0f;
    const float* base = p.logits + (int64_t)img * p.sB;
    const uint8_t* excl = p.exclude ? p.exclude + (int64_t)img * p.N : nullptr;
    float* omap = p.out_map ? p.out_map + (int64_t)img * p.N : nullptr;
    const uint32_t lds_x = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)&s_x[wave][0][0]);
    const uint32_t lds_e = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)&s_e[wave][0][0]);

    auto issue = [&](int g, int buf) {
        const int64_t pix0 = ((int64_t)blk * G + g) * kBlock + wave * kWave;          // first pixel of this wave's slice
        const int64_t left = p.N - pix0;
        const int nflt = (int)(left <= 0 ? 0 : (left < kWave ? left : kWave)) * CMAX;  // valid floats of the slice
        const float* src0 = base + (left > 0 ? pix0 : 0) * CMAX;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the slot's previous contents have been read
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int f = (j * kWave + lane) * 4;
            glds16(src0 + (f < nflt ? f : 0), lds_x + (uint32_t)(buf * SLOT * 4 + j * 1024));   // every lane active: fixed op count
        }
        if (excl) glds1(excl + (lane < left ? pix0 + lane : 0), lds_e + (uint32_t)(buf * kWave * 4));
    };

    uint32_t kh[G], kl[G];
    issue(0, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (g + 1 < G) {
            issue(g + 1, (g + 1) & 1);
            if (excl) wait_vmcnt<NCH + 1>(); else wait_vmcnt<NCH>();     // slice g has landed, slice g+1 stays in flight
        } else {
            wait_vmcnt<0>();
        }
        const int64_t pix = ((int64_t)blk * G + g) * kBlock + tid;
        if (pix < p.N) {
            const float* sx = &s_x[wave][g & 1][lane * CMAX];
            float x[CMAX];
#pragma unroll
            for (int c = 0; c < CMAX; ++c) x[c] = sx[c];
            float sc = pixel_score_fast<CMAX, true>(x, p.C, p.strategy);
            if (excl && (s_e[wave][g & 1][lane] & 0xFFu)) sc = fill;
            if (omap) omap[pix] = sc;
            kh[g] = order_key(sc, largest);
            kl[g] = 0xFFFFFFFFu - (uint32_t)pix;
        } else {
            kh[g] = 0u; kl[g] = 0u;
        }
    }
    if (p.cand)
        block_emit_topk<G>(kh, kl, p.k, p.cand + ((int64_t)img * p.blocks_per_image + blk) * p.k, p.reduce_mode, s_surv, s_cnt, s_top);
}

// ---- acquisition straight from the LOW-resolution classifier logits -------------------
// Replaces  deeplab.py:55-56  F.interpolate(pred, size=inputs.shape[2:], mode='bilinear', align_corners=True)
//        +  query.py:190      softmax(model(x)["pred"][:, :, :h, :w])  + score + exclusion + top-k
// without ever writing the full-resolution logits (10 MB per 256x512x19 image written and read back; the algorithmic
// input drops 16x to the 64x128x19 classifier output).  A block owns a 64-column x 4*PPT-row tile of OUTPUT pixels,
// stages the low-resolution patch the tile interpolates from in LDS ((4*PPT*s+2) x (64*s+2) pixels, 13.7 KB at s = 1/4),
// and every lane interpolates its pixels' class vectors from LDS with bilerp() - the same bits pp_bilinear_fwd writes -
// then scores them like acq_kernel.  Wave w of the tile owns PPT consecutive rows, so the row weights are wave-uniform.
struct LowresParams {
    const float* low;    // [B,h,w,ldx] channels-last, C valid channels
    int64_t ldx;
    const uint8_t* exclude;   // [B,Hc,Wc] or null
    float* out_map;           // [B,Hc,Wc] or null
    uint64_t* cand;           // [B, waves_per_image, k] or null
    int h, w, Hc, Wc;
    float sh, sw;
    int align;
    int C, tiles_x, tiles_y, k, strategy, reduce_mode;
    int patch_cap;            // floats of dynamic LDS available for the patch
    uint32_t* qhist = nullptr;   // large-k selection: [B][kQBins] bin counts of the scores written to out_map (zeroed by the host); NULL: none
    float qscale = 0.0f;
};

template <int CMAX, bool EXACT, int PPT, bool LDS, int MATH, int STRAT = -1>
__global__ __launch_bounds__(kBlock, 2) void acq_lowres_kernel(LowresParams p)
{
    extern __shared__ __attribute__((aligned(16))) float s_patch[];
    __shared__ uint64_t s_surv[kBlock / kWave][kSurvCap];
    __shared__ uint32_t s_cnt[kBlock / kWave];
    __shared__ uint64_t s_top[(kBlock / kWave) * kSmallKMax];
    constexpr int TR = (kBlock / kWave) * PPT, TC = kWave;
    const int tiles = p.tiles_x * p.tiles_y;
    const int img = blockIdx.x / tiles;
    const int t = blockIdx.x - img * tiles;
    const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // large-k selection: the block counts its scores into the image's histogram (as acq_kernel<..., HIST>; bins in the survivor lists)
    uint32_t* qbins = reinterpret_cast<uint32_t*>(&s_surv[0][0]);
    const bool hist = p.qhist != nullptr;
    if (hist) {
        for (int i = tid; i < kQBins; i += kBlock) qbins[i] = 0u;
        __syncthreads();
    }
    const int C = EXACT ? CMAX : p.C;
    const int CP = C | 1;                     // odd pixel pitch: lanes 4 columns apart hit different banks
    const bool largest = p.strategy != PP_ACQ_MARGIN;
    const float fill = largest ? 0.0f : 1.0f;
    const int64_t N = (int64_t)p.Hc * p.Wc;
    const int X0 = tx * TC, Y0 = ty * TR;
    const int X1 = min(X0 + TC - 1, p.Wc - 1), Y1 = min(Y0 + TR - 1, p.Hc - 1);
    const int c_lo = lerp_src(X0, p.w, p.sw, p.align).i0, c_hi = lerp_src(X1, p.w, p.sw, p.align).i1;
    const int r_lo = lerp_src(Y0, p.h, p.sh, p.align).i0, r_hi = lerp_src(Y1, p.h, p.sh, p.align).i1;
    const int pw = c_hi - c_lo + 1, ph = r_hi - r_lo + 1;
    const float* base = p.low + (int64_t)img * p.h * p.w * p.ldx;
    if constexpr (LDS) {
        if (ph * pw * CP > p.patch_cap) __builtin_trap();   // host sizing bug: never silently read past the patch
        const int n = ph * pw * C;
        for (int e = tid; e < n; e += kBlock) {
            const int pc = e / C, ch = e - pc * C;
            const int r = pc / pw, c = pc - r * pw;
            s_patch[pc * CP + ch] = base[((int64_t)(r_lo + r) * p.w + c_lo + c) * p.ldx + ch];
        }
        __syncthreads();
    }
    const uint8_t* excl = p.exclude ? p.exclude + (int64_t)img * N : nullptr;
    float* omap = p.out_map ? p.out_map + (int64_t)img * N : nullptr;
    const int X = X0 + lane;
    const bool xin = X < p.Wc;
    const Lerp lw = lerp_src(xin ? X : X1, p.w, p.sw, p.align);
    const int64_t pitch = LDS ? (int64_t)CP : p.ldx;
    const int64_t o0 = (int64_t)(LDS ? lw.i0 - c_lo : lw.i0) * pitch, o1 = (int64_t)(LDS ? lw.i1 - c_lo : lw.i1) * pitch;
    const float* src = LDS ? s_patch : base;
    const int64_t row_pitch = (LDS ? pw : p.w) * pitch;

    uint32_t kh[PPT], kl[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int Y = Y0 + wv * PPT + j;
        if (Y < p.Hc && xin) {
            const Lerp lh = lerp_src(Y, p.h, p.sh, p.align);
            const float* r0 = src + (int64_t)(LDS ? lh.i0 - r_lo : lh.i0) * row_pitch;
            const float* r1 = src + (int64_t)(LDS ? lh.i1 - r_lo : lh.i1) * row_pitch;
            float x[CMAX];
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (EXACT || c < C)
                    x[c] = bilerp(lh.l0, lh.l1, lw.l0, lw.l1, r0[o0 + c], r0[o1 + c], r1[o0 + c], r1[o1 + c]);
            float sc;
            if constexpr (MATH == 0) sc = pixel_score_fast<CMAX, EXACT, STRAT>(x, p.C, p.strategy);
            else sc = pixel_score<CMAX, EXACT>(x, p.C, p.strategy, 0);
            const int64_t pix = (int64_t)Y * p.Wc + X;
            if (excl && excl[pix]) sc = fill;
            if (omap) omap[pix] = sc;
            if (hist) atomicAdd(&qbins[qbin(sc, largest, p.qscale)], 1u);
            kh[j] = order_key(sc, largest);
            kl[j] = 0xFFFFFFFFu - (uint32_t)pix;
        } else {
            kh[j] = 0u; kl[j] = 0u;
        }
        __builtin_amdgcn_sched_barrier(0);   // one pixel's class vector live at a time
    }
    if (hist) {
        __syncthreads();
        uint32_t* Hg = p.qhist + (int64_t)img * kQBins;
        for (int d = tid; d < kQBins; d += kBlock) {
            const uint32_t c = qbins[d];
            if (c) atomicAdd(&Hg[d], c);
        }
        return;
    }
    if (p.cand)
        block_emit_topk<PPT>(kh, kl, p.k, p.cand + ((int64_t)img * tiles + t) * p.k, p.reduce_mode, s_surv, s_cnt, s_top);
}

// The strategy's score at a list of picked pixels (QueryStats: entropy at the queried pixels, query.py:262-266),
// interpolated from the low-resolution logits with the same arithmetic.  One thread per pixel.
template <int CMAX, bool EXACT>
__global__ __launch_bounds__(kBlock) void acq_lowres_at_kernel(const float* low, int64_t ldx, int h, int w, float sh,
                                                             float sw, int align, int Wc, int C, int strategy,
                                                             const int32_t* img_idx, const int32_t* pix_idx,
                                                             int64_t n, float* out)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int pix = pix_idx[i], Y = pix / Wc, X = pix - Y * Wc;
    const Lerp lh = lerp_src(Y, h, sh, align), lw = lerp_src(X, w, sw, align);
    const float* base = low + (int64_t)img_idx[i] * h * w * ldx;
    const float* p00 = base + ((int64_t)lh.i0 * w + lw.i0) * ldx;
    const float* p01 = base + ((int64_t)lh.i0 * w + lw.i1) * ldx;
    const float* p10 = base + ((int64_t)lh.i1 * w + lw.i0) * ldx;
    const float* p11 = base + ((int64_t)lh.i1 * w + lw.i1) * ldx;
    float x[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
        if (EXACT || c < C) x[c] = bilerp(lh.l0, lh.l1, lw.l0, lw.l1, p00[c], p01[c], p10[c], p11[c]);
    out[i] = pixel_score_fast<CMAX, EXACT>(x, C, strategy);
}

// ---- any class count: C > PP_ACQ_REG_CLASSES ------------------------------------------------------------
// The scorers above keep a pixel's class vector in registers (C <= 64).  The reference takes whatever class count the model emits
// (query.py:190 softmaxes dim 1 of any width), so wider heads STREAM the class vector instead: two (default scorer) or three
// (reference operation order) passes over the pixel's classes, the later ones served by L2, with the same expressions in the same
// order as pixel_score_fast / pixel_score - forced onto a C <= 64 input (pp_debug_set_acq_tuning occ = 10) the maps are bit-equal
// to the register kernels' (tests/test_acq_gpu.py).  Load(c, v) fills v[0..VEC) with class c of the thread's VEC pixels.
template <int VEC, int MATH, typename Load>
__device__ __forceinline__ void score_stream(Load load, int C, int strategy, float (&s)[VEC])
{
#pragma clang fp contract(off)      // as pixel_score; the default scorer's fused steps are spelled fmaf()
    float v[VEC];
    if constexpr (MATH == 2) {                       // input holds probabilities (UncertaintySampler.__call__)
        float acc[VEC], t1[VEC], t2[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) { acc[i] = 0.0f; t1[i] = -INFINITY; t2[i] = -INFINITY; }
        for (int c = 0; c < C; ++c) {
            load(c, v);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                if (strategy == PP_ACQ_ENTROPY) acc[i] += (-v[i]) * logf(v[i]);
                else { t2[i] = fmaxf(t2[i], fminf(t1[i], v[i])); t1[i] = fmaxf(t1[i], v[i]); }
            }
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i)
            s[i] = strategy == PP_ACQ_ENTROPY ? acc[i] : (strategy == PP_ACQ_LEAST_CONFIDENCE ? 1.0f - t1[i] : fabsf(t1[i] - t2[i]));
        return;
    }
    float m[VEC], x2[VEC], xmin[VEC];
    load(0, v);
#pragma unroll
    for (int i = 0; i < VEC; ++i) { m[i] = v[i]; x2[i] = -INFINITY; xmin[i] = v[i]; }
    for (int c = 1; c < C; ++c) {
        load(c, v);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            x2[i] = fmaxf(x2[i], fminf(m[i], v[i]));
            m[i] = fmaxf(m[i], v[i]);
            xmin[i] = fminf(xmin[i], v[i]);
        }
    }
    float S[VEC], T[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { S[i] = 0.0f; T[i] = 0.0f; }
    if constexpr (MATH == 0) {
        for (int c = 0; c < C; ++c) {
            load(c, v);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float d = v[i] - m[i];
                const float e = fast_exp(d);
                S[i] += e;
                T[i] = fmaf(e, -d, T[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            if (strategy == PP_ACQ_ENTROPY) {
                float ent = logf(S[i]) + T[i] / S[i];
                if (xmin[i] - m[i] < -87.0f) {
                    if (expf(xmin[i] - m[i]) / S[i] == 0.0f) ent = __uint_as_float(0x7FC00000u);
                }
                s[i] = ent;
            } else if (strategy == PP_ACQ_LEAST_CONFIDENCE) {
                s[i] = 1.0f - 1.0f / S[i];
            } else {
                s[i] = fabsf(1.0f / S[i] - expf(x2[i] - m[i]) / S[i]);
            }
        }
    } else {                                         // reference operation order (query.py:190,229-239)
        for (int c = 0; c < C; ++c) {
            load(c, v);
#pragma unroll
            for (int i = 0; i < VEC; ++i) S[i] += expf(v[i] - m[i]);
        }
        if (strategy == PP_ACQ_LEAST_CONFIDENCE) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) s[i] = 1.0f - 1.0f / S[i];
            return;
        }
        float t1[VEC], t2[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) { T[i] = 0.0f; t1[i] = -INFINITY; t2[i] = -INFINITY; }
        for (int c = 0; c < C; ++c) {
            load(c, v);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float e = expf(v[i] - m[i]);
                if (strategy == PP_ACQ_ENTROPY) {
                    const float pr = e / S[i];
                    T[i] += (-pr) * logf(pr);
                } else {
                    t2[i] = fmaxf(t2[i], fminf(t1[i], e));
                    t1[i] = fmaxf(t1[i], e);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) s[i] = strategy == PP_ACQ_ENTROPY ? T[i] : fabsf(t1[i] / S[i] - t2[i] / S[i]);
    }
}

// acq_kernel's block geometry (a block covers kBlock * VEC * G consecutive pixels of one image), class vector streamed
template <int VEC, int G, int MATH>
__global__ __launch_bounds__(kBlock) void acq_stream_kernel(AcqParams p)
{
    constexpr int PPT = VEC * G;
    __shared__ uint64_t s_surv[kBlock / kWave][kSurvCap];
    __shared__ uint32_t s_cnt[kBlock / kWave];
    __shared__ uint64_t s_top[(kBlock / kWave) * kSmallKMax];
    const int img = blockIdx.x / p.blocks_per_image;
    const int blk = blockIdx.x - img * p.blocks_per_image;
    const int tid = threadIdx.x;
    const bool largest = p.strategy != PP_ACQ_MARGIN;
    const float fill = largest ? 0.0f : 1.0f;
    const float* base = p.logits + (int64_t)img * p.sB;
    const uint8_t* excl = p.exclude ? p.exclude + (int64_t)img * p.N : nullptr;
    float* omap = p.out_map ? p.out_map + (int64_t)img * p.N : nullptr;
    uint32_t kh[PPT], kl[PPT];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int64_t pix0 = ((int64_t)blk * G + g) * (kBlock * VEC) + (int64_t)tid * VEC;
        if (pix0 < p.N) {
            float s[VEC];
            if constexpr (VEC == 4) {
                const float* px = base + pix0;
                const int64_t sC = p.sC;
                score_stream<4, MATH>([&](int c, float (&v)[4]) {
                    const float4 q = *reinterpret_cast<const float4*>(px + (int64_t)c * sC);
                    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
                }, p.C, p.strategy, s);
                const uint32_t ex = excl ? *reinterpret_cast<const uint32_t*>(excl + pix0) : 0u;
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    if ((ex >> (8 * v)) & 0xFFu) s[v] = fill;
                if (omap) *reinterpret_cast<float4*>(omap + pix0) = make_float4(s[0], s[1], s[2], s[3]);
            } else {
                const int64_t h = pix0 / p.W, w = pix0 - h * p.W;
                const float* px = base + h * p.sH + w * p.sW;
                const int64_t sC = p.sC;
                score_stream<1, MATH>([&](int c, float (&v)[1]) { v[0] = px[(int64_t)c * sC]; }, p.C, p.strategy, s);
                if (excl && excl[pix0]) s[0] = fill;
                if (omap) omap[pix0] = s[0];
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                kh[g * VEC + v] = order_key(s[v], largest);
                kl[g * VEC + v] = 0xFFFFFFFFu - (uint32_t)(pix0 + v);
            }
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) { kh[g * VEC + v] = 0u; kl[g * VEC + v] = 0u; }
        }
    }
    if (p.cand)
        block_emit_topk<PPT>(kh, kl, p.k, p.cand + ((int64_t)img * p.blocks_per_image + blk) * p.k, p.reduce_mode, s_surv, s_cnt, s_top);
}

// acq_lowres_kernel's tile geometry (PPT = 4, no LDS patch): every class of a pixel is interpolated from the four low-resolution
// neighbours in memory on each pass (bilerp(): the bits pp_bilinear_fwd writes)
template <int MATH>
__global__ __launch_bounds__(kBlock) void acq_lowres_stream_kernel(LowresParams p)
{
    constexpr int PPT = 4;
    __shared__ uint64_t s_surv[kBlock / kWave][kSurvCap];
    __shared__ uint32_t s_cnt[kBlock / kWave];
    __shared__ uint64_t s_top[(kBlock / kWave) * kSmallKMax];
    constexpr int TR = (kBlock / kWave) * PPT, TC = kWave;
    const int tiles = p.tiles_x * p.tiles_y;
    const int img = blockIdx.x / tiles;
    const int t = blockIdx.x - img * tiles;
    const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool largest = p.strategy != PP_ACQ_MARGIN;
    const float fill = largest ? 0.0f : 1.0f;
    const int64_t N = (int64_t)p.Hc * p.Wc;
    const int X = tx * TC + lane;
    const bool xin = X < p.Wc;
    const Lerp lw = lerp_src(xin ? X : p.Wc - 1, p.w, p.sw, p.align);
    const float* base = p.low + (int64_t)img * p.h * p.w * p.ldx;
    const uint8_t* excl = p.exclude ? p.exclude + (int64_t)img * N : nullptr;
    float* omap = p.out_map ? p.out_map + (int64_t)img * N : nullptr;
    uint32_t kh[PPT], kl[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int Y = ty * TR + wv * PPT + j;
        if (Y < p.Hc && xin) {
            const Lerp lh = lerp_src(Y, p.h, p.sh, p.align);
            const float* p00 = base + ((int64_t)lh.i0 * p.w + lw.i0) * p.ldx;
            const float* p01 = base + ((int64_t)lh.i0 * p.w + lw.i1) * p.ldx;
            const float* p10 = base + ((int64_t)lh.i1 * p.w + lw.i0) * p.ldx;
            const float* p11 = base + ((int64_t)lh.i1 * p.w + lw.i1) * p.ldx;
            float s[1];
            score_stream<1, MATH>([&](int c, float (&v)[1]) {
                v[0] = bilerp(lh.l0, lh.l1, lw.l0, lw.l1, p00[c], p01[c], p10[c], p11[c]);
            }, p.C, p.strategy, s);
            const int64_t pix = (int64_t)Y * p.Wc + X;
            if (excl && excl[pix]) s[0] = fill;
            if (omap) omap[pix] = s[0];
            kh[j] = order_key(s[0], largest);
            kl[j] = 0xFFFFFFFFu - (uint32_t)pix;
        } else {
            kh[j] = 0u; kl[j] = 0u;
        }
    }
    if (p.cand)
        block_emit_topk<PPT>(kh, kl, p.k, p.cand + ((int64_t)img * tiles + t) * p.k, p.reduce_mode, s_surv, s_cnt, s_top);
}

__global__ __launch_bounds__(kBlock) void acq_lowres_at_stream_kernel(const float* low, int64_t ldx, int h, int w, float sh, float sw,
                                                                     int align, int Wc, int C, int strategy, const int32_t* img_idx,
                                                                     const int32_t* pix_idx, int64_t n, float* out)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int pix = pix_idx[i], Y = pix / Wc, X = pix - Y * Wc;
    const Lerp lh = lerp_src(Y, h, sh, align), lw = lerp_src(X, w, sw, align);
    const float* base = low + (int64_t)img_idx[i] * h * w * ldx;
    const float* p00 = base + ((int64_t)lh.i0 * w + lw.i0) * ldx;
    const float* p01 = base + ((int64_t)lh.i0 * w + lw.i1) * ldx;
    const float* p10 = base + ((int64_t)lh.i1 * w + lw.i0) * ldx;
    const float* p11 = base + ((int64_t)lh.i1 * w + lw.i1) * ldx;
    float s[1];
    score_stream<1, 0>([&](int c, float (&v)[1]) { v[0] = bilerp(lh.l0, lh.l1, lw.l0, lw.l1, p00[c], p01[c], p10[c], p11[c]); },
                       C, strategy, s);
    out[i] = s[0];
}

// ---- small-k selection straight from a score map (pp_topk_select) ------------------------------------
template <int G>
__global__ __launch_bounds__(kBlock) void topk_small_from_scores_kernel(const float* scores, int64_t N,
                                                                        int blocks_per_image, int k,
                                                                        int largest, uint64_t* cand, int mode)
{
    const int img = blockIdx.x / blocks_per_image;
    const int blk = blockIdx.x - img * blocks_per_image;
    const int tid = threadIdx.x;
    const float* s = scores + (int64_t)img * N;
    uint32_t kh[G], kl[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int64_t pix = ((int64_t)blk * G + g) * kBlock + tid;
        if (pix < N) {
            kh[g] = order_key(s[pix], largest != 0);
            kl[g] = 0xFFFFFFFFu - (uint32_t)pix;
        } else {
            kh[g] = 0u; kl[g] = 0u;
        }
    }
    __shared__ uint64_t s_surv[kBlock / kWave][kSurvCap];
    __shared__ uint32_t s_cnt[kBlock / kWave];
    __shared__ uint64_t s_top[(kBlock / kWave) * kSmallKMax];
    block_emit_topk<G>(kh, kl, k, cand + ((int64_t)img * blocks_per_image + blk) * k, mode, s_surv, s_cnt, s_top);
}

// ---- candidate merge ---------------------------------------------------------------------------------
// grid (nchunks, B): block (c,b) reduces up to kMergeChunk candidate keys of image b to their top-k with the machinery the scoring
// kernels end in - per-wave threshold prefilter + rank inside LDS (block_emit_topk) - instead of k rounds of a block-wide arg-max
// with two barriers each (10.5 us per level at k = 20; this form: ~3 us).  The last level (nchunks == 1) decodes to out_idx / out_val.
__global__ __launch_bounds__(kBlock) void cand_merge_kernel(const uint64_t* in, int64_t n_in, uint64_t* out_keys,
                                                           int32_t* out_idx, float* out_val, int k, int largest,
                                                           int mode)
{
    __shared__ uint64_t s_surv[kBlock / kWave][kSurvCap];
    __shared__ uint32_t s_cnt[kBlock / kWave];
    __shared__ uint64_t s_top[(kBlock / kWave) * kSmallKMax];
    const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
    const uint64_t* src = in + (int64_t)b * n_in;
    const int64_t lo = (int64_t)c * kMergeChunk;
    uint32_t kh[kMergeItems], kl[kMergeItems];
#pragma unroll
    for (int j = 0; j < kMergeItems; ++j) {
        const int64_t i = lo + (int64_t)j * kBlock + tid;
        uint64_t v = i < n_in ? src[i] : 0ull;
        kh[j] = (uint32_t)(v >> 32);
        kl[j] = (uint32_t)v;
    }
    if (out_keys) {
        block_emit_topk<kMergeItems>(kh, kl, k, out_keys + ((int64_t)b * gridDim.x + c) * k, mode, s_surv, s_cnt, s_top);
        return;
    }
    block_emit_topk<kMergeItems>(kh, kl, k, s_top, mode, s_surv, s_cnt, s_top);      // merged list -> s_top[0..k)
    if (tid < kWave) {
        __builtin_amdgcn_wave_barrier();
        for (int r = tid; r < k; r += kWave) {
            const uint64_t v = s_top[r];
            const uint32_t mh = (uint32_t)(v >> 32), ml = (uint32_t)v;
            out_idx[(int64_t)b * k + r] = mh == 0u ? -1 : (int32_t)(0xFFFFFFFFu - ml);
            if (out_val) out_val[(int64_t)b * k + r] = key_to_float(mh, largest != 0);
        }
    }
}

// ---- large-k: radix select + bitonic sort, one 1024-thread block per image ------------------------------
// query.py:36 top_n_percent mode: k = int(h*w*0.05) (6553 at 256x512) value-sorted indices.
// s: the image's scores; hist: 256 words, misc: 64 words of LDS; buf: P 64-bit words (LDS or global); oi / ov: the image's output rows.
// All kLargeThreads threads of the block.
__device__ __forceinline__ void topk_large_body(const float* s, int64_t N, int k, bool lg, uint32_t* hist, uint32_t* misc, uint64_t* buf,
                                                int P, int32_t* oi, float* ov)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // 1. radix select: T = k-th largest key, need_eq = how many keys == T to take (lowest index first)
    uint32_t prefix = 0, mask = 0, remaining = (uint32_t)k;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (int64_t i = tid; i < N; i += kLargeThreads) {
            const uint32_t key = order_key(s[i], lg);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t cum = 0;
            int d = 255;
            for (; d > 0; --d) {
                const uint32_t cnt = hist[d];
                if (cum + cnt >= remaining) break;
                cum += cnt;
            }
            misc[0] = (uint32_t)d;
            misc[1] = remaining - cum;
        }
        __syncthreads();
        prefix |= misc[0] << shift;
        mask |= 0xFFu << shift;
        remaining = misc[1];
        __syncthreads();
    }
    const uint32_t T = prefix, need_eq = remaining;

    // 2. compaction (index order matters only among keys == T)
    if (tid == 0) { misc[2] = 0; /* out count */ misc[3] = 0; /* eq seen so far */ }
    __syncthreads();
    for (int64_t base = 0; base < N; base += kLargeThreads) {
        const int64_t i = base + tid;
        const uint32_t key = i < N ? order_key(s[i], lg) : 0u;
        const bool gt = key > T, eq = (i < N) && key == T;
        const unsigned long long bal = __ballot(eq);
        if (lane == 0) misc[8 + wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t before = misc[3];
        for (int q = 0; q < wave; ++q) before += misc[8 + q];
        const uint32_t rank = before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        const bool take = gt || (eq && rank < need_eq);
        if (take) {
            const uint32_t pos = atomicAdd(&misc[2], 1u);
            buf[pos] = ((uint64_t)key << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)i);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t tot = 0;
            for (int q = 0; q < kLargeThreads / kWave; ++q) tot += misc[8 + q];
            misc[3] += tot;
        }
        __syncthreads();
    }
    for (int j = k + tid; j < P; j += kLargeThreads) buf[j] = 0ull;
    __syncthreads();

    // 3. bitonic sort, descending
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (P >> 1); t += kLargeThreads) {
                const int pos = 2 * t - (t & (stride - 1));
                const uint64_t a = buf[pos], b = buf[pos + stride];
                const bool desc = (pos & size) == 0;
                if ((a < b) == desc) { buf[pos] = b; buf[pos + stride] = a; }
            }
            __syncthreads();
        }
    }
    for (int j = tid; j < k; j += kLargeThreads) {
        const uint64_t v = buf[j];
        oi[j] = (int32_t)(0xFFFFFFFFu - (uint32_t)v);
        if (ov) ov[j] = key_to_float((uint32_t)(v >> 32), lg);
    }
}

__global__ __launch_bounds__(kLargeThreads) void topk_large_kernel(const float* scores, int64_t N, int k, int largest,
                                                                  uint64_t* gbuf, int P, int32_t* out_idx,
                                                                  float* out_val, const int* only_if = nullptr)
{
    if (only_if && !only_if[blockIdx.x]) return;      // fallback launch of the quantised select: flagged images only
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem);            // 256
    uint32_t* misc = hist + 256;                                   // 64
    uint64_t* buf = gbuf ? gbuf + (int64_t)blockIdx.x * P : reinterpret_cast<uint64_t*>(smem + (256 + 64) * 4);
    topk_large_body(scores + (int64_t)blockIdx.x * N, N, k, largest != 0, hist, misc, buf, P, out_idx + (int64_t)blockIdx.x * k,
                    out_val ? out_val + (int64_t)blockIdx.x * k : nullptr);
}


// ---- large-k, multi-block: the radix select of topk_large_kernel taken out of the one-block-per-image kernel -------------
// Four histogram passes over the score map (8 key bits each, 256 bins), every pass a grid of blocks_per_image x B
// blocks that add their LDS histograms into the image's global one with integer atomics (order-independent, exact).  No
// separate scan launch: the blocks of pass p (and the final kernel) re-derive the digits chosen so far from the earlier
// histograms - a 256-bin scan per earlier pass.  The final one-block-per-image kernel then knows the threshold key T, how
// many keys equal to T it must take and how many exist, compacts in one barrier-free sweep and sorts.
constexpr int kSelBins = 256;         // 8-bit digits, four passes: 8 KiB of LDS histograms per block instead of 64 KiB (the 2048-bin
constexpr int kSelPasses = 4;         // / three-pass form spent most of a pass zeroing and flushing its LDS: 84 us per pass)
constexpr int kSelSub = 8;            // sub-histograms per block (lane & 7): same-address LDS atomics conflict 8x less

__device__ __forceinline__ uint32_t sel_digit(uint32_t key, int pass) { return (key >> (24 - 8 * pass)) & 255u; }
__device__ __forceinline__ int sel_shift(int pass) { return 24 - 8 * pass; }
__device__ __forceinline__ uint32_t sel_mask(int pass) { return pass == 0 ? 0u : (0xFFFFFFFFu << (32 - 8 * pass)); }

// From the top bin down: the digit d with  #(bins above d) < remaining <= #(bins above d) + hist[d]  (d = 0 if the bins above
// it hold fewer than `remaining`).  Parallel: thread t < 256 owns bin 255 - t, an inclusive scan over the threads gives the
// count of bins at or above each one (a serial walk with its dependent LDS reads costs ~10 us per 256-bin histogram).
// h: 256 words in LDS or global; sh: >= 8 words of LDS; out[0] = d, out[1] = remaining - #above, out[2] = hist[d].
// Must be called by all threads of the block (two barriers); blocks of 256 or more threads.
__device__ __forceinline__ void sel_find_digit(const uint32_t* h, uint32_t remaining, uint32_t* sh, uint32_t* out)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t own = 0, incl = 0;
    if (t < kSelBins) {
        own = h[kSelBins - 1 - t];
        incl = own;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, o, 64);
            if (lane >= o) incl += v;
        }
        if (lane == 63) sh[wave] = incl;
    }
    __syncthreads();
    if (t < kSelBins) {
        uint32_t base = 0;
        for (int w = 0; w < wave; ++w) base += sh[w];
        incl += base;
        const uint32_t excl = incl - own;
        const bool hit = excl < remaining && remaining <= incl;
        if ((hit && t < kSelBins - 1) || (t == kSelBins - 1 && excl < remaining)) {      // bin 0 takes whatever is left
            out[0] = (uint32_t)(kSelBins - 1 - t);
            out[1] = remaining - excl;
            out[2] = own;
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void sel_scan(const uint32_t* hist, uint32_t remaining, uint32_t* sh /*[kSelBins]*/, uint32_t* out /*[3]*/)
{
    sel_find_digit(hist, remaining, sh, out);
}

// hist: [B][kSelPasses][kSelBins] (zeroed by the host before pass 0)
__global__ __launch_bounds__(kBlock) void select_hist_kernel(const float* scores, int64_t N, int k, int largest, int pass, uint32_t* hist)
{
    __shared__ uint32_t lh[kSelSub][kSelBins];
    __shared__ uint32_t sh[kBlock];
    __shared__ uint32_t res[3];
    const int b = blockIdx.y;
    const float* s = scores + (int64_t)b * N;
    uint32_t* H = hist + (int64_t)b * kSelPasses * kSelBins;
    const bool lg = largest != 0;
    uint32_t prefix = 0, remaining = (uint32_t)k;
    for (int q = 0; q < pass; ++q) {
        sel_scan(H + q * kSelBins, remaining, sh, res);
        prefix |= res[0] << sel_shift(q);
        remaining = res[1];
        __syncthreads();
    }
    for (int i = threadIdx.x; i < kSelSub * kSelBins; i += kBlock) (&lh[0][0])[i] = 0u;
    __syncthreads();
    const uint32_t mask = sel_mask(pass);
    const int sub = threadIdx.x & (kSelSub - 1);
    const int64_t per = (N + gridDim.x - 1) / gridDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * per, i1 = i0 + per < N ? i0 + per : N;
    int64_t i = i0 + threadIdx.x;
    for (; i + 3 * kBlock < i1; i += 4 * kBlock) {            // four loads in flight per thread
        const float v0 = s[i], v1 = s[i + kBlock], v2 = s[i + 2 * kBlock], v3 = s[i + 3 * kBlock];
        const uint32_t k0 = order_key(v0, lg), k1 = order_key(v1, lg), k2 = order_key(v2, lg), k3 = order_key(v3, lg);
        if ((k0 & mask) == prefix) atomicAdd(&lh[sub][sel_digit(k0, pass)], 1u);
        if ((k1 & mask) == prefix) atomicAdd(&lh[sub][sel_digit(k1, pass)], 1u);
        if ((k2 & mask) == prefix) atomicAdd(&lh[sub][sel_digit(k2, pass)], 1u);
        if ((k3 & mask) == prefix) atomicAdd(&lh[sub][sel_digit(k3, pass)], 1u);
    }
    for (; i < i1; i += kBlock) {
        const uint32_t key = order_key(s[i], lg);
        if ((key & mask) == prefix) atomicAdd(&lh[sub][sel_digit(key, pass)], 1u);
    }
    __syncthreads();
    uint32_t* Hp = H + pass * kSelBins;
    for (int d = threadIdx.x; d < kSelBins; d += kBlock) {
        uint32_t c = 0;
#pragma unroll
        for (int u = 0; u < kSelSub; ++u) c += lh[u][d];
        if (c) atomicAdd(&Hp[d], c);
    }
}

// One 1024-thread block per image: threshold from the three histograms, barrier-free compaction, bitonic sort.
__global__ __launch_bounds__(kLargeThreads) void topk_large_sel_kernel(const float* scores, int64_t N, int k, int largest, const uint32_t* hist,
                                                                      uint64_t* gbuf, int P, int32_t* out_idx, float* out_val)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* shs = reinterpret_cast<uint32_t*>(smem);            // 256: scan partials
    uint32_t* misc = shs + 256;                                    // 64
    uint64_t* buf = gbuf ? gbuf + (int64_t)blockIdx.x * P : reinterpret_cast<uint64_t*>(smem + (256 + 64) * 4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* s = scores + (int64_t)blockIdx.x * N;
    const uint32_t* H = hist + (int64_t)blockIdx.x * kSelPasses * kSelBins;
    const bool lg = largest != 0;
    uint32_t T = 0, need_eq = (uint32_t)k, total_eq = 0;
    for (int q = 0; q < kSelPasses; ++q) {
        sel_scan(H + q * kSelBins, need_eq, shs, misc);
        T |= misc[0] << sel_shift(q);
        need_eq = misc[1];
        total_eq = misc[2];
        __syncthreads();
    }
    if (tid == 0) { misc[4] = 0; /* out count */ misc[5] = 0; /* eq seen so far (ordered path) */ }
    __syncthreads();
    const bool all_eq = need_eq == total_eq;          // every key equal to T is taken: their order is irrelevant (the sort fixes it)
    auto push = [&](uint32_t key, int64_t i, bool valid) {
        const bool take = valid && (key > T || (all_eq && key == T));
        const unsigned long long bal = __ballot(take);
        if (bal) {
            uint32_t wbase = 0;
            if (lane == 0) wbase = atomicAdd(&misc[4], (uint32_t)__popcll(bal));
            wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
            if (take) buf[wbase + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = ((uint64_t)key << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)i);
        }
    };
    {
        int64_t base = 0;
        for (; base + 4 * kLargeThreads <= N; base += 4 * kLargeThreads) {        // four loads in flight per thread
            const int64_t i = base + tid;
            const float v0 = s[i], v1 = s[i + kLargeThreads], v2 = s[i + 2 * kLargeThreads], v3 = s[i + 3 * kLargeThreads];
            push(order_key(v0, lg), i, true);
            push(order_key(v1, lg), i + kLargeThreads, true);
            push(order_key(v2, lg), i + 2 * kLargeThreads, true);
            push(order_key(v3, lg), i + 3 * kLargeThreads, true);
        }
        for (; base < N; base += kLargeThreads) {
            const int64_t i = base + tid;
            push(i < N ? order_key(s[i], lg) : 0u, i, i < N);
        }
    }
    __syncthreads();
    if (!all_eq) {
        // ties at the threshold (constant regions): the need_eq keys == T with the LOWEST indices, found in index order
        for (int64_t base = 0; base < N; base += kLargeThreads) {
            const int64_t i = base + tid;
            const bool eq = i < N && order_key(s[i], lg) == T;
            const unsigned long long bal = __ballot(eq);
            if (lane == 0) misc[8 + wave] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t before = misc[5];
            for (int q = 0; q < wave; ++q) before += misc[8 + q];
            const uint32_t rank = before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            if (eq && rank < need_eq) {
                const uint32_t pos = atomicAdd(&misc[4], 1u);
                buf[pos] = ((uint64_t)T << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)i);
            }
            __syncthreads();
            if (tid == 0) {
                uint32_t tot = 0;
                for (int q = 0; q < kLargeThreads / kWave; ++q) tot += misc[8 + q];
                misc[5] += tot;
            }
            __syncthreads();
            if (misc[5] >= need_eq) break;
        }
        __syncthreads();
    }
    for (int j = k + tid; j < P; j += kLargeThreads) buf[j] = 0ull;
    __syncthreads();
    // bitonic sort, descending.  Strides >= 8 exchange through `buf`; the three last stages of every size (strides 4, 2, 1)
    // and the whole of sizes 2..8 run in registers on the thread's own 8 consecutive keys: 66 barrier phases instead of 91.
    auto reg_stages = [&](int size, bool head) {
        for (int g = tid; g < (P >> 3); g += kLargeThreads) {
            uint64_t v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = buf[8 * g + j];
            auto stage = [&](auto S_, int sz) {
                constexpr int S = decltype(S_)::value;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if ((j & S) == 0) {
                        const bool desc = ((8 * g + j) & sz) == 0;
                        const uint64_t a = v[j], b = v[j + S];
                        const bool sw = (a < b) == desc;
                        v[j] = sw ? b : a;
                        v[j + S] = sw ? a : b;
                    }
                }
            };
            using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>;
            using I4 = std::integral_constant<int, 4>;
            if (head) {
                stage(I1{}, 2);
                stage(I2{}, 4); stage(I1{}, 4);
                stage(I4{}, 8); stage(I2{}, 8); stage(I1{}, 8);
            } else {
                stage(I4{}, size); stage(I2{}, size); stage(I1{}, size);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) buf[8 * g + j] = v[j];
        }
        __syncthreads();
    };
    if (P >= 8) {
        reg_stages(8, true);                                // sizes 2, 4, 8 entirely in registers
        for (int size = 16; size <= P; size <<= 1) {
            for (int stride = size >> 1; stride >= 8; stride >>= 1) {
                for (int t = tid; t < (P >> 1); t += kLargeThreads) {
                    const int pos = 2 * t - (t & (stride - 1));
                    const uint64_t a = buf[pos], b = buf[pos + stride];
                    const bool desc = (pos & size) == 0;
                    if ((a < b) == desc) { buf[pos] = b; buf[pos + stride] = a; }
                }
                __syncthreads();
            }
            reg_stages(size, false);                        // strides 4, 2, 1 of this size
        }
    } else {
        for (int size = 2; size <= P; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int t = tid; t < (P >> 1); t += kLargeThreads) {
                    const int pos = 2 * t - (t & (stride - 1));
                    const uint64_t a = buf[pos], b = buf[pos + stride];
                    const bool desc = (pos & size) == 0;
                    if ((a < b) == desc) { buf[pos] = b; buf[pos + stride] = a; }
                }
                __syncthreads();
            }
        }
    }
    for (int j = tid; j < k; j += kLargeThreads) {
        const uint64_t v = buf[j];
        out_idx[(int64_t)blockIdx.x * k + j] = (int32_t)(0xFFFFFFFFu - (uint32_t)v);
        if (out_val) out_val[(int64_t)blockIdx.x * k + j] = key_to_float((uint32_t)(v >> 32), lg);
    }
}


// ---- large-k with a KNOWN score range: one quantised histogram pass, compaction, register / shuffle bitonic sort ------------
// The scorers' outputs live in a known interval (entropy [0, ln C], least confidence and margin [0, 1]), so the 32-bit radix
// digits of the generic select are not needed to find a threshold: kQBins LINEAR bins of the score - a monotone map of the order
// key, so every element of a higher bin precedes every element of a lower one - are filled in ONE pass over the map, the bin that
// holds the k-th element is found by a scan, and everything in that bin or above (k + the bin's population: k + ~130 of 131072
// at the reference's default k = 6553) is dropped into LDS bin by bin and ranked exactly on the full (key, index) words.  Replaces four
// histogram passes + a 66-barrier LDS sort; an image whose candidates do not fit (kQCap in all, kQMaxPop in a bin: heavy ties, constant
// maps) raises its overflow flag and is redone by topk_large_kernel (the exact one-block radix select): the result never depends on the data.
constexpr int kQSub = 8;             // sub-histograms per block (lane & 7)
constexpr int kQCap = 8192;          // candidates per image held in LDS (64 KiB of (key, index) words)

// Generic maps (pp_topk_select: the random strategy's maps, MC-dropout means - no score range known): one pass takes every image's finite
// minimum and maximum as order keys (mm[2 b] = largest key seen, mm[2 b + 1] = largest complemented key, both zeroed by the host), and the
// quantised select bins (s - lo) * kQBins / (hi - lo) per image.  Infinities fall into the end bins, NaN where order_key puts it; an image
// with a single value fills one bin and goes to the exact fallback like any heavily tied map.
__device__ __forceinline__ void image_range(const uint32_t* mm, int b, float& lo, float& scale)
{
    const uint32_t kmax = mm[2 * b], kmin = ~mm[2 * b + 1];
    const float hi = key_to_float(kmax, true);
    lo = key_to_float(kmin, true);
    if (kmax == 0u || !(hi > lo)) { lo = 0.0f; scale = 1.0f; return; }          // no finite value, or a single one
    scale = (float)kQBins / (hi - lo);
    if (!(scale < 3.0e38f)) scale = 3.0e38f;
}

__global__ __launch_bounds__(kBlock) void select_minmax_kernel(const float* scores, int64_t N, uint32_t* mm)
{
    __shared__ uint32_t smax[kBlock / kWave], smin[kBlock / kWave];
    const int b = blockIdx.y;
    const float* s = scores + (int64_t)b * N;
    const int64_t per = (N + gridDim.x - 1) / gridDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * per, i1 = i0 + per < N ? i0 + per : N;
    uint32_t kx = 0u, kn = 0u;                                  // largest key, largest complemented key (0 = nothing seen)
    for (int64_t i = i0 + threadIdx.x; i < i1; i += kBlock) {
        const float v = s[i];
        if (fabsf(v) <= 3.4028234e38f) {                        // finite (NaN compares false)
            const uint32_t key = order_key(v, true);
            kx = key > kx ? key : kx;
            kn = ~key > kn ? ~key : kn;
        }
    }
    kx = wave_umax(kx, 0);
    kn = wave_umax(kn, 0);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    if (lane == 0) { smax[wave] = kx; smin[wave] = kn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / kWave; ++w) { kx = smax[w] > kx ? smax[w] : kx; kn = smin[w] > kn ? smin[w] : kn; }
        if (kx) atomicMax(&mm[2 * b], kx);
        if (kn) atomicMax(&mm[2 * b + 1], kn);
    }
}

// hist: [B][kQBins], zeroed by the host
__global__ __launch_bounds__(kBlock) void select_qhist_kernel(const float* scores, int64_t N, int largest, float scale, uint32_t* hist, const uint32_t* mm = nullptr)
{
    __shared__ uint32_t lh[kQSub][kQBins];
    const int b = blockIdx.y;
    const float* s = scores + (int64_t)b * N;
    const bool lg = largest != 0;
    float lo = 0.0f;
    if (mm) image_range(mm, b, lo, scale);                     // generic maps: the image's own finite value range (select_minmax_kernel)
    for (int i = threadIdx.x; i < kQSub * kQBins; i += kBlock) (&lh[0][0])[i] = 0u;
    __syncthreads();
    const int sub = threadIdx.x & (kQSub - 1);
    const int64_t per = (N + gridDim.x - 1) / gridDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * per, i1 = i0 + per < N ? i0 + per : N;
    int64_t i = i0 + threadIdx.x;
    if (i1 > i0 && (per & 3) == 0 && (reinterpret_cast<uintptr_t>(s + i0) & 15) == 0) {
        const float4* s4 = reinterpret_cast<const float4*>(s + i0);
        const int64_t n4 = (i1 - i0) >> 2;                      // (a ragged end of the last block goes through the scalar loop below)
        int64_t j = threadIdx.x;
        auto add4 = [&](const float4& q) {
            atomicAdd(&lh[sub][qbin(q.x - lo, lg, scale)], 1u); atomicAdd(&lh[sub][qbin(q.y - lo, lg, scale)], 1u);
            atomicAdd(&lh[sub][qbin(q.z - lo, lg, scale)], 1u); atomicAdd(&lh[sub][qbin(q.w - lo, lg, scale)], 1u);
        };
        for (; j + 3 * kBlock < n4; j += 4 * kBlock) {          // four 16-byte loads in flight per thread
            const float4 q0 = s4[j], q1 = s4[j + kBlock], q2 = s4[j + 2 * kBlock], q3 = s4[j + 3 * kBlock];
            add4(q0); add4(q1); add4(q2); add4(q3);
        }
        for (; j < n4; j += kBlock) add4(s4[j]);
        i = i0 + 4 * n4 + threadIdx.x;
    }
    for (; i + 3 * kBlock < i1; i += 4 * kBlock) {            // four loads in flight per thread
        const float v0 = s[i], v1 = s[i + kBlock], v2 = s[i + 2 * kBlock], v3 = s[i + 3 * kBlock];
        atomicAdd(&lh[sub][qbin(v0 - lo, lg, scale)], 1u);
        atomicAdd(&lh[sub][qbin(v1 - lo, lg, scale)], 1u);
        atomicAdd(&lh[sub][qbin(v2 - lo, lg, scale)], 1u);
        atomicAdd(&lh[sub][qbin(v3 - lo, lg, scale)], 1u);
    }
    for (; i < i1; i += kBlock) atomicAdd(&lh[sub][qbin(s[i] - lo, lg, scale)], 1u);
    __syncthreads();
    uint32_t* H = hist + (int64_t)b * kQBins;
    for (int d = threadIdx.x; d < kQBins; d += kBlock) {
        uint32_t c = 0;
#pragma unroll
        for (int u = 0; u < kQSub; ++u) c += lh[u][d];
        if (c) atomicAdd(&H[d], c);
    }
}

// One 1024-thread block per image.  The histogram already IS the coarse sort: the number of elements in the bins above bin d is
// where bin d's elements start in the value-sorted output.  Every candidate is dropped into its bin's segment of an LDS list (one
// LDS atomic on the bin's cursor - no wave ballots, no ordering among the threads), then ranks itself inside its segment by
// counting the segment's larger (key, index) words - ~130 comparisons per candidate at 1024 bins - and goes straight to its final
// slot.  No sorting network at all: the round-3 kernel spent ~150 us in a 66-barrier bitonic sort of 8192 words per image.
constexpr int kQMaxPop = 1024;
constexpr int kQSelLds = kQCap * 8 + 2 * kQBins * 4 + 128;       // a bin holding more candidates than this (ties, constant regions) sends the image to the fallback

__global__ __launch_bounds__(kLargeThreads) void topk_qsel_kernel(const float* scores, int64_t N, int k, int largest, float scale,
                                                                 const uint32_t* hist, int32_t* out_idx, float* out_val, int* overflow,
                                                                 const uint32_t* mm = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* buf = reinterpret_cast<uint64_t*>(smem);                                   // kQCap candidates, grouped by bin
    uint32_t* start = reinterpret_cast<uint32_t*>(smem + (size_t)kQCap * 8);             // [kQBins] first slot of a bin's segment
    uint32_t* cursor = start + kQBins;                                                   // [kQBins] candidates dropped so far
    uint32_t* misc = cursor + kQBins;                                                    // 32 words
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* s = scores + (int64_t)blockIdx.x * N;
    const bool lg = largest != 0;
    float lo = 0.0f;
    if (mm) image_range(mm, (int)blockIdx.x, lo, scale);
    // thread t owns bin kQBins - 1 - t; an inclusive scan over the threads counts the elements at or above each bin
    {
        const int bin = kQBins - 1 - tid;
        const uint32_t own = hist[(int64_t)blockIdx.x * kQBins + bin];
        uint32_t incl = own;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, o, 64);
            if (lane >= o) incl += v;
        }
        if (lane == 63) misc[8 + wave] = incl;
        if (tid == 0) misc[2] = 0;
        __syncthreads();
        uint32_t base = 0;
        for (int w = 0; w < wave; ++w) base += misc[8 + w];
        incl += base;
        const uint32_t excl = incl - own;
        start[bin] = excl;
        cursor[bin] = 0;
        if (excl < (uint32_t)k && (uint32_t)k <= incl) { misc[0] = (uint32_t)bin; misc[1] = incl; }
        if (excl < (uint32_t)k && own > (uint32_t)kQMaxPop) misc[2] = 1;      // an over-full bin among those that hold candidates
        __syncthreads();
    }
    const uint32_t tb = misc[0], count = misc[1];
    if (count > (uint32_t)kQCap || misc[2]) {          // block-uniform: this image goes to the exact fallback launch
        if (tid == 0) overflow[blockIdx.x] = 1;
        return;
    }
    if (tid == 0) overflow[blockIdx.x] = 0;
    auto drop = [&](float v, int64_t i) {
        const uint32_t q = qbin(v - lo, lg, scale);
        if (q >= tb) {
            const uint32_t pos = start[q] + atomicAdd(&cursor[q], 1u);
            buf[pos] = ((uint64_t)order_key(v, lg) << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)i);
        }
    };
    if ((N & 3) == 0 && (reinterpret_cast<uintptr_t>(s) & 15) == 0) {
        // one block reads its image's whole map (512 KB at 256 x 512): four 16-byte loads in flight per thread = 64 KB per CU
        const float4* s4 = reinterpret_cast<const float4*>(s);
        const int64_t n4 = N >> 2;
        auto drop4 = [&](const float4& q, int64_t i4) { drop(q.x, 4 * i4); drop(q.y, 4 * i4 + 1); drop(q.z, 4 * i4 + 2); drop(q.w, 4 * i4 + 3); };
        int64_t base = 0;
        for (; base + 4 * kLargeThreads <= n4; base += 4 * kLargeThreads) {
            const int64_t i = base + tid;
            const float4 q0 = s4[i], q1 = s4[i + kLargeThreads], q2 = s4[i + 2 * kLargeThreads], q3 = s4[i + 3 * kLargeThreads];
            drop4(q0, i); drop4(q1, i + kLargeThreads); drop4(q2, i + 2 * kLargeThreads); drop4(q3, i + 3 * kLargeThreads);
        }
        for (int64_t i = base + tid; i < n4; i += kLargeThreads) drop4(s4[i], i);
    } else {
        int64_t base = 0;
        for (; base + 4 * kLargeThreads <= N; base += 4 * kLargeThreads) {        // four loads in flight per thread
            const int64_t i = base + tid;
            const float v0 = s[i], v1 = s[i + kLargeThreads], v2 = s[i + 2 * kLargeThreads], v3 = s[i + 3 * kLargeThreads];
            drop(v0, i); drop(v1, i + kLargeThreads); drop(v2, i + 2 * kLargeThreads); drop(v3, i + 3 * kLargeThreads);
        }
        for (int64_t i = base + tid; i < N; i += kLargeThreads) drop(s[i], i);
    }
    __syncthreads();
    // rank inside the bin's segment; neighbouring slots share a segment, so most of a wave's reads are broadcasts
    for (uint32_t p = (uint32_t)tid; p < count; p += kLargeThreads) {
        const uint64_t me = buf[p];
        const uint32_t q = qbin(key_to_float((uint32_t)(me >> 32), lg) - lo, lg, scale);
        const uint32_t a = start[q], b = a + cursor[q];
        uint32_t rank = a;
        for (uint32_t t = a; t < b; ++t) rank += buf[t] > me ? 1u : 0u;
        if (rank < (uint32_t)k) {
            out_idx[(int64_t)blockIdx.x * k + rank] = (int32_t)(0xFFFFFFFFu - (uint32_t)me);
            if (out_val) out_val[(int64_t)blockIdx.x * k + rank] = key_to_float((uint32_t)(me >> 32), lg);
        }
    }
}

// ---- large-k WITHOUT the score map: sampled threshold -> candidate emission in the scorer -> list select ---------------------------------
// The map-writing scorer pays 16 % for a 5 % larger stream (mixed read / write traffic).  Instead: (1) acq_sample_thr_kernel scores a sample of
// the image - sample_locs jittered locations of 4 * sample_gpl consecutive pixels (spatially correlated maps give about one independent draw
// per location) - and takes the edge of the quantised bin at which the sample's count from the top reaches mult * k / N of the sample as the
// image's threshold key (a conservative guess: about mult * k pixels pass); (2) the scorer (acq_kernel<..., EMIT>) writes only the pixels at or
// beyond that key, as (key, index) words into per-wave segments (~1 B / pixel instead of 4); (3) topk_lsel_kernel histograms the image's lists
// into kLBins bins spanning only [threshold, end of the score range] (the lists hold nothing else: bins ~30x finer than topk_qsel_kernel's for
// the same LDS), finds the bin of the k-th element, drops that bin and the ones above into LDS and ranks every candidate inside its bin.  The
// RESULT never depends on the sample: an image whose lists hold fewer than k words (the sample misled), or whose candidates do not fit (ties),
// is redone exactly by its own block (score map + one-block radix select, see topk_lsel_kernel).
constexpr int kSampleThreads = 512;
constexpr int kLBins = 4096;

template <int CMAX>
__global__ __launch_bounds__(kSampleThreads) void acq_sample_thr_kernel(AcqParams p, float qscale, uint32_t want, uint32_t* tkey_out)
{
    __shared__ uint32_t lh[kQBins];
    __shared__ uint32_t wsum[kSampleThreads / kWave];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const bool largest = p.strategy != PP_ACQ_MARGIN;
    const float fill = largest ? 0.0f : 1.0f;
    const float* base = p.logits + (int64_t)img * p.sB;
    const uint8_t* excl = p.exclude ? p.exclude + (int64_t)img * p.N : nullptr;
    for (int i = tid; i < kQBins; i += kSampleThreads) lh[i] = 0u;
    __syncthreads();
    const int gpl = p.sample_gpl, groups = p.sample_locs * gpl;
    const int64_t stride = p.N / p.sample_locs;                    // >= 4 * gpl pixels (host)
    const uint32_t slots = (uint32_t)(stride / (4 * gpl));
#pragma unroll 1
    for (int g = tid; g < groups; g += kSampleThreads) {
        const uint32_t loc = (uint32_t)(g / gpl);
        const uint32_t h = (loc * 2654435761u + (uint32_t)img * 40503u + 0x9E3779B9u) >> 7;
        const int64_t pix0 = ((int64_t)loc * stride + (int64_t)(h % slots) * (4 * gpl) + (g - (int)loc * gpl) * 4) & ~3ll;
        float x[4][CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(base + (int64_t)c * p.sC + pix0);
            x[0][c] = v.x; x[1][c] = v.y; x[2][c] = v.z; x[3][c] = v.w;
        }
        const uint32_t ex = excl ? *reinterpret_cast<const uint32_t*>(excl + pix0) : 0u;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            float sc = pixel_score_fast<CMAX, true>(x[v], p.C, p.strategy);
            if ((ex >> (8 * v)) & 0xFFu) sc = fill;
            atomicAdd(&lh[qbin(sc, largest, qscale)], 1u);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    // thread t owns OWN bins from the top: kQBins - 1 - OWN t - u; inclusive scan over the threads
    constexpr int OWN = kQBins / kSampleThreads;
    uint32_t own[OWN], tot = 0;
#pragma unroll
    for (int u = 0; u < OWN; ++u) { own[u] = lh[kQBins - 1 - (OWN * tid + u)]; tot += own[u]; }
    uint32_t incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t cum = incl - tot;
    for (int w = 0; w < wave; ++w) cum += wsum[w];
#pragma unroll
    for (int u = 0; u < OWN; ++u) {
        const uint32_t before = cum;
        cum += own[u];
        if (before < want && want <= cum) {
            const int qb = kQBins - 1 - (OWN * tid + u);                // qbin value: every score with qbin >= qb should pass
            const int qi = largest ? qb : kQBins - 1 - qb;             // linear bin of the score
            const float edge = largest ? (float)qi / qscale : (float)(qi + 1) / qscale;      // the bin's far edge
            uint32_t tk = order_key(edge, largest);
            if (qb == 0 || tk < 1u) tk = 1u;                            // the last bin: everything passes
            tkey_out[img] = tk;
        }
    }
}

// bins of the list select: linear in the distance from the image's threshold towards the selected end of the score range
__device__ __forceinline__ uint32_t lbin(uint32_t key, bool lg, float tv, float lscale)
{
    const float s = key_to_float(key, lg);
    if (s != s) return lg ? (uint32_t)(kLBins - 1) : 0u;
    const float t = (lg ? s - tv : tv - s) * lscale;
    return (uint32_t)(t >= (float)(kLBins - 1) ? kLBins - 1 : (t > 0.0f ? (int)t : 0));
}

// One 1024-thread block per image; cnt / list: the scorer's per-wave segments (nseg segments, segsz entries apart).  range: the scorers'
// value range (ln C or 1).  An image whose lists do not do (fewer than k words: the sample misled; candidates that do not fit: ties) is
// redone exactly by its own block, right here: the score map of the image - the default scorer's arithmetic, written over the image's own,
// by then consumed, list segments - and the one-block radix select on it.  Slow (one block scores 10 MB of logits) and rare; nothing is
// launched for it, so the usual case pays nothing (two idle launches - a map kernel over a flagged list and topk_large_kernel - cost 9 us).
template <int CMAX>
__global__ __launch_bounds__(kLargeThreads) void topk_lsel_kernel(AcqParams p, const uint64_t* list, const uint32_t* cnt, const uint32_t* tkey,
                                                                 int64_t eimg, int nseg, int segsz, int k, int largest, float range,
                                                                 int32_t* out_idx, float* out_val)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* buf = reinterpret_cast<uint64_t*>(smem);                                   // kQCap candidates, grouped by bin
    uint32_t* start = reinterpret_cast<uint32_t*>(smem + (size_t)kQCap * 8);             // [kLBins] first slot of a bin's segment
    uint32_t* hist = start + kLBins;                                                     // [kLBins] bin counts, then the drop cursors
    uint32_t* misc = hist + kLBins;                                                      // 32 words
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool lg = largest != 0;
    const uint64_t* L = list + (int64_t)blockIdx.x * eimg;
    const uint32_t* C = cnt + (int64_t)blockIdx.x * nseg;
    const uint32_t tk = tkey[blockIdx.x];
    float tv = key_to_float(tk, lg);
    if (tk == 1u) tv = lg ? 0.0f : range;                              // "everything passes"
    if (tv != tv) tv = lg ? range : 0.0f;
    if (lg && tv < 0.0f) tv = 0.0f;
    const float span = lg ? range - tv : tv;
    const float lscale = (float)kLBins / (span > 1e-12f ? span : 1e-12f);
    constexpr int OWN = kLBins / kLargeThreads;
#pragma unroll
    for (int u = 0; u < OWN; ++u) hist[tid + u * kLargeThreads] = 0u;
    if (tid < 4) misc[tid] = 0u;
    __syncthreads();
    // A wave takes kSegW consecutive segments of every 16 * kSegW: their counts in one load, then one load per segment in flight (a segment
    // usually holds fewer than 64 words; the rest of a fuller one is walked afterwards).  With <= 16 * kSegW segments per image (256 x 512 at
    // 8 pixels per thread: exactly) the words stay in registers between the histogram pass and the drop pass: the lists are read once
    // (PMC: 61.9 -> ~35 MB fetched per launch).
    constexpr int kSegW = 16;
    const bool one_chunk = nseg <= (kLargeThreads / 64) * kSegW;
    uint32_t c[kSegW];
    uint64_t v[kSegW];
    auto load_chunk = [&](int c0) {
        const int s0 = c0 + wave * kSegW;
        const uint32_t mine = (lane < kSegW && s0 + lane < nseg) ? C[s0 + lane] : 0u;
#pragma unroll
        for (int u = 0; u < kSegW; ++u) c[u] = (uint32_t)__shfl((int)mine, u, 64);
#pragma unroll
        for (int u = 0; u < kSegW; ++u) v[u] = (uint32_t)lane < c[u] ? L[(int64_t)(s0 + u) * segsz + lane] : 0ull;
    };
    auto walk_chunk = [&](int c0, auto&& fn) {
        const int s0 = c0 + wave * kSegW;
#pragma unroll
        for (int u = 0; u < kSegW; ++u)
            if ((uint32_t)lane < c[u]) fn(v[u]);
#pragma unroll
        for (int u = 0; u < kSegW; ++u)
            for (uint32_t e = (uint32_t)lane + 64u; e < c[u]; e += 64u) fn(L[(int64_t)(s0 + u) * segsz + e]);
    };
    auto sweep = [&](bool reload, auto&& fn) {
        for (int c0 = 0; c0 < nseg; c0 += (kLargeThreads / 64) * kSegW) {
            if (reload) load_chunk(c0);
            walk_chunk(c0, fn);
        }
    };
    sweep(true, [&](uint64_t w) { atomicAdd(&hist[lbin((uint32_t)(w >> 32), lg, tv, lscale)], 1u); });
    __syncthreads();
    {
        // thread t owns bins kLBins - 1 - OWN t - u (from the top)
        uint32_t own[OWN], tot = 0;
#pragma unroll
        for (int u = 0; u < OWN; ++u) { own[u] = hist[kLBins - 1 - (OWN * tid + u)]; tot += own[u]; }
        uint32_t incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, o, 64);
            if (lane >= o) incl += v;
        }
        if (lane == 63) misc[8 + wave] = incl;
        __syncthreads();
        uint32_t cum = incl - tot;
        for (int w = 0; w < wave; ++w) cum += misc[8 + w];
#pragma unroll
        for (int u = 0; u < OWN; ++u) {
            const int bin = kLBins - 1 - (OWN * tid + u);
            const uint32_t excl = cum;
            cum += own[u];
            start[bin] = excl;
            hist[bin] = 0u;                                                    // from here on: the bin's drop cursor
            if (excl < (uint32_t)k && (uint32_t)k <= cum) { misc[0] = (uint32_t)bin; misc[1] = cum; misc[3] = 1u; }
            if (excl < (uint32_t)k && own[u] > (uint32_t)kQMaxPop) misc[2] = 1;   // an over-full bin among those that hold candidates
        }
        __syncthreads();
    }
    const uint32_t tb = misc[0], count = misc[1];
    // fewer than k words in the lists (misc[3] unset: the sampled threshold was too high), or no room: the exact fallback
    if (!misc[3] || count > (uint32_t)kQCap || misc[2]) {              // block-uniform
        const int img = blockIdx.x;
        const float fill = lg ? 0.0f : 1.0f;
        const float* base = p.logits + (int64_t)img * p.sB;
        const uint8_t* excl = p.exclude ? p.exclude + (int64_t)img * p.N : nullptr;
        float* fmap = reinterpret_cast<float*>(const_cast<uint64_t*>(L));          // N floats <= eimg words
        const float qscale = (float)kQBins / range;
        __syncthreads();                                                             // (the list passes' bins are read no more)
        hist[tid] = 0u;                                                              // kQBins == kLargeThreads words
        __syncthreads();
        auto score1 = [&](const float (&xx)[CMAX]) -> float {       // (the strategy-specialised scorers: the same bits as the generic one, fewer registers)
            if (p.strategy == PP_ACQ_ENTROPY) return pixel_score_fast<CMAX, true, PP_ACQ_ENTROPY>(xx, p.C, p.strategy);
            if (p.strategy == PP_ACQ_LEAST_CONFIDENCE) return pixel_score_fast<CMAX, true, PP_ACQ_LEAST_CONFIDENCE>(xx, p.C, p.strategy);
            return pixel_score_fast<CMAX, true, PP_ACQ_MARGIN>(xx, p.C, p.strategy);
        };
        constexpr int PXF = CMAX > 19 ? 1 : 2;                       // pixels per thread and pass: PXF x CMAX registers under the 1024-thread budget
        for (int64_t pix = (int64_t)tid * PXF; pix < p.N; pix += kLargeThreads * PXF) {  // (flat planes, N % 4 == 0: the emitting scorer's layout)
            float x[PXF][CMAX], sc[PXF];
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                if constexpr (PXF == 2) {
                    const f32x2_nt v = *reinterpret_cast<const f32x2_nt*>(base + (int64_t)c * p.sC + pix);
                    x[0][c] = v.x; x[1][c] = v.y;
                } else {
                    x[0][c] = base[(int64_t)c * p.sC + pix];
                }
            }
#pragma unroll
            for (int u = 0; u < PXF; ++u) {
                sc[u] = score1(x[u]);
                if (excl && excl[pix + u]) sc[u] = fill;
                fmap[pix + u] = sc[u];
                atomicAdd(&hist[qbin(sc[u], lg, qscale)], 1u);
                __builtin_amdgcn_sched_barrier(0);                  // one pixel's temporaries at a time
            }
        }
        __syncthreads();
        // First the quantised select over the whole score range on the map just written (what topk_qsel_kernel does: kQBins bins were
        // counted while scoring): an image that is here because the sample misled - not because of ties - is done in two more passes
        // over its 0.5 MB map; only ties / constant regions go on to the radix select (one bad sample would otherwise cost 1.4 ms).
        {
            const int bin = kQBins - 1 - tid;                       // kLargeThreads == kQBins
            const uint32_t own = hist[bin];
            uint32_t incl = own;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t x2 = (uint32_t)__shfl_up((int)incl, o, 64);
                if (lane >= o) incl += x2;
            }
            if (lane == 63) misc[8 + wave] = incl;
            if (tid == 0) { misc[2] = 0u; misc[3] = 0u; }
            __syncthreads();
            uint32_t pre = 0;
            for (int w = 0; w < wave; ++w) pre += misc[8 + w];
            incl += pre;
            const uint32_t excl2 = incl - own;
            __syncthreads();                                        // (every thread has read its bin: the words become the drop cursors)
            start[bin] = excl2;
            hist[bin] = 0u;
            if (excl2 < (uint32_t)k && (uint32_t)k <= incl) { misc[0] = (uint32_t)bin; misc[1] = incl; misc[3] = 1u; }
            if (excl2 < (uint32_t)k && own > (uint32_t)kQMaxPop) misc[2] = 1u;
            __syncthreads();
        }
        if (misc[3] && misc[1] <= (uint32_t)kQCap && !misc[2]) {     // block-uniform
            const uint32_t tb2 = misc[0], cnt2 = misc[1];
            for (int64_t pix = tid; pix < p.N; pix += kLargeThreads) {
                const float sc = fmap[pix];
                const uint32_t q = qbin(sc, lg, qscale);
                if (q >= tb2) buf[start[q] + atomicAdd(&hist[q], 1u)] = ((uint64_t)order_key(sc, lg) << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)pix);
            }
            __syncthreads();
            for (uint32_t e = (uint32_t)tid; e < cnt2; e += kLargeThreads) {
                const uint64_t me = buf[e];
                const uint32_t q = qbin(key_to_float((uint32_t)(me >> 32), lg), lg, qscale);
                const uint32_t a = start[q], b = a + hist[q];
                uint32_t rank = a;
                for (uint32_t t = a; t < b; ++t) rank += buf[t] > me ? 1u : 0u;
                if (rank < (uint32_t)k) {
                    out_idx[(int64_t)img * k + rank] = (int32_t)(0xFFFFFFFFu - (uint32_t)me);
                    if (out_val) out_val[(int64_t)img * k + rank] = key_to_float((uint32_t)(me >> 32), lg);
                }
            }
            return;
        }
        __syncthreads();
        uint32_t* rh = reinterpret_cast<uint32_t*>(smem);                           // 256 + 64 words, then P 64-bit words (P <= 8192: k <= 7281)
        int P = 1;
        while (P < k) P <<= 1;
        topk_large_body(fmap, p.N, k, lg, rh, rh + 256, reinterpret_cast<uint64_t*>(smem + (256 + 64) * 4), P, out_idx + (int64_t)img * k,
                        out_val ? out_val + (int64_t)img * k : nullptr);
        return;
    }
    sweep(!one_chunk, [&](uint64_t w) {
        const uint32_t q = lbin((uint32_t)(w >> 32), lg, tv, lscale);
        if (q >= tb) buf[start[q] + atomicAdd(&hist[q], 1u)] = w;
    });
    __syncthreads();
    for (uint32_t p = (uint32_t)tid; p < count; p += kLargeThreads) {
        const uint64_t me = buf[p];
        const uint32_t q = lbin((uint32_t)(me >> 32), lg, tv, lscale);
        const uint32_t a = start[q], b = a + hist[q];
        uint32_t rank = a;
        for (uint32_t t = a; t < b; ++t) rank += buf[t] > me ? 1u : 0u;
        if (rank < (uint32_t)k) {
            out_idx[(int64_t)blockIdx.x * k + rank] = (int32_t)(0xFFFFFFFFu - (uint32_t)me);
            if (out_val) out_val[(int64_t)blockIdx.x * k + rank] = key_to_float((uint32_t)(me >> 32), lg);
        }
    }
}
constexpr int kLSelLds = kQCap * 8 + 2 * kLBins * 4 + 128;

// ---- host side ---------------------------------------------------------------------------------------
// Sum over T forward passes of softmax(logits[t]) per pixel and of the strategy's score of each pass (the MC-dropout
// branch, query.py:181-187: `uc_map += uc_map_; prob += prob_`), scaled: p_c = exp(x_c - m) / S and the score formulas in
// the reference's operation order (libm expf / logf), one thread per pixel, any strides.
__global__ __launch_bounds__(kBlock) void softmax_sum_kernel(const float* logits, int T, int C, int W, int64_t N, int64_t sT, int64_t sC,
                                                            int64_t sH, int64_t sW, float* out, float* uc_out, int strategy, float scale,
                                                            int accumulate)
{
    const int64_t pix = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (pix >= N) return;
    const int64_t hh = pix / W, ww = pix - hh * W;
    const float* base = logits + hh * sH + ww * sW;
    // per-class sums live in registers, PP_ACQ_MAX_CLASSES at a time: wider heads repeat the T passes per chunk of classes (the
    // per-pass maximum and sum are recomputed; the strategy's score is accumulated in the first chunk's round only)
    float uc = 0.0f;
    for (int c0 = 0; c0 < C; c0 += PP_ACQ_MAX_CLASSES) {
        const int cn = C - c0 < PP_ACQ_MAX_CLASSES ? C - c0 : PP_ACQ_MAX_CLASSES;
        float acc[PP_ACQ_MAX_CLASSES];
        for (int c = 0; c < cn; ++c) acc[c] = 0.0f;
        for (int t = 0; t < T; ++t) {
            const float* xt = base + (int64_t)t * sT;
            float m = xt[0];
            for (int c = 1; c < C; ++c) m = fmaxf(m, xt[(int64_t)c * sC]);
            float S = 0.0f;
            for (int c = 0; c < C; ++c) S += expf(xt[(int64_t)c * sC] - m);
            if (c0 == 0) {
                float ent = 0.0f, t1 = -INFINITY, t2 = -INFINITY;
                for (int c = 0; c < C; ++c) {
                    const float pc = expf(xt[(int64_t)c * sC] - m) / S;
                    if (c < cn) acc[c] += pc;
                    ent += (-pc) * logf(pc);
                    t2 = fmaxf(t2, fminf(t1, pc));
                    t1 = fmaxf(t1, pc);
                }
                uc += strategy == PP_ACQ_ENTROPY ? ent : (strategy == PP_ACQ_LEAST_CONFIDENCE ? 1.0f - t1 : fabsf(t1 - t2));
            } else {
                for (int c = 0; c < cn; ++c) acc[c] += expf(xt[(int64_t)(c0 + c) * sC] - m) / S;
            }
        }
        if (out)
            for (int c = 0; c < cn; ++c) {
                const int64_t o = (int64_t)(c0 + c) * N + pix;
                out[o] = accumulate ? fmaf(scale, acc[c], out[o]) : scale * acc[c];
            }
    }
    if (uc_out) uc_out[pix] = accumulate ? fmaf(scale, uc, uc_out[pix]) : scale * uc;
}

// (k <= N < 2^31 is checked by every caller: for k above 2^30 the doubling used to overflow and never end - found by tests/test_abi_asan.py)
static int next_pow2(int64_t v) { int64_t p = 1; while (p < v && p < (1ll << 30)) p <<= 1; return (int)p; }

struct Plan {
    bool nhwc = false;    // dense channels-last input: LDS-transposed path
    bool vec4;            // flat float4 path
    int ppt;              // pixels per thread
    int blocks_per_image;
    int waves_per_image;
    bool xcd = false;     // XCD-contiguous block order (flat float4 path, class planes >= 4 MB)
};

static bool is_flat_vec4(const float* logits, const uint8_t* exclude, const float* out_map, int64_t H, int64_t W,
                         int64_t sB, int64_t sC, int64_t sH, int64_t sW)
{
    const int64_t N = H * W;
    return sW == 1 && (sH == W || H == 1) && N % 4 == 0 && sC % 4 == 0 && sB % 4 == 0 &&
           (reinterpret_cast<uintptr_t>(logits) & 15) == 0 && (reinterpret_cast<uintptr_t>(exclude) & 3) == 0 &&
           (reinterpret_cast<uintptr_t>(out_map) & 15) == 0;
}

static bool is_dense_nhwc(const float* logits, int64_t C, int64_t H, int64_t W, int64_t sB, int64_t sC, int64_t sH, int64_t sW)
{
    return C > 1 && C <= 32 && sC == 1 && sW == C && (sH == W * C || H == 1) && (H * W) % 4 == 0 && sB % 4 == 0 &&
           (reinterpret_cast<uintptr_t>(logits) & 15) == 0;
}

// Deterministic in (B, N, vec4) so that pp_acq_workspace_bytes can size the candidate buffer.
static Plan make_plan(int64_t B, int64_t N, bool vec4, bool force_ppt4 = false)
{
    Plan pl;
    pl.vec4 = vec4;
    // want >= ~2048 waves in flight (256 CUs x 8) before growing the per-thread tile
    // measured on MI355X (profiles/r01_acq_tuning.txt): 8 pixels/thread at 3 waves/SIMD streams at 5.95 TB/s;
    // 16 pixels/thread spills, 4 waves/SIMD spills.
    const int64_t waves8 = B * cdiv(N, (int64_t)kBlock * 8) * (kBlock / kWave);
    pl.ppt = (waves8 >= 2048 && !force_ppt4) ? 8 : 4;
    // class planes of 4 MB and more: XCD-contiguous block order (1024 x 2048, least confidence, B = 8: 0.655 -> 0.677 of 8 TB/s; on
    // 512 KB planes it costs 5 %, on 2 MB planes it is neutral: profiles/r04_acq_layout.txt)
    pl.xcd = vec4 && g_tune_xcd != 1 && (g_tune_xcd == 2 || N * 4 >= (4ll << 20));
    if (g_tune_ppt && vec4 && !force_ppt4) pl.ppt = g_tune_ppt;
    pl.blocks_per_image = (int)cdiv(N, (int64_t)kBlock * pl.ppt);
    pl.waves_per_image = pl.blocks_per_image;          // candidate lists per image: one per BLOCK since round 5 (block_emit_topk)
    return pl;
}

static size_t merge_ws_bytes(int64_t B, int64_t n_cand, int64_t k)
{
    // ping-pong: level-0 list + the (smaller) next level
    size_t a = align_up((size_t)B * n_cand * 8, 256);
    size_t b = align_up((size_t)B * cdiv(n_cand, kMergeChunk) * k * 8, 256);
    return a + b;
}

static int run_merge(uint64_t* cand, int64_t n_cand, uint64_t* other, int64_t B, int k, int largest,
                     int32_t* out_idx, float* out_val, hipStream_t st)
{
    uint64_t* cur = cand;
    uint64_t* nxt = other;
    int64_t n = n_cand;
    for (;;) {
        const int nch = (int)cdiv(n, kMergeChunk);
        dim3 grid(nch, (unsigned)B);
        if (nch == 1) {
            hipLaunchKernelGGL(cand_merge_kernel, grid, dim3(kBlock), 0, st, cur, n, (uint64_t*)nullptr, out_idx,
                               out_val, k, largest, g_reduce_mode);
            return check_launch("cand_merge_kernel");
        }
        hipLaunchKernelGGL(cand_merge_kernel, grid, dim3(kBlock), 0, st, cur, n, nxt, (int32_t*)nullptr,
                           (float*)nullptr, k, largest, g_reduce_mode);
        if (int rc = check_launch("cand_merge_kernel")) return rc;
        n = (int64_t)nch * k;
        uint64_t* t = cur; cur = nxt; nxt = t;
    }
}

static int g_large_multiblock = 1;     // 0: the one-block-per-image radix select (pp_debug_set_reduce_mode bit 8), for A/B
static int g_large_q = 1;              // 0: no quantised-histogram select (pp_debug_set_reduce_mode bit 9), for A/B
static int g_generic_q = 1;            // 0: generic maps through the radix select (pp_debug_set_reduce_mode bit 22), for A/B

static size_t large_ws_bytes(int64_t B, int64_t k)
{
    const int P = next_pow2(k);
    const size_t g = P <= kLargeLdsMaxP ? 256 : align_up((size_t)B * P * 8, 256);
    // histograms: four 256-bin radix passes, or the 1024 linear bins of the quantised select (the same bytes) + its overflow flags
    // + the per-image value range of the generic quantised select
    return g + align_up((size_t)B * kSelPasses * kSelBins * 4, 256) + align_up((size_t)B * 4, 256) + align_up((size_t)B * 8, 256);
}

// The scorers' value range as bins per unit score (0: unknown - the generic radix select)
static float score_qscale(int strategy, int64_t C)
{
    const float range = strategy == PP_ACQ_ENTROPY ? logf((float)(C > 1 ? C : 2)) : 1.0f;
    return (float)kQBins / range;
}

// the quantised select's conditions and its histogram's place in the large-k scratch (also asked by the scorer launches that fill it)
static bool large_q_ok(int64_t B, int64_t k, float qscale)
{
    return qscale > 0.0f && g_large_q && g_large_multiblock && B <= 65535 && k + (k / 8 > 256 ? k / 8 : 256) <= kQCap;
}
static uint32_t* large_hist(void* ws, int64_t B, int64_t k)
{
    const int P = next_pow2(k);
    return reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ws) + (P <= kLargeLdsMaxP ? 256 : align_up((size_t)B * P * 8, 256)));
}

// hist_done: the scorer launch already filled the (host-zeroed) quantised histogram
static int run_large(const float* map, int64_t B, int64_t N, int64_t k, int largest, void* ws,
                     int32_t* out_idx, float* out_val, hipStream_t st, float qscale = 0.0f, bool hist_done = false)
{
    const int P = next_pow2(k);
    const bool in_lds = P <= kLargeLdsMaxP;
    uint64_t* gbuf = reinterpret_cast<uint64_t*>(ws);
    uint32_t* hist = large_hist(ws, B, k);
    const size_t lds = (256 + 64) * 4 + (in_lds ? (size_t)P * 8 : 0);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topk_large_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (256 + 64) * 4 + kLargeLdsMaxP * 8);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topk_large_sel_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (256 + 64) * 4 + kLargeLdsMaxP * 8);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topk_qsel_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, kQSelLds);
        attr_set = true;
    }
    // room for the threshold bin's population beside the k picks: the quantised select - over the scorers' known range, or (generic maps,
    // qscale == 0) over every image's own finite range, taken in one more pass: three passes over the map instead of the radix select's five
    const bool generic_q = qscale <= 0.0f && g_generic_q && large_q_ok(B, k, 1.0f);
    if (large_q_ok(B, k, qscale) || generic_q) {
        static_assert(kQBins * 4 == kSelPasses * kSelBins * 4, "the two histogram layouts share their workspace slot");
        int* flags = reinterpret_cast<int*>(reinterpret_cast<char*>(hist) + align_up((size_t)B * kQBins * 4, 256));
        uint32_t* mm = nullptr;
        if (generic_q) {
            mm = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(flags) + align_up((size_t)B * 4, 256));
            if (hipMemsetAsync(mm, 0, (size_t)B * 8, st) != hipSuccess) return fail(PP_ERR_LAUNCH, "topk: memset failed");
            int64_t bpi = cdiv(2048, B);
            const int64_t by_size = cdiv(N, 4096);
            if (bpi > by_size) bpi = by_size;
            if (bpi < 1) bpi = 1;
            hipLaunchKernelGGL(select_minmax_kernel, dim3((unsigned)bpi, (unsigned)B), dim3(kBlock), 0, st, map, N, mm);
            if (int rc = check_launch("select_minmax_kernel")) return rc;
            qscale = 1.0f;
        }
        if (!hist_done) {
            if (hipMemsetAsync(hist, 0, (size_t)B * kQBins * 4, st) != hipSuccess) return fail(PP_ERR_LAUNCH, "topk: memset failed");
            int64_t bpi = cdiv(2048, B);
            const int64_t by_size = cdiv(N, 4096);
            if (bpi > by_size) bpi = by_size;
            if (bpi < 1) bpi = 1;
            hipLaunchKernelGGL(select_qhist_kernel, dim3((unsigned)bpi, (unsigned)B), dim3(kBlock), 0, st, map, N, largest, qscale, hist,
                               (const uint32_t*)mm);
            if (int rc = check_launch("select_qhist_kernel")) return rc;
        }
        hipLaunchKernelGGL(topk_qsel_kernel, dim3((unsigned)B), dim3(kLargeThreads), kQSelLds, st, map, N, (int)k, largest, qscale,
                           hist, out_idx, out_val, flags, (const uint32_t*)mm);
        if (int rc = check_launch("topk_qsel_kernel")) return rc;
        // images whose candidates did not fit (ties at the threshold, constant maps): the exact one-block radix select; the others return at once
        hipLaunchKernelGGL(topk_large_kernel, dim3((unsigned)B), dim3(kLargeThreads), lds, st, map, N, (int)k, largest,
                           in_lds ? (uint64_t*)nullptr : gbuf, P, out_idx, out_val, (const int*)flags);
        return check_launch("topk_large_kernel");
    }
    if (!g_large_multiblock || B > 65535) {
        hipLaunchKernelGGL(topk_large_kernel, dim3((unsigned)B), dim3(kLargeThreads), lds, st, map, N, (int)k, largest,
                           in_lds ? (uint64_t*)nullptr : gbuf, P, out_idx, out_val, (const int*)nullptr);
        return check_launch("topk_large_kernel");
    }
    if (hipMemsetAsync(hist, 0, (size_t)B * kSelPasses * kSelBins * 4, st) != hipSuccess) return fail(PP_ERR_LAUNCH, "topk: memset failed");
    // >= ~2048 blocks per pass, at least 4096 elements per block
    int64_t bpi = cdiv(2048, B);
    const int64_t by_size = cdiv(N, 4096);
    if (bpi > by_size) bpi = by_size;
    if (bpi < 1) bpi = 1;
    for (int pass = 0; pass < kSelPasses; ++pass) {
        hipLaunchKernelGGL(select_hist_kernel, dim3((unsigned)bpi, (unsigned)B), dim3(kBlock), 0, st, map, N, (int)k, largest, pass, hist);
        if (int rc = check_launch("select_hist_kernel")) return rc;
    }
    hipLaunchKernelGGL(topk_large_sel_kernel, dim3((unsigned)B), dim3(kLargeThreads), lds, st, map, N, (int)k, largest, hist,
                       in_lds ? (uint64_t*)nullptr : gbuf, P, out_idx, out_val);
    return check_launch("topk_large_sel_kernel");
}

constexpr int kAcqOcc21 = 2, kAcqOcc11 = 3;     // defaults of the C = 21 / C = 11 scorers (see launch_acq)

template <int CMAX, bool EXACT>
static int launch_acq(const AcqParams& p, const Plan& pl, int64_t B, hipStream_t st)
{
    EventScope ev(st);
    dim3 grid((unsigned)(B * pl.blocks_per_image)), block(kBlock);
    // the non-default scorers (reference-order, from-prob) are only built for the 4-pixel tile
    const bool alt = p.from_prob || exact_formula();
    if (pl.nhwc) {
        if constexpr (EXACT && CMAX <= 21) {
            // asynchronous LDS-DMA variant for the three dataset class counts (tuning value 8 keeps the synchronous kernel)
            if (!alt && g_tune_occ != 8) {
                if (pl.ppt == 8) hipLaunchKernelGGL((acq_nhwc_dma_kernel<CMAX, 8>), grid, block, 0, st, p);
                else             hipLaunchKernelGGL((acq_nhwc_dma_kernel<CMAX, 4>), grid, block, 0, st, p);
                return check_launch("acq_nhwc_dma_kernel");
            }
        }
        if constexpr (CMAX <= 32) {
            if (p.from_prob)          hipLaunchKernelGGL((acq_nhwc_kernel<CMAX, EXACT, 4, 2>), grid, block, 0, st, p);
            else if (exact_formula()) hipLaunchKernelGGL((acq_nhwc_kernel<CMAX, EXACT, 4, 1>), grid, block, 0, st, p);
            else if (pl.ppt == 8)     hipLaunchKernelGGL((acq_nhwc_kernel<CMAX, EXACT, 8, 0>), grid, block, 0, st, p);
            else                      hipLaunchKernelGGL((acq_nhwc_kernel<CMAX, EXACT, 4, 0>), grid, block, 0, st, p);
            return check_launch("acq_nhwc_kernel");
        }
    }
#define PP_LAUNCH_ACQ4(VEC, G)                                                                                    \
    do {                                                                                                          \
        if (p.from_prob)          hipLaunchKernelGGL((acq_kernel<CMAX, EXACT, VEC, G, 2>), grid, block, 0, st, p); \
        else if (exact_formula()) hipLaunchKernelGGL((acq_kernel<CMAX, EXACT, VEC, G, 1>), grid, block, 0, st, p); \
        else                      hipLaunchKernelGGL((acq_kernel<CMAX, EXACT, VEC, G, 0>), grid, block, 0, st, p); \
    } while (0)
    if (pl.vec4) {
        if constexpr (CMAX == 11 || CMAX == 19 || CMAX == 21) {
            // the three dataset class counts: pixels per thread x waves per SIMD chosen per count (profiles/r04_acq_layout.txt);
            // pp_debug_set_acq_tuning overrides either for A/B.  C = 21: 21 planes x 8 pixels do not fit the 170-VGPR budget of
            // 3 waves/SIMD (23 spilled registers, 0.56 of the HBM roofline on VOC 320x320) - 2 waves/SIMD, or 4 pixels per thread.
            if (!alt) {
                int occ = CMAX == 21 ? kAcqOcc21 : (CMAX == 11 ? kAcqOcc11 : 3);
                const int g = pl.ppt == 8 ? 2 : 1;
                if (g_tune_occ >= 2 && g_tune_occ <= 4) occ = g_tune_occ;
                AcqParams q = p;
                if (pl.xcd) {
                    q.nb = (int)grid.x;
                    q.xcd_per = (int)cdiv(q.nb, 8);
                    grid = dim3((unsigned)(q.xcd_per * 8));
                }
                if (q.elist) {       // large-k selection without the map: candidates beyond the sampled threshold key
#define PP_ACQ_EMIT(G, S) hipLaunchKernelGGL((acq_kernel<CMAX, EXACT, 4, G, 0, 3, S, false, true, true>), grid, block, 0, st, q)
                    if (g == 2) {
                        if (q.strategy == PP_ACQ_ENTROPY) PP_ACQ_EMIT(2, PP_ACQ_ENTROPY);
                        else if (q.strategy == PP_ACQ_LEAST_CONFIDENCE) PP_ACQ_EMIT(2, PP_ACQ_LEAST_CONFIDENCE);
                        else PP_ACQ_EMIT(2, PP_ACQ_MARGIN);
                    } else {
                        if (q.strategy == PP_ACQ_ENTROPY) PP_ACQ_EMIT(1, PP_ACQ_ENTROPY);
                        else if (q.strategy == PP_ACQ_LEAST_CONFIDENCE) PP_ACQ_EMIT(1, PP_ACQ_LEAST_CONFIDENCE);
                        else PP_ACQ_EMIT(1, PP_ACQ_MARGIN);
                    }
#undef PP_ACQ_EMIT
                    return check_launch("acq_kernel<emit>");
                }
                if (q.qhist) {       // large-k selection: map + the image's score histogram in one pass (run_large skips its histogram pass)
                    constexpr int O = CMAX == 21 ? kAcqOcc21 : (CMAX == 11 ? kAcqOcc11 : 3);
                    if (g == 2) hipLaunchKernelGGL((acq_kernel<CMAX, EXACT, 4, 2, 0, O, -1, true>), grid, block, 0, st, q);
                    else        hipLaunchKernelGGL((acq_kernel<CMAX, EXACT, 4, 1, 0, O, -1, true>), grid, block, 0, st, q);
                    return check_launch("acq_kernel<hist>");
                }
#define PP_ACQ_GO(G, O) hipLaunchKernelGGL((acq_kernel<CMAX, EXACT, 4, G, 0, O>), grid, block, 0, st, q)
#define PP_ACQ_SPEC(G, S) do { if (g_nt_off) hipLaunchKernelGGL((acq_kernel<CMAX, EXACT, 4, G, 0, 3, S, false, false>), grid, block, 0, st, q); \
                               else hipLaunchKernelGGL((acq_kernel<CMAX, EXACT, 4, G, 0, 3, S>), grid, block, 0, st, q); } while (0)
                if (g_acq_strat_spec && !g_tune_occ && !q.out_map) {            // (with the map written - large k - the generic kernel measured no slower)
                    // the strategy as a compile-time constant: the chains the other two strategies need are not computed (100-145 VGPRs
                    // instead of 168-184: entropy runs four waves per SIMD, C = 21 three instead of two); bit 10 of the tuning word: off
                    if (g == 2) {
                        if (q.strategy == PP_ACQ_ENTROPY) PP_ACQ_SPEC(2, PP_ACQ_ENTROPY);
                        else if (q.strategy == PP_ACQ_LEAST_CONFIDENCE) PP_ACQ_SPEC(2, PP_ACQ_LEAST_CONFIDENCE);
                        else PP_ACQ_SPEC(2, PP_ACQ_MARGIN);
                    } else {
                        if (q.strategy == PP_ACQ_ENTROPY) PP_ACQ_SPEC(1, PP_ACQ_ENTROPY);
                        else if (q.strategy == PP_ACQ_LEAST_CONFIDENCE) PP_ACQ_SPEC(1, PP_ACQ_LEAST_CONFIDENCE);
                        else PP_ACQ_SPEC(1, PP_ACQ_MARGIN);
                    }
                    return check_launch("acq_kernel");
                }
                if (g == 2) { if (occ == 2) PP_ACQ_GO(2, 2); else if (occ == 4) PP_ACQ_GO(2, 4); else PP_ACQ_GO(2, 3); }
                else        { if (occ == 2) PP_ACQ_GO(1, 2); else if (occ == 4) PP_ACQ_GO(1, 4); else PP_ACQ_GO(1, 3); }
#undef PP_ACQ_SPEC
#undef PP_ACQ_GO
                return check_launch("acq_kernel");
            }
        }
        if constexpr (CMAX <= 32) {
            if (pl.ppt == 8 && !alt) hipLaunchKernelGGL((acq_kernel<CMAX, EXACT, 4, 2, 0>), grid, block, 0, st, p);
            else                     PP_LAUNCH_ACQ4(4, 1);
        }
    } else {
        if (pl.ppt == 8 && !alt) hipLaunchKernelGGL((acq_kernel<CMAX, EXACT, 1, 8, 0>), grid, block, 0, st, p);
        else                     PP_LAUNCH_ACQ4(1, 4);
    }
#undef PP_LAUNCH_ACQ4
    return check_launch("acq_kernel");
}

// class counts beyond the register-resident scorers (or forced, for A/B): streamed class vector, 4 pixels per thread
static bool stream_classes(int64_t C) { return C > PP_ACQ_MAX_CLASSES || g_tune_occ == 10; }

static int launch_acq_stream(const AcqParams& p, const Plan& pl, int64_t B, hipStream_t st)
{
    EventScope ev(st);
    if (pl.ppt != 4) return fail(PP_ERR_BAD_ARG, "streamed scorer: plan with %d pixels per thread", pl.ppt);
    dim3 grid((unsigned)(B * pl.blocks_per_image)), block(kBlock);
    const int math = p.from_prob ? 2 : (exact_formula() ? 1 : 0);
#define PP_STREAM(VEC, G)                                                                                  \
    do {                                                                                                   \
        if (math == 2)      hipLaunchKernelGGL((acq_stream_kernel<VEC, G, 2>), grid, block, 0, st, p);     \
        else if (math == 1) hipLaunchKernelGGL((acq_stream_kernel<VEC, G, 1>), grid, block, 0, st, p);     \
        else                hipLaunchKernelGGL((acq_stream_kernel<VEC, G, 0>), grid, block, 0, st, p);     \
    } while (0)
    if (pl.vec4) PP_STREAM(4, 1); else PP_STREAM(1, 4);
#undef PP_STREAM
    return check_launch("acq_stream_kernel");
}

// which launches can take the histogram epilogue (acq_kernel<..., HIST>): the flat float4 form of the three dataset class counts
static int g_hist_fuse = 1;        // pp_debug_set_reduce_mode bit 10: off (A/B)
static bool acq_hist_fusable(const AcqParams& p, const Plan& pl)
{
    return g_hist_fuse && pl.vec4 && !p.from_prob && !exact_formula() && !g_tune_occ && !stream_classes(p.C) && p.out_map && !p.cand &&
           (p.C == 11 || p.C == 19 || p.C == 21);
}

// ---- the list select (sampled threshold + candidate emission) ---------------------------------------------------------------------
static int g_emit = 1;             // pp_debug_set_reduce_mode bit 11: off (A/B: the map-writing scorer + topk_qsel_kernel)
static int g_emit_mult16 = 40;     // sampled threshold aims at mult16 / 16 x k pixels passing (pp_debug_set_reduce_mode bits 12-17; 0 = default)
static int g_sample_locs = 128;    // pp_debug_set_reduce_mode bits 18-19: 128 / 64 / 256 / 512 locations per image
static int g_sample_gpl = 4;       // bits 20-21: 4 / 2 / 1 / 8 four-pixel groups per location
static bool emit_size_ok(int64_t N, int64_t k) { return N >= 16384 && k * 8 <= N; }
// Entries per image: a wave's segment holds PPT * 64 words and is followed by kSegPad unused ones, so that the segments do not all start
// on a 4 KB boundary.  (The scorer launch costs 0.365 ms without the list stores and 0.38-0.42 with them - the spread is between boxes /
// hours, not between layouts: the pad, whole-wave stores from an LDS staging list and non-temporal stores all measured the same.)
constexpr int kSegPad = 32;
static size_t emit_list_entries(int64_t N) { return align_up((size_t)N, 2048) / 256 * (256 + kSegPad); }
// first region of the large-k workspace: the score map, or the per-wave candidate segments (the exact fallback writes an image's score
// map over that image's own segments once its block has consumed them)
static size_t map_region_bytes(int64_t B, int64_t N, int64_t k)
{
    return emit_size_ok(N, k) ? align_up((size_t)B * emit_list_entries(N) * 8, 256) : align_up((size_t)B * N * 4, 256);
}
static size_t emit_extra_bytes(int64_t B, int64_t N, int64_t k)
{
    if (!emit_size_ok(N, k)) return 0;
    // threshold keys, per-wave counts
    return align_up((size_t)B * 4, 256) + align_up((size_t)B * (align_up((size_t)N, 2048) / 256) * 4, 256);
}
static bool acq_emit_ok(const AcqParams& p, const Plan& pl, int64_t B, int64_t k, float qs, bool caller_map)
{
    return g_emit && !caller_map && emit_size_ok(p.N, k) && large_q_ok(B, k, qs) && pl.vec4 && !p.from_prob && !exact_formula() &&
           !g_tune_occ && !stream_classes(p.C) && (p.C == 11 || p.C == 19 || p.C == 21);
}

static int dispatch_acq(const AcqParams& p, Plan pl, int64_t B, hipStream_t st);

// p: the map-writing launch's parameters; region0: the first workspace region (candidate segments); tail: what follows the large-k scratch
static int run_emit_select(AcqParams p, const Plan& pl, int64_t B, int64_t k, int largest, float qs, void* region0, void* tail,
                           int32_t* out_idx, float* out_val, hipStream_t st)
{
    uint32_t* tkey = reinterpret_cast<uint32_t*>(tail);
    uint32_t* ecnt = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(tail) + align_up((size_t)B * 4, 256));
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topk_lsel_kernel<11>), hipFuncAttributeMaxDynamicSharedMemorySize, kLSelLds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topk_lsel_kernel<19>), hipFuncAttributeMaxDynamicSharedMemorySize, kLSelLds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topk_lsel_kernel<21>), hipFuncAttributeMaxDynamicSharedMemorySize, kLSelLds);
        attr_set = true;
    }
    const int segsz = pl.ppt * kWave + kSegPad;
    AcqParams q = p;
    q.out_map = nullptr;
    q.tkey = tkey;
    q.elist = reinterpret_cast<uint64_t*>(region0);
    q.ecnt = ecnt;
    q.eimg = (int64_t)emit_list_entries(p.N);
    q.nseg = pl.blocks_per_image * (kBlock / kWave);
    q.segst = segsz;
    q.sample_locs = g_sample_locs;
    q.sample_gpl = g_sample_gpl;
    while ((int64_t)q.sample_locs * q.sample_gpl * 4 * 4 > p.N && q.sample_locs > 16) q.sample_locs >>= 1;      // (16384 pixels and up: >= 4 slots per location)
    const int64_t sample_n = (int64_t)q.sample_locs * q.sample_gpl * 4;
    int64_t want = (sample_n * g_emit_mult16 * k + 16 * p.N - 1) / (16 * p.N);
    if (want < 1) want = 1;
    if (want > sample_n) want = sample_n;
    const dim3 sgrid((unsigned)B), sblock(kSampleThreads), lblock(kLargeThreads);
    const float range = (float)kQBins / qs;
#define PP_BY_C(KERNEL, GRID, BLOCK, LDS, ...)                                                                 \
    switch (p.C) {                                                                                             \
        case 11: hipLaunchKernelGGL((KERNEL<11>), GRID, BLOCK, LDS, st, __VA_ARGS__); break;                   \
        case 19: hipLaunchKernelGGL((KERNEL<19>), GRID, BLOCK, LDS, st, __VA_ARGS__); break;                   \
        default: hipLaunchKernelGGL((KERNEL<21>), GRID, BLOCK, LDS, st, __VA_ARGS__); break;                   \
    }
    PP_BY_C(acq_sample_thr_kernel, sgrid, sblock, 0, q, qs, (uint32_t)want, tkey);
    if (int rc = check_launch("acq_sample_thr_kernel")) return rc;
    if (int rc = dispatch_acq(q, pl, B, st)) return rc;
    PP_BY_C(topk_lsel_kernel, sgrid, lblock, kLSelLds, q, (const uint64_t*)q.elist, (const uint32_t*)ecnt, (const uint32_t*)tkey, q.eimg, q.nseg,
            segsz, (int)k, largest, range, out_idx, out_val);
#undef PP_BY_C
    return check_launch("topk_lsel_kernel");
}

static int dispatch_acq(const AcqParams& p, Plan pl, int64_t B, hipStream_t st)
{
    if (stream_classes(p.C)) return launch_acq_stream(p, pl, B, st);
    if (p.C > 32) pl.vec4 = false;  // 64-class bucket only on the scalar path (register budget)
    if (!pl.vec4 && g_tune_occ != 9)   // (tuning value 9 forces the generic strided path for A/B)
        pl.nhwc = is_dense_nhwc(p.logits, p.C, p.N / p.W, p.W, p.sB, p.sC, p.sH, p.sW);
    if (pl.nhwc && (exact_formula() || p.from_prob) && pl.ppt != 4) pl.nhwc = false;
    switch (p.C) {
        case 11: return launch_acq<11, true>(p, pl, B, st);   // CamVid      (args.py:109-116)
        case 19: return launch_acq<19, true>(p, pl, B, st);   // Cityscapes  (args.py:89-94)
        case 21: return launch_acq<21, true>(p, pl, B, st);   // VOC 2012    (args.py:131-141)
        default: break;
    }
    if (p.C <= 32) return launch_acq<32, false>(p, pl, B, st);
    return launch_acq<64, false>(p, pl, B, st);
}

static int validate(const float* logits, int64_t B, int64_t C, int64_t H, int64_t W, int strategy)
{
    if (!logits) return fail(PP_ERR_BAD_ARG, "logits is null");
    if (B < 1 || C < 1 || H < 1 || W < 1) return fail(PP_ERR_BAD_ARG, "bad shape B=%lld C=%lld H=%lld W=%lld",
                                                        (long long)B, (long long)C, (long long)H, (long long)W);
    if (C > 0x7FFFFFFFll) return fail(PP_ERR_UNSUPPORTED, "C=%lld", (long long)C);
    if (H * W > 0x7FFFFFFFll || B * H * W / 1024 > 0x7FFFFFFFll) return fail(PP_ERR_UNSUPPORTED, "image too large");
    if (strategy < 0 || strategy > 2) return fail(PP_ERR_BAD_ARG, "unknown strategy %d", strategy);
    return PP_OK;
}

// ---- low-resolution path: host side ---------------------------------------------------------------------
struct LowresPlan {
    int ppt, tiles_x, tiles_y, waves_per_image;
    bool lds;
    size_t lds_bytes;
    int patch_cap;
};

constexpr size_t kLowresLdsMax = 54 * 1024;      // + 8.2 KB of static survivor lists = the 64 KB a block may ask for
constexpr size_t kLowresLdsSoft = 32 * 1024;     // above this the 32-row tile gives way to the 16-row tile

static void lowres_scales(int64_t h, int64_t w, int64_t H, int64_t W, int align, float& sh, float& sw)
{
    if (align) {
        sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.0f;
        sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.0f;
    } else {
        sh = (float)h / (float)H;
        sw = (float)w / (float)W;
    }
}

static LowresPlan make_lowres_plan(int64_t B, int64_t C, int64_t h, int64_t w, int64_t Hc, int64_t Wc, float sh, float sw,
                                   bool force_ppt4)
{
    LowresPlan pl;
    const int64_t tiles8 = cdiv(Wc, kWave) * cdiv(Hc, (kBlock / kWave) * 8);
    pl.ppt = (!force_ppt4 && B * tiles8 * (kBlock / kWave) >= 2048) ? 8 : 4;
    // x2 models (FPNSeg, decoders.py:101): a 32-row tile interpolates from 19 x 35 source pixels = 50.5 KB at C = 19; the 16-row
    // tile's patch (29 KB) keeps the LDS path and two more blocks per CU
    if (pl.ppt == 8 && g_tune_ppt != 8) {      // pp_debug_set_acq_tuning(., 4 | 8) forces a tile height (A/B)
        const int64_t pw8 = std::min<int64_t>(w, (int64_t)std::ceil((double)sw * (kWave - 1)) + 3);
        const int64_t ph8 = std::min<int64_t>(h, (int64_t)std::ceil((double)sh * ((kBlock / kWave) * 8 - 1)) + 3);
        if ((size_t)(ph8 * pw8 * (C | 1)) * 4 > kLowresLdsSoft) pl.ppt = 4;
    }
    if (g_tune_ppt == 4) pl.ppt = 4;
    pl.tiles_x = (int)cdiv(Wc, kWave);
    pl.tiles_y = (int)cdiv(Hc, (kBlock / kWave) * pl.ppt);
    pl.waves_per_image = pl.tiles_x * pl.tiles_y;      // candidate lists per image: one per block (tile)
    // a tile of T output pixels spans at most ceil(scale*(T-1)) + 3 source pixels (i0 of the first .. i1 of the last)
    const int64_t pw = std::min<int64_t>(w, (int64_t)std::ceil((double)sw * (kWave - 1)) + 3);
    const int64_t ph = std::min<int64_t>(h, (int64_t)std::ceil((double)sh * ((kBlock / kWave) * pl.ppt - 1)) + 3);
    const int64_t fl = ph * pw * (C | 1);
    pl.lds = (size_t)fl * 4 <= kLowresLdsMax;
    pl.patch_cap = pl.lds ? (int)fl : 0;
    pl.lds_bytes = pl.lds ? (size_t)fl * 4 : 0;
    return pl;
}

template <int CMAX, bool EXACT>
static int launch_lowres(const LowresParams& p, const LowresPlan& pl, int64_t B, hipStream_t st)
{
    EventScope ev(st);
    dim3 grid((unsigned)(B * pl.tiles_x * pl.tiles_y)), block(kBlock);
    if (exact_formula()) {        // reference operation order: 4-row variant only
        if (pl.lds) hipLaunchKernelGGL((acq_lowres_kernel<CMAX, EXACT, 4, true, 1>), grid, block, pl.lds_bytes, st, p);
        else        hipLaunchKernelGGL((acq_lowres_kernel<CMAX, EXACT, 4, false, 1>), grid, block, 0, st, p);
    } else if (EXACT && pl.lds && g_acq_strat_spec) {
        // the dataset class counts with the patch in LDS (every production shape): strategy-specialised scorer (this kernel is VALU-bound)
#define PP_LOWRES_SPEC(P, S) hipLaunchKernelGGL((acq_lowres_kernel<CMAX, EXACT, P, true, 0, S>), grid, block, pl.lds_bytes, st, p)
        if (pl.ppt == 8) {
            if (p.strategy == PP_ACQ_ENTROPY) PP_LOWRES_SPEC(8, PP_ACQ_ENTROPY);
            else if (p.strategy == PP_ACQ_LEAST_CONFIDENCE) PP_LOWRES_SPEC(8, PP_ACQ_LEAST_CONFIDENCE);
            else PP_LOWRES_SPEC(8, PP_ACQ_MARGIN);
        } else {
            if (p.strategy == PP_ACQ_ENTROPY) PP_LOWRES_SPEC(4, PP_ACQ_ENTROPY);
            else if (p.strategy == PP_ACQ_LEAST_CONFIDENCE) PP_LOWRES_SPEC(4, PP_ACQ_LEAST_CONFIDENCE);
            else PP_LOWRES_SPEC(4, PP_ACQ_MARGIN);
        }
#undef PP_LOWRES_SPEC
    } else if (pl.ppt == 8) {
        if (pl.lds) hipLaunchKernelGGL((acq_lowres_kernel<CMAX, EXACT, 8, true, 0>), grid, block, pl.lds_bytes, st, p);
        else        hipLaunchKernelGGL((acq_lowres_kernel<CMAX, EXACT, 8, false, 0>), grid, block, 0, st, p);
    } else {
        if (pl.lds) hipLaunchKernelGGL((acq_lowres_kernel<CMAX, EXACT, 4, true, 0>), grid, block, pl.lds_bytes, st, p);
        else        hipLaunchKernelGGL((acq_lowres_kernel<CMAX, EXACT, 4, false, 0>), grid, block, 0, st, p);
    }
    return check_launch("acq_lowres_kernel");
}

static int dispatch_lowres(const LowresParams& p, const LowresPlan& pl, int64_t B, hipStream_t st)
{
    if (stream_classes(p.C)) {
        EventScope ev(st);
        if (pl.ppt != 4) return fail(PP_ERR_BAD_ARG, "streamed low-resolution scorer: plan with %d rows per wave", pl.ppt);
        dim3 grid((unsigned)(B * pl.tiles_x * pl.tiles_y)), block(kBlock);
        if (exact_formula()) hipLaunchKernelGGL((acq_lowres_stream_kernel<1>), grid, block, 0, st, p);
        else                 hipLaunchKernelGGL((acq_lowres_stream_kernel<0>), grid, block, 0, st, p);
        return check_launch("acq_lowres_stream_kernel");
    }
    switch (p.C) {
        case 11: return launch_lowres<11, true>(p, pl, B, st);
        case 19: return launch_lowres<19, true>(p, pl, B, st);
        case 21: return launch_lowres<21, true>(p, pl, B, st);
        default: break;
    }
    if (p.C <= 32) return launch_lowres<32, false>(p, pl, B, st);
    return launch_lowres<64, false>(p, pl, B, st);
}

static int validate_lowres(const float* low, int64_t ldx, int64_t B, int64_t C, int64_t h, int64_t w, int64_t H, int64_t W,
                           int64_t Hc, int64_t Wc, int strategy)
{
    if (int rc = validate(low, B, C, Hc, Wc, strategy)) return rc;
    if (h < 1 || w < 1 || H < 1 || W < 1) return fail(PP_ERR_BAD_ARG, "bad low-res/full-res shape %lldx%lld -> %lldx%lld",
                                                    (long long)h, (long long)w, (long long)H, (long long)W);
    if (Hc > H || Wc > W) return fail(PP_ERR_BAD_ARG, "crop %lldx%lld exceeds the interpolated size %lldx%lld", (long long)Hc,
                                      (long long)Wc, (long long)H, (long long)W);
    if (ldx < C) return fail(PP_ERR_BAD_ARG, "ldx=%lld < C=%lld", (long long)ldx, (long long)C);
    if (B * h * w * ldx > 0x7FFFFFFFFFFFll) return fail(PP_ERR_UNSUPPORTED, "low-res tensor too large");
    return PP_OK;
}

}  // namespace pp

using namespace pp;

extern "C" {

#ifdef PP_DEBUG_KNOBS
void pp_debug_set_reduce_mode(int mode)
{
    g_large_multiblock = (mode & 256) ? 0 : 1;      // bit 8: large-k selection through the one-block-per-image radix select (A/B)
    g_large_q = (mode & 512) ? 0 : 1;               // bit 9: no quantised-histogram select where the score range is known (A/B)
    g_hist_fuse = (mode & 1024) ? 0 : 1;            // bit 10: the score histogram in its own pass over the map (select_qhist_kernel), not in the scorer launch
    g_generic_q = (mode & (1 << 22)) ? 0 : 1;       // bit 22: maps without a known range (pp_topk_select) through the four-pass radix select
    g_emit = (mode & 2048) ? 0 : 1;                 // bit 11: no sampled-threshold candidate emission (the map-writing scorer + topk_qsel_kernel)
    g_emit_mult16 = ((mode >> 12) & 63) ? ((mode >> 12) & 63) : 40;    // bits 12-17: the sample aims at this / 16 x k passing pixels
    { static const int locs[4] = {128, 64, 256, 512}, gpl[4] = {4, 2, 1, 8};
      g_sample_locs = locs[(mode >> 18) & 3]; g_sample_gpl = gpl[(mode >> 20) & 3]; }      // bits 18-21: the sample's shape
    mode &= 255;
    g_reduce_mode = (mode >= 0 && mode <= 2) ? mode : 0;
}
#endif

#ifdef PP_DEBUG_KNOBS
void pp_debug_set_exact_formula(int on) { g_exact_formula = on ? 1 : 0; }
#endif

#ifdef PP_DEBUG_KNOBS
void pp_debug_set_acq_tuning(int occ, int ppt)
{
    g_tune_xcd = (occ >> 8) & 3;
    g_acq_strat_spec = (occ >> 10) & 1 ? 0 : 1;
    g_nt_off = (occ >> 11) & 1;
    occ &= 0xFF;
    g_tune_occ = (occ == 2 || occ == 3 || occ == 4 || occ == 8 || occ == 9 || occ == 10) ? occ : 0;   // 10: streamed class vector at any C (A/B, tests)
    g_tune_ppt = (ppt == 4 || ppt == 8) ? ppt : 0;
}
#endif

size_t pp_topk_workspace_bytes(int64_t B, int64_t N, int64_t k)
{
    if (B < 1 || N < 1 || k < 1 || N > 0x7FFFFFFFll || k > N || B > 0x7FFFFFFFll) return 0;
    if (k <= kSmallKMax) {
        Plan pl = make_plan(B, N, false);
        return merge_ws_bytes(B, (int64_t)pl.waves_per_image * k, k);
    }
    return large_ws_bytes(B, k);
}

size_t pp_acq_workspace_bytes(int64_t B, int64_t C, int64_t H, int64_t W, int64_t k)
{
    (void)C;
    if (B < 1 || H < 1 || W < 1 || k < 1 || H > 0x7FFFFFFFll || W > 0x7FFFFFFFll || H * W > 0x7FFFFFFFll || k > H * W || B > 0x7FFFFFFFll) return 0;
    const int64_t N = H * W;
    if (k <= kSmallKMax) {
        // sized for the 4-pixel tile (most waves), an upper bound for every plan
        Plan pl = make_plan(B, N, true, true);
        return merge_ws_bytes(B, (int64_t)pl.waves_per_image * k, k);
    }
    // score map (used when the caller passes no out_map) or candidate segments + large-k scratch + the list select's threshold keys / counts
    return map_region_bytes(B, N, k) + pp_topk_workspace_bytes(B, N, k) + emit_extra_bytes(B, N, k);
}

int pp_acq_score_map(const float* logits, int64_t B, int64_t C, int64_t H, int64_t W, int64_t sB, int64_t sC,
                     int64_t sH, int64_t sW, const uint8_t* exclude, int strategy, float* out_map,
                     pp_stream_t stream)
{
    const ExactScope exact_scope(strategy);
    if (int rc = validate(logits, B, C, H, W, strategy)) return rc;
    if (!out_map) return fail(PP_ERR_BAD_ARG, "out_map is null");
    const int64_t N = H * W;
    Plan pl = make_plan(B, N, is_flat_vec4(logits, exclude, out_map, H, W, sB, sC, sH, sW), exact_formula() != 0 || stream_classes(C));
    AcqParams p{logits, exclude, out_map, nullptr, sB, sC, sH, sW, (int)C, (int)W, N, pl.blocks_per_image, 0,
                strategy, g_reduce_mode, 0};
    return dispatch_acq(p, pl, B, as_stream(stream));
}

int pp_acq_softmax_sum(const float* logits, int64_t T, int64_t C, int64_t H, int64_t W, int64_t sT, int64_t sC, int64_t sH,
                       int64_t sW, float* prob_out, float* uc_out, int strategy, float scale, int accumulate, pp_stream_t stream)
{
    const ExactScope exact_scope(strategy);
    if (int rc = validate(logits, T, C, H, W, strategy)) return rc;
    if (!prob_out && !uc_out) return fail(PP_ERR_BAD_ARG, "softmax_sum: both outputs are null");
    const int64_t N = H * W;
    const int64_t blocks = (N + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(softmax_sum_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, as_stream(stream), logits, (int)T, (int)C, (int)W, N,
                       sT, sC, sH, sW, prob_out, uc_out, strategy, scale, accumulate);
    return check_launch("softmax_sum_kernel");
}

int pp_uncertainty_from_prob(const float* prob, int64_t B, int64_t C, int64_t H, int64_t W, int64_t sB, int64_t sC,
                             int64_t sH, int64_t sW, int strategy, float* out_map, pp_stream_t stream)
{
    const ExactScope exact_scope(strategy);
    if (int rc = validate(prob, B, C, H, W, strategy)) return rc;
    if (!out_map) return fail(PP_ERR_BAD_ARG, "out_map is null");
    const int64_t N = H * W;
    Plan pl = make_plan(B, N, is_flat_vec4(prob, nullptr, out_map, H, W, sB, sC, sH, sW), true);     // (4 pixels per thread: also what the streamed scorer needs)
    AcqParams p{prob, nullptr, out_map, nullptr, sB, sC, sH, sW, (int)C, (int)W, N, pl.blocks_per_image, 0,
                strategy, g_reduce_mode, 1};
    return dispatch_acq(p, pl, B, as_stream(stream));
}

int pp_acq_score_topk(const float* logits, int64_t B, int64_t C, int64_t H, int64_t W, int64_t sB, int64_t sC,
                      int64_t sH, int64_t sW, const uint8_t* exclude, int strategy, int64_t k, int32_t* out_idx,
                      float* out_val, float* out_map, void* workspace, size_t ws_bytes, pp_stream_t stream)
{
    const ExactScope exact_scope(strategy);
    if (int rc = validate(logits, B, C, H, W, strategy)) return rc;
    const int64_t N = H * W;
    if (k < 1 || k > N) return fail(PP_ERR_BAD_K, "k=%lld outside [1, H*W=%lld]", (long long)k, (long long)N);
    if (!out_idx) return fail(PP_ERR_BAD_ARG, "out_idx is null");
    const size_t need = pp_acq_workspace_bytes(B, C, H, W, k);
    if (!workspace || ws_bytes < need)
        return fail(PP_ERR_WORKSPACE, "workspace %zu B < required %zu B", ws_bytes, need);
    if (reinterpret_cast<uintptr_t>(workspace) & 255) return fail(PP_ERR_BAD_ARG, "workspace must be 256-B aligned");
    hipStream_t st = as_stream(stream);
    const int largest = strategy != PP_ACQ_MARGIN;

    if (k <= kSmallKMax) {
        Plan pl = make_plan(B, N, is_flat_vec4(logits, exclude, out_map, H, W, sB, sC, sH, sW), exact_formula() != 0 || stream_classes(C));
        const int64_t n_cand = (int64_t)pl.waves_per_image * k;
        uint64_t* cand = reinterpret_cast<uint64_t*>(workspace);
        uint64_t* other = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(workspace) +
                                                      align_up((size_t)B * n_cand * 8, 256));
        AcqParams p{logits, exclude, out_map, cand, sB, sC, sH, sW, (int)C, (int)W, N, pl.blocks_per_image,
                    (int)k, strategy, g_reduce_mode, 0};
        if (int rc = dispatch_acq(p, pl, B, st)) return rc;
        return run_merge(cand, n_cand, other, B, (int)k, largest, out_idx, out_val, st);
    }
    // large k: materialise the score map once, then radix-select + sort per image
    float* map = out_map ? out_map : reinterpret_cast<float*>(workspace);
    uint64_t* gbuf = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(workspace) + map_region_bytes(B, N, k));
    Plan pl = make_plan(B, N, is_flat_vec4(logits, exclude, map, H, W, sB, sC, sH, sW), exact_formula() != 0 || stream_classes(C));
    AcqParams p{logits, exclude, map, nullptr, sB, sC, sH, sW, (int)C, (int)W, N, pl.blocks_per_image, 0, strategy,
                g_reduce_mode, 0};
    const float qs = score_qscale(strategy, C);
    if (acq_emit_ok(p, pl, B, k, qs, out_map != nullptr))
        return run_emit_select(p, pl, B, k, largest, qs, workspace, reinterpret_cast<char*>(gbuf) + pp_topk_workspace_bytes(B, N, k),
                               out_idx, out_val, st);
    const bool fuse_hist = large_q_ok(B, k, qs) && acq_hist_fusable(p, pl);
    if (fuse_hist) {
        p.qhist = large_hist(gbuf, B, k);
        p.qscale = qs;
        if (hipMemsetAsync(p.qhist, 0, (size_t)B * kQBins * 4, st) != hipSuccess) return fail(PP_ERR_LAUNCH, "topk: memset failed");
    }
    if (int rc = dispatch_acq(p, pl, B, st)) return rc;
    return run_large(map, B, N, k, largest, gbuf, out_idx, out_val, st, qs, fuse_hist);
}

size_t pp_acq_lowres_workspace_bytes(int64_t B, int64_t C, int64_t Hc, int64_t Wc, int64_t k)
{
    (void)C;
    if (B < 1 || Hc < 1 || Wc < 1 || k < 1 || Hc > 0x7FFFFFFFll || Wc > 0x7FFFFFFFll || Hc * Wc > 0x7FFFFFFFll || k > Hc * Wc || B > 0x7FFFFFFFll) return 0;
    if (k <= kSmallKMax) {   // sized for the 4-row tile (most waves)
        const int64_t waves = cdiv(Wc, kWave) * cdiv(Hc, (kBlock / kWave) * 4) * (kBlock / kWave);
        return merge_ws_bytes(B, waves * k, k);
    }
    return align_up((size_t)B * Hc * Wc * 4, 256) + pp_topk_workspace_bytes(B, Hc * Wc, k);
}

int pp_acq_lowres_score_topk(const float* low, int64_t ldx, int64_t B, int64_t C, int64_t h, int64_t w, int64_t H,
                             int64_t W, int align_corners, int64_t Hc, int64_t Wc, const uint8_t* exclude, int strategy,
                             int64_t k, int32_t* out_idx, float* out_val, float* out_map, void* workspace,
                             size_t ws_bytes, pp_stream_t stream)
{
    const ExactScope exact_scope(strategy);
    if (int rc = validate_lowres(low, ldx, B, C, h, w, H, W, Hc, Wc, strategy)) return rc;
    const int64_t N = Hc * Wc;
    hipStream_t st = as_stream(stream);
    float sh, sw;
    lowres_scales(h, w, H, W, align_corners, sh, sw);
    LowresParams p{low, ldx, exclude, out_map, nullptr, (int)h, (int)w, (int)Hc, (int)Wc, sh, sw, align_corners ? 1 : 0,
                   (int)C, 0, 0, 0, strategy, g_reduce_mode, 0};
    if (k == 0) {     // score map only
        if (!out_map) return fail(PP_ERR_BAD_ARG, "k == 0 (map only) needs out_map");
        LowresPlan pl = make_lowres_plan(B, C, h, w, Hc, Wc, sh, sw, exact_formula() != 0 || stream_classes(C));
        p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.patch_cap = pl.patch_cap;
        return dispatch_lowres(p, pl, B, st);
    }
    if (k < 1 || k > N) return fail(PP_ERR_BAD_K, "k=%lld outside [1, H*W=%lld]", (long long)k, (long long)N);
    if (!out_idx) return fail(PP_ERR_BAD_ARG, "out_idx is null");
    const size_t need = pp_acq_lowres_workspace_bytes(B, C, Hc, Wc, k);
    if (!workspace || ws_bytes < need)
        return fail(PP_ERR_WORKSPACE, "workspace %zu B < required %zu B", ws_bytes, need);
    if (reinterpret_cast<uintptr_t>(workspace) & 255) return fail(PP_ERR_BAD_ARG, "workspace must be 256-B aligned");
    const int largest = strategy != PP_ACQ_MARGIN;
    LowresPlan pl = make_lowres_plan(B, C, h, w, Hc, Wc, sh, sw, exact_formula() != 0 || stream_classes(C));
    p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.patch_cap = pl.patch_cap;
    if (k <= kSmallKMax) {
        const int64_t n_cand = (int64_t)pl.waves_per_image * k;
        uint64_t* cand = reinterpret_cast<uint64_t*>(workspace);
        uint64_t* other = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(workspace) +
                                                      align_up((size_t)B * n_cand * 8, 256));
        p.cand = cand;
        p.k = (int)k;
        if (int rc = dispatch_lowres(p, pl, B, st)) return rc;
        return run_merge(cand, n_cand, other, B, (int)k, largest, out_idx, out_val, st);
    }
    float* map = out_map ? out_map : reinterpret_cast<float*>(workspace);
    uint64_t* gbuf = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(workspace) + align_up((size_t)B * N * 4, 256));
    p.out_map = map;
    const float qs = score_qscale(strategy, C);
    const bool fuse_hist = g_hist_fuse && large_q_ok(B, k, qs) && !stream_classes(C);      // (acq_lowres_kernel only: the streamed form has no histogram epilogue)
    if (fuse_hist) {
        p.qhist = large_hist(gbuf, B, k);
        p.qscale = qs;
        if (hipMemsetAsync(p.qhist, 0, (size_t)B * kQBins * 4, st) != hipSuccess) return fail(PP_ERR_LAUNCH, "topk: memset failed");
    }
    if (int rc = dispatch_lowres(p, pl, B, st)) return rc;
    return run_large(map, B, N, k, largest, gbuf, out_idx, out_val, st, qs, fuse_hist);
}

int pp_acq_lowres_score_at(const float* low, int64_t ldx, int64_t B, int64_t C, int64_t h, int64_t w, int64_t H,
                           int64_t W, int align_corners, int64_t Hc, int64_t Wc, int strategy, const int32_t* img_idx,
                           const int32_t* pix_idx, int64_t n, float* out, pp_stream_t stream)
{
    const ExactScope exact_scope(strategy);
    if (int rc = validate_lowres(low, ldx, B, C, h, w, H, W, Hc, Wc, strategy)) return rc;
    if (n == 0) return PP_OK;
    if (n < 0 || !img_idx || !pix_idx || !out) return fail(PP_ERR_BAD_ARG, "score_at: null pointer or n < 0");
    float sh, sw;
    lowres_scales(h, w, H, W, align_corners, sh, sw);
    hipStream_t st = as_stream(stream);
    dim3 grid((unsigned)cdiv(n, kBlock)), block(kBlock);
    const int al = align_corners ? 1 : 0;
#define PP_AT(CM, EX)                                                                                                   \
    hipLaunchKernelGGL((acq_lowres_at_kernel<CM, EX>), grid, block, 0, st, low, ldx, (int)h, (int)w, sh, sw, al, (int)Wc, \
                       (int)C, strategy, img_idx, pix_idx, n, out)
    if (stream_classes(C)) {
        hipLaunchKernelGGL(acq_lowres_at_stream_kernel, grid, block, 0, st, low, ldx, (int)h, (int)w, sh, sw, al, (int)Wc, (int)C, strategy,
                           img_idx, pix_idx, n, out);
        return check_launch("acq_lowres_at_stream_kernel");
    }
    switch (C) {
        case 11: PP_AT(11, true); break;
        case 19: PP_AT(19, true); break;
        case 21: PP_AT(21, true); break;
        default:
            if (C <= 32) PP_AT(32, false);
            else PP_AT(64, false);
    }
#undef PP_AT
    return check_launch("acq_lowres_at_kernel");
}

int pp_topk_select(const float* scores, int64_t B, int64_t N, int64_t k, int largest, int32_t* out_idx,
                   float* out_val, void* workspace, size_t ws_bytes, pp_stream_t stream)
{
    if (!scores || !out_idx) return fail(PP_ERR_BAD_ARG, "null pointer");
    if (B < 1 || N < 1 || N > 0x7FFFFFFFll) return fail(PP_ERR_BAD_ARG, "bad shape B=%lld N=%lld", (long long)B, (long long)N);
    if (k < 1 || k > N) return fail(PP_ERR_BAD_K, "k=%lld outside [1, N=%lld]", (long long)k, (long long)N);
    const size_t need = pp_topk_workspace_bytes(B, N, k);
    if (!workspace || ws_bytes < need)
        return fail(PP_ERR_WORKSPACE, "workspace %zu B < required %zu B", ws_bytes, need);
    if (reinterpret_cast<uintptr_t>(workspace) & 255) return fail(PP_ERR_BAD_ARG, "workspace must be 256-B aligned");
    hipStream_t st = as_stream(stream);
    if (k <= kSmallKMax) {
        Plan pl = make_plan(B, N, false);
        const int64_t n_cand = (int64_t)pl.waves_per_image * k;
        uint64_t* cand = reinterpret_cast<uint64_t*>(workspace);
        uint64_t* other = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(workspace) +
                                                      align_up((size_t)B * n_cand * 8, 256));
        dim3 grid((unsigned)(B * pl.blocks_per_image)), block(kBlock);
        if (pl.ppt == 8)
            hipLaunchKernelGGL((topk_small_from_scores_kernel<8>), grid, block, 0, st, scores, N,
                               pl.blocks_per_image, (int)k, largest, cand, g_reduce_mode);
        else
            hipLaunchKernelGGL((topk_small_from_scores_kernel<4>), grid, block, 0, st, scores, N,
                               pl.blocks_per_image, (int)k, largest, cand, g_reduce_mode);
        if (int rc = check_launch("topk_small_from_scores_kernel")) return rc;
        return run_merge(cand, n_cand, other, B, (int)k, largest, out_idx, out_val, st);
    }
    return run_large(scores, B, N, k, largest, reinterpret_cast<uint64_t*>(workspace), out_idx, out_val, st);
}

}  // extern "C"
