/*
 * oracle/acq_oracle.c — CPU restatement of PixelPick's acquisition arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under pixelpick_amd/ may import, link or call this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * Parity status: PINNED — checked in tests/test_oracle_golden.py against golden vectors generated
 * by importing the reference (tools/gen_golden_acq.py -> tests/golden/acq_*.npz).
 *
 * Each function cites the reference lines (in /root/reference) it restates.  Plain scalar C,
 * float32 arithmetic in the reference's operation order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { ORC_ENTROPY = 0, ORC_LEAST_CONFIDENCE = 1, ORC_MARGIN = 2 };

/* query.py:190 F.softmax(dim=1) then query.py:229-239 UncertaintySampler.{_entropy,_least_confidence,
 * _margin_sampling}; one pixel, C classes at stride sC. */
static float score_pixel(const float* x, int64_t C, int64_t sC, int strategy, float* p /*[C] scratch*/)
{
    float m = x[0];
    for (int64_t c = 1; c < C; ++c) { float v = x[c * sC]; if (v > m) m = v; }
    float S = 0.0f;
    for (int64_t c = 0; c < C; ++c) { p[c] = expf(x[c * sC] - m); S += p[c]; }
    for (int64_t c = 0; c < C; ++c) p[c] = p[c] / S;                      /* softmax */
    if (strategy == ORC_ENTROPY) {                                        /* query.py:230 */
        float acc = 0.0f;
        for (int64_t c = 0; c < C; ++c) acc += (-p[c]) * logf(p[c]);      /* 0*log 0 = NaN, as the reference */
        return acc;
    } else if (strategy == ORC_LEAST_CONFIDENCE) {                        /* query.py:234 */
        float mx = p[0];
        for (int64_t c = 1; c < C; ++c) if (p[c] > mx) mx = p[c];
        return 1.0f - mx;
    } else {                                                              /* query.py:238-239 */
        float t1 = -INFINITY, t2 = -INFINITY;
        for (int64_t c = 0; c < C; ++c) {
            float v = p[c];
            if (v > t1) { t2 = t1; t1 = v; } else if (v > t2) { t2 = v; }
        }
        return fabsf(t1 - t2);
    }
}

/* Score map for B images; logits addressed by element strides (NCHW or NHWC).  out_map [B,H,W]. */
int orc_acq_score_map(const float* logits, int64_t B, int64_t C, int64_t H, int64_t W,
                      int64_t sB, int64_t sC, int64_t sH, int64_t sW, int strategy, float* out_map)
{
    if (C < 1 || C > 4096 || strategy < 0 || strategy > 2) return -1;
    float* p = (float*)malloc(sizeof(float) * (size_t)C);
    for (int64_t b = 0; b < B; ++b)
        for (int64_t h = 0; h < H; ++h)
            for (int64_t w = 0; w < W; ++w)
                out_map[(b * H + h) * W + w] =
                    score_pixel(logits + b * sB + h * sH + w * sW, C, sC, strategy, p);
    free(p);
    return 0;
}

/* query.py:195-201: uc_map[mask] = fill ; uc_map[mask_void] = fill  (fill 0.0 entropy/LC, 1.0 margin) */
int orc_acq_apply_exclude(float* map, const uint8_t* exclude, int64_t n, int strategy)
{
    const float fill = (strategy == ORC_MARGIN) ? 1.0f : 0.0f;
    for (int64_t i = 0; i < n; ++i) if (exclude[i]) map[i] = fill;
    return 0;
}

/* Orderable key: larger key == selected earlier.  Policy: ties -> lower flat index first;
 * NaN sorts as +inf for largest=1 (torch.topk picks NaN first) and last for largest=0; -0.0 == +0.0. */
static uint32_t order_key(float v, int largest)
{
    uint32_t u;
    if (v != v) u = 0xFFFFFFFFu;
    else {
        v = v + 0.0f;                      /* -0.0 -> +0.0 */
        memcpy(&u, &v, 4);
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        if (u == 0xFFFFFFFFu) u = 0xFFFFFFFEu; /* unreachable for non-NaN; keep NaN strictly on top */
    }
    return largest ? u : ~u;
}

typedef struct { uint64_t k; } ck_t;
static int cmp_desc(const void* a, const void* b)
{
    uint64_t x = ((const ck_t*)a)->k, y = ((const ck_t*)b)->k;
    return (x < y) - (x > y);
}

/* query.py:57-61: uc_map.flatten().topk(k, largest).indices — value-sorted, fixed tiebreak.
 * scores [n]; out_idx [k] (int32 flat index), out_val [k] (may be NULL). */
int orc_topk(const float* scores, int64_t n, int64_t k, int largest, int32_t* out_idx, float* out_val)
{
    if (k < 0 || k > n || n > 0x7FFFFFFF) return -1;
    ck_t* a = (ck_t*)malloc(sizeof(ck_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i)
        a[i].k = ((uint64_t)order_key(scores[i], largest) << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)i);
    qsort(a, (size_t)n, sizeof(ck_t), cmp_desc);
    for (int64_t j = 0; j < k; ++j) {
        uint32_t idx = 0xFFFFFFFFu - (uint32_t)(a[j].k & 0xFFFFFFFFu);
        out_idx[j] = (int32_t)idx;
        if (out_val) out_val[j] = scores[idx];
    }
    free(a);
    return 0;
}

/* The fused operation the C ABI's pp_acq_score_topk performs (query.py:190-204 for B images). */
int orc_acq_score_topk(const float* logits, int64_t B, int64_t C, int64_t H, int64_t W,
                       int64_t sB, int64_t sC, int64_t sH, int64_t sW, const uint8_t* exclude,
                       int strategy, int64_t k, int32_t* out_idx, float* out_val, float* out_map)
{
    const int64_t n = H * W;
    float* map = out_map ? out_map : (float*)malloc(sizeof(float) * (size_t)(B * n));
    int rc = orc_acq_score_map(logits, B, C, H, W, sB, sC, sH, sW, strategy, map);
    if (rc == 0 && exclude) orc_acq_apply_exclude(map, exclude, B * n, strategy);
    for (int64_t b = 0; rc == 0 && b < B; ++b)
        rc = orc_topk(map + b * n, n, k, strategy != ORC_MARGIN, out_idx + b * k, out_val ? out_val + b * k : NULL);
    if (!out_map) free(map);
    return rc;
}
