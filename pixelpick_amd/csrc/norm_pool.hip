// norm_pool.hip — GroupNorm(+ReLU) and MaxPool for the FPN-ResNet50 model, NHWC fp32.
//
//   nn.GroupNorm(32, C) + nn.ReLU        networks/decoders.py:92-94 (UpsampleBlock)
//   nn.MaxPool2d(3, stride 2, pad 1)     networks/backbones/resnet_models.py:121
//
// Bandwidth-bound; same conventions as nn_ops.hip (pixel stride ld, float4 along channels, deterministic
// two-stage reductions).  GroupNorm statistics are per (image, group) over H*W*(C/G) elements: per-channel
// column sums per image (partials per row block) are combined per group in fp64.
#include "pp_common.h"

namespace pp {

constexpr int kTN = 256;

struct GnGeom {
    int cq, cq_blk, rows_per_pass, nblk_cols, nblk_rows;
    int64_t rows_per_block;
};

static GnGeom gn_geom(int64_t P, int C, int B)
{
    GnGeom g;
    g.cq = C / 4;
    g.cq_blk = g.cq < kTN ? g.cq : kTN;
    g.rows_per_pass = kTN / g.cq_blk;
    g.nblk_cols = (int)cdiv(g.cq, g.cq_blk);
    int64_t want = 1024 / (g.nblk_cols * (int64_t)B);
    if (want < 1) want = 1;
    int64_t rpb = cdiv(cdiv(P, want), g.rows_per_pass) * g.rows_per_pass;
    if (rpb < g.rows_per_pass * 4) rpb = g.rows_per_pass * 4;
    g.rows_per_block = rpb;
    g.nblk_rows = (int)cdiv(P, rpb);
    return g;
}

// MODE 0: per-channel (sum x, sum x^2) per image.  MODE 1: (sum g, sum g*xhat), g = dy*[y>0], xhat from (mean,rstd).
// part layout: [B][nblk_rows][2][C]
template <int MODE>
__global__ __launch_bounds__(kTN) void gn_partial_kernel(const float* x, int64_t ldx, const float* dy, int64_t lddy,
                                                         const float* y, int64_t ldy, const float* mean,
                                                         const float* rstd, int64_t P, int C, int G, GnGeom g, float* part)
{
    __shared__ float4 sh[2][kTN];
    const int t = threadIdx.x;
    const int ql = t % g.cq_blk, ry = t / g.cq_blk;
    const int q = blockIdx.y * g.cq_blk + ql;
    const int b = blockIdx.z;
    const bool active = ry < g.rows_per_pass && q < g.cq;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (active) {
        const int cpg = C / G;
        float mu[4] = {0, 0, 0, 0}, rs[4] = {0, 0, 0, 0};
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int grp = (q * 4 + j) / cpg;
                mu[j] = mean[b * G + grp];
                rs[j] = rstd[b * G + grp];
            }
        }
        const int64_t r0 = (int64_t)blockIdx.x * g.rows_per_block;
        const int64_t r1 = r0 + g.rows_per_block < P ? r0 + g.rows_per_block : P;
        for (int64_t r = r0 + ry; r < r1; r += g.rows_per_pass) {
            const int64_t row = (int64_t)b * P + r;
            const float4 v = *reinterpret_cast<const float4*>(x + row * ldx + q * 4);
            if (MODE == 0) {
                s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w;
                s1.x = fmaf(v.x, v.x, s1.x); s1.y = fmaf(v.y, v.y, s1.y); s1.z = fmaf(v.z, v.z, s1.z); s1.w = fmaf(v.w, v.w, s1.w);
            } else {
                float4 gg = *reinterpret_cast<const float4*>(dy + row * lddy + q * 4);
                const float4 ya = *reinterpret_cast<const float4*>(y + row * ldy + q * 4);
                gg.x = ya.x > 0.f ? gg.x : 0.f; gg.y = ya.y > 0.f ? gg.y : 0.f;
                gg.z = ya.z > 0.f ? gg.z : 0.f; gg.w = ya.w > 0.f ? gg.w : 0.f;
                s0.x += gg.x; s0.y += gg.y; s0.z += gg.z; s0.w += gg.w;
                s1.x = fmaf(gg.x, (v.x - mu[0]) * rs[0], s1.x); s1.y = fmaf(gg.y, (v.y - mu[1]) * rs[1], s1.y);
                s1.z = fmaf(gg.z, (v.z - mu[2]) * rs[2], s1.z); s1.w = fmaf(gg.w, (v.w - mu[3]) * rs[3], s1.w);
            }
        }
    }
    sh[0][t] = s0;
    sh[1][t] = s1;
    __syncthreads();
    if (ry == 0 && q < g.cq) {
        for (int k = 1; k < g.rows_per_pass; ++k) {
            const float4 a = sh[0][k * g.cq_blk + ql], bb = sh[1][k * g.cq_blk + ql];
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += bb.x; s1.y += bb.y; s1.z += bb.z; s1.w += bb.w;
        }
        float* p0 = part + (((int64_t)b * g.nblk_rows + blockIdx.x) * 2 + 0) * C + q * 4;
        *reinterpret_cast<float4*>(p0) = s0;
        *reinterpret_cast<float4*>(p0 + C) = s1;
    }
}

// Row-block partials [B][nblk][2][C] -> [B][1][2][C]: 32 lanes per output add blocks l, l+32, .. in fp64, then a fixed
// LDS tree (deterministic).  The per-(image, group) / per-channel combines below then see nblk == 1; walking up to
// 256 row blocks serially in ONE thread per group made them the slowest kernels of the FPN step (305 / 112 us).
__global__ __launch_bounds__(kTN) void gn_reduce_blocks_kernel(const float* part, int nblk, int C2 /* 2*C */, int64_t nout,
                                                               float* red)
{
    __shared__ double sh[kTN];
    const int t = threadIdx.x, lane = t >> 3;
    const int64_t o = (int64_t)blockIdx.x * 8 + (t & 7);          // output index in [B][2C]
    double s = 0.0;
    if (o < nout) {
        const int64_t b = o / C2, j = o - b * C2;
        const float* p = part + b * nblk * C2 + j;
        int k = lane;
        for (; k + 96 < nblk; k += 128) {
            const float v0 = p[(int64_t)k * C2], v1 = p[(int64_t)(k + 32) * C2];
            const float v2 = p[(int64_t)(k + 64) * C2], v3 = p[(int64_t)(k + 96) * C2];
            s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
        }
        for (; k < nblk; k += 32) s += (double)p[(int64_t)k * C2];
    }
    sh[t] = s;
    __syncthreads();
    for (int off = 128; off >= 8; off >>= 1) {
        if (t < off) sh[t] += sh[t + off];
        __syncthreads();
    }
    if (t < 8 && o < nout) red[o] = (float)sh[t];
}

// one thread per (image, group): fixed-order fp64 combine over row blocks and the group's channels
__global__ __launch_bounds__(kTN) void gn_stats_kernel(const float* part, int B, int nblk, int C, int G, double count,
                                                       float eps, float* mean, float* rstd)
{
    const int i = blockIdx.x * kTN + threadIdx.x;
    if (i >= B * G) return;
    const int b = i / G, grp = i % G, cpg = C / G;
    double s = 0.0, ss = 0.0;
    for (int k = 0; k < nblk; ++k) {
        const float* p0 = part + (((int64_t)b * nblk + k) * 2) * C + grp * cpg;
        for (int c = 0; c < cpg; ++c) { s += (double)p0[c]; ss += (double)p0[C + c]; }
    }
    const double mu = s / count;
    double var = ss / count - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[i] = (float)mu;
    rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
}

// y = relu((x - mean_bg) * rstd_bg * gamma_c + beta_c)
__global__ __launch_bounds__(kTN) void gn_apply_kernel(const float* x, int64_t ldx, const float* mean, const float* rstd,
                                                       const float* gamma, const float* beta, int relu, float* y, int64_t ldy,
                                                       int B, int64_t P, int C, int G)
{
    const int cq = C / 4, cpg = C / G;
    const int64_t total = (int64_t)B * P * cq;
    for (int64_t e = (int64_t)blockIdx.x * kTN + threadIdx.x; e < total; e += (int64_t)gridDim.x * kTN) {
        const int64_t row = total <= 0xFFFFFFFFll ? (int64_t)((unsigned)e / (unsigned)cq) : e / cq;
        const int q = (int)(e - row * cq);
        const int b = (int)(row / P);
        const float4 v = *reinterpret_cast<const float4*>(x + row * ldx + q * 4);
        const float4 ga = *reinterpret_cast<const float4*>(gamma + q * 4);
        const float4 be = *reinterpret_cast<const float4*>(beta + q * 4);
        const float in[4] = {v.x, v.y, v.z, v.w}, gm[4] = {ga.x, ga.y, ga.z, ga.w}, bt[4] = {be.x, be.y, be.z, be.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int grp = (q * 4 + j) / cpg;
            const float z = (in[j] - mean[b * G + grp]) * rstd[b * G + grp] * gm[j] + bt[j];
            o[j] = relu ? fmaxf(z, 0.0f) : z;
        }
        *reinterpret_cast<float4*>(y + row * ldy + q * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// per channel: dgamma_c = sum_b B_bc, dbeta_c = sum_b A_bc ; per (b,g): S1 = sum_{c in g} gamma_c A_bc, S2 = sum gamma_c B_bc
__global__ __launch_bounds__(kTN) void gn_bwd_stats_kernel(const float* part, int B, int nblk, int C, int G, const float* gamma,
                                                           float* dgamma, float* dbeta, float* s1, float* s2)
{
    const int i = blockIdx.x * kTN + threadIdx.x;
    const int cpg = C / G;
    if (i < C) {
        double a = 0.0, bsum = 0.0;
        for (int b = 0; b < B; ++b)
            for (int k = 0; k < nblk; ++k) {
                const float* p0 = part + (((int64_t)b * nblk + k) * 2) * C + i;
                a += (double)p0[0];
                bsum += (double)p0[C];
            }
        dbeta[i] = (float)a;
        dgamma[i] = (float)bsum;
    }
    if (i < B * G) {
        const int b = i / G, grp = i % G;
        double a = 0.0, bb = 0.0;
        for (int c = grp * cpg; c < (grp + 1) * cpg; ++c) {
            double ac = 0.0, bc = 0.0;
            for (int k = 0; k < nblk; ++k) {
                const float* p0 = part + (((int64_t)b * nblk + k) * 2) * C + c;
                ac += (double)p0[0];
                bc += (double)p0[C];
            }
            a += (double)gamma[c] * ac;
            bb += (double)gamma[c] * bc;
        }
        s1[i] = (float)a;
        s2[i] = (float)bb;
    }
}

// dx = rstd * (gamma*g - S1/n - xhat*S2/n)
__global__ __launch_bounds__(kTN) void gn_bwd_apply_kernel(const float* x, int64_t ldx, const float* dy, int64_t lddy,
                                                           const float* y, int64_t ldy, const float* mean, const float* rstd,
                                                           const float* gamma, const float* s1, const float* s2, float inv_n,
                                                           float* dx, int64_t lddx, int B, int64_t P, int C, int G)
{
    const int cq = C / 4, cpg = C / G;
    const int64_t total = (int64_t)B * P * cq;
    for (int64_t e = (int64_t)blockIdx.x * kTN + threadIdx.x; e < total; e += (int64_t)gridDim.x * kTN) {
        const int64_t row = total <= 0xFFFFFFFFll ? (int64_t)((unsigned)e / (unsigned)cq) : e / cq;
        const int q = (int)(e - row * cq);
        const int b = (int)(row / P);
        const float4 v = *reinterpret_cast<const float4*>(x + row * ldx + q * 4);
        const float4 gg = *reinterpret_cast<const float4*>(dy + row * lddy + q * 4);
        const float4 ya = *reinterpret_cast<const float4*>(y + row * ldy + q * 4);
        const float4 ga = *reinterpret_cast<const float4*>(gamma + q * 4);
        const float in[4] = {v.x, v.y, v.z, v.w}, gr[4] = {gg.x, gg.y, gg.z, gg.w}, yy[4] = {ya.x, ya.y, ya.z, ya.w};
        const float gm[4] = {ga.x, ga.y, ga.z, ga.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int grp = (q * 4 + j) / cpg;
            const float mu = mean[b * G + grp], rs = rstd[b * G + grp];
            const float g = yy[j] > 0.0f ? gr[j] : 0.0f;
            const float xh = (in[j] - mu) * rs;
            o[j] = rs * (gm[j] * g - s1[b * G + grp] * inv_n - xh * s2[b * G + grp] * inv_n);
        }
        *reinterpret_cast<float4*>(dx + row * lddx + q * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// ---- max pool 3x3 / stride s / pad p, first-maximum rule of torch (strictly greater wins, NaN propagates) ----------
__global__ __launch_bounds__(kTN) void maxpool_fwd_kernel(const float* x, int64_t ldx, int B, int H, int W, int cq, int ks,
                                                          int stride, int pad, float* y, int64_t ldy, unsigned char* idx,
                                                          int Ho, int Wo)
{
    const int64_t total = (int64_t)B * Ho * Wo * cq;
    for (int64_t e = (int64_t)blockIdx.x * kTN + threadIdx.x; e < total; e += (int64_t)gridDim.x * kTN) {
        const int q = (int)(e % cq);
        int64_t t = e / cq;
        const int ow = (int)(t % Wo); t /= Wo;
        const int oh = (int)(t % Ho);
        const int b = (int)(t / Ho);
        float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        unsigned char am[4] = {255, 255, 255, 255};
        for (int th = 0; th < ks; ++th) {
            const int ih = oh * stride - pad + th;
            if ((unsigned)ih >= (unsigned)H) continue;
            for (int tw = 0; tw < ks; ++tw) {
                const int iw = ow * stride - pad + tw;
                if ((unsigned)iw >= (unsigned)W) continue;
                const float4 v = *reinterpret_cast<const float4*>(x + (((int64_t)b * H + ih) * W + iw) * ldx + q * 4);
                const float in[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)   // torch: first in-bounds element seeds the index; then (val > max || isnan(val))
                    if (am[j] == 255 || in[j] > m[j] || in[j] != in[j]) { m[j] = in[j]; am[j] = (unsigned char)(th * ks + tw); }
            }
        }
        const int64_t o = (((int64_t)b * Ho + oh) * Wo + ow);
        *reinterpret_cast<float4*>(y + o * ldy + q * 4) = make_float4(m[0], m[1], m[2], m[3]);
        *reinterpret_cast<uchar4*>(idx + o * (cq * 4) + q * 4) = make_uchar4(am[0], am[1], am[2], am[3]);
    }
}

// gather: input pixel collects dy of every window whose recorded argmax is this pixel
__global__ __launch_bounds__(kTN) void maxpool_bwd_kernel(const float* dy, int64_t lddy, const unsigned char* idx, int B, int Ho,
                                                          int Wo, int cq, int ks, int stride, int pad, float* dx, int64_t lddx,
                                                          int H, int W)
{
    const int64_t total = (int64_t)B * H * W * cq;
    for (int64_t e = (int64_t)blockIdx.x * kTN + threadIdx.x; e < total; e += (int64_t)gridDim.x * kTN) {
        const int q = (int)(e % cq);
        int64_t t = e / cq;
        const int iw = (int)(t % W); t /= W;
        const int ih = (int)(t % H);
        const int b = (int)(t / H);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int th = 0; th < ks; ++th) {
            const int nh = ih + pad - th;
            if (nh < 0 || nh % stride != 0) continue;
            const int oh = nh / stride;
            if (oh >= Ho) continue;
            for (int tw = 0; tw < ks; ++tw) {
                const int nw = iw + pad - tw;
                if (nw < 0 || nw % stride != 0) continue;
                const int ow = nw / stride;
                if (ow >= Wo) continue;
                const int64_t o = (((int64_t)b * Ho + oh) * Wo + ow);
                const uchar4 a = *reinterpret_cast<const uchar4*>(idx + o * (cq * 4) + q * 4);
                const float4 g = *reinterpret_cast<const float4*>(dy + o * lddy + q * 4);
                const unsigned char tap = (unsigned char)(th * ks + tw);
                if (a.x == tap) acc[0] += g.x;
                if (a.y == tap) acc[1] += g.y;
                if (a.z == tap) acc[2] += g.z;
                if (a.w == tap) acc[3] += g.w;
            }
        }
        *reinterpret_cast<float4*>(dx + (((int64_t)b * H + ih) * W + iw) * lddx + q * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

static inline unsigned gridn(int64_t total)
{
    int64_t b = cdiv(total, kTN);
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace pp

using namespace pp;

extern "C" {

size_t pp_groupnorm_workspace_bytes(int B, int64_t P, int C)
{
    if (B < 1 || P < 1 || C < 4) return 256;
    GnGeom g = gn_geom(P, C, B);
    return align_up((size_t)B * g.nblk_rows * 2 * C * 4 + (size_t)2 * B * C * 4 + (size_t)2 * B * C * 4, 256);
}

int pp_groupnorm_relu_fwd(const float* x, int64_t ldx, int B, int64_t P, int C, int G, const float* gamma, const float* beta,
                          float eps, int relu, float* y, int64_t ldy, float* mean, float* rstd, void* workspace,
                          size_t ws_bytes, pp_stream_t stream)
{
    if (!x || !gamma || !beta || !y || !mean || !rstd) return fail(PP_ERR_BAD_ARG, "groupnorm fwd: null");
    if (C % 4 != 0 || G < 1 || C % G != 0) return fail(PP_ERR_UNSUPPORTED, "groupnorm fwd: C=%d G=%d", C, G);
    if (!workspace || ws_bytes < pp_groupnorm_workspace_bytes(B, P, C)) return fail(PP_ERR_WORKSPACE, "groupnorm fwd: workspace");
    hipStream_t st = as_stream(stream);
    GnGeom g = gn_geom(P, C, B);
    float* part = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL((gn_partial_kernel<0>), dim3(g.nblk_rows, g.nblk_cols, B), dim3(kTN), 0, st, x, ldx, (const float*)nullptr,
                       (int64_t)0, (const float*)nullptr, (int64_t)0, (const float*)nullptr, (const float*)nullptr, P, C, G, g, part);
    if (int rc = check_launch("gn_partial_kernel<0>")) return rc;
    float* red = part + (size_t)B * g.nblk_rows * 2 * C + (size_t)2 * B * C;
    const int64_t nout = (int64_t)B * 2 * C;
    hipLaunchKernelGGL(gn_reduce_blocks_kernel, dim3((unsigned)cdiv(nout, 8)), dim3(kTN), 0, st, part, g.nblk_rows, 2 * C, nout, red);
    if (int rc = check_launch("gn_reduce_blocks_kernel")) return rc;
    hipLaunchKernelGGL(gn_stats_kernel, dim3((unsigned)cdiv(B * G, kTN)), dim3(kTN), 0, st, red, B, 1, C, G,
                       (double)P * (C / G), eps, mean, rstd);
    if (int rc = check_launch("gn_stats_kernel")) return rc;
    hipLaunchKernelGGL(gn_apply_kernel, dim3(gridn((int64_t)B * P * (C / 4))), dim3(kTN), 0, st, x, ldx, mean, rstd, gamma, beta,
                       relu, y, ldy, B, P, C, G);
    return check_launch("gn_apply_kernel");
}

int pp_groupnorm_relu_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* y, int64_t ldy, int B,
                          int64_t P, int C, int G, const float* mean, const float* rstd, const float* gamma, float* dgamma,
                          float* dbeta, float* dx, int64_t lddx, void* workspace, size_t ws_bytes, pp_stream_t stream)
{
    if (!x || !dy || !y || !mean || !rstd || !gamma || !dgamma || !dbeta || !dx) return fail(PP_ERR_BAD_ARG, "groupnorm bwd: null");
    if (C % 4 != 0 || G < 1 || C % G != 0) return fail(PP_ERR_UNSUPPORTED, "groupnorm bwd: C=%d G=%d", C, G);
    if (!workspace || ws_bytes < pp_groupnorm_workspace_bytes(B, P, C)) return fail(PP_ERR_WORKSPACE, "groupnorm bwd: workspace");
    hipStream_t st = as_stream(stream);
    GnGeom g = gn_geom(P, C, B);
    float* part = reinterpret_cast<float*>(workspace);
    float* s1 = part + (size_t)B * g.nblk_rows * 2 * C;
    float* s2 = s1 + (size_t)B * C;
    hipLaunchKernelGGL((gn_partial_kernel<1>), dim3(g.nblk_rows, g.nblk_cols, B), dim3(kTN), 0, st, x, ldx, dy, lddy, y, ldy, mean,
                       rstd, P, C, G, g, part);
    if (int rc = check_launch("gn_partial_kernel<1>")) return rc;
    float* red = s2 + (size_t)B * C;
    const int64_t nout = (int64_t)B * 2 * C;
    hipLaunchKernelGGL(gn_reduce_blocks_kernel, dim3((unsigned)cdiv(nout, 8)), dim3(kTN), 0, st, part, g.nblk_rows, 2 * C, nout, red);
    if (int rc = check_launch("gn_reduce_blocks_kernel")) return rc;
    const int n = C > B * G ? C : B * G;
    hipLaunchKernelGGL(gn_bwd_stats_kernel, dim3((unsigned)cdiv(n, kTN)), dim3(kTN), 0, st, red, B, 1, C, G, gamma,
                       dgamma, dbeta, s1, s2);
    if (int rc = check_launch("gn_bwd_stats_kernel")) return rc;
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(gridn((int64_t)B * P * (C / 4))), dim3(kTN), 0, st, x, ldx, dy, lddy, y, ldy, mean,
                       rstd, gamma, s1, s2, 1.0f / ((float)P * (float)(C / G)), dx, lddx, B, P, C, G);
    return check_launch("gn_bwd_apply_kernel");
}

int pp_maxpool2d_fwd(const float* x, int64_t ldx, int B, int H, int W, int C, int ksize, int stride, int pad, float* y,
                     int64_t ldy, unsigned char* argmax, pp_stream_t stream)
{
    if (!x || !y || !argmax) return fail(PP_ERR_BAD_ARG, "maxpool fwd: null");
    if (C % 4 != 0 || ksize < 1 || ksize > 15) return fail(PP_ERR_UNSUPPORTED, "maxpool fwd: C=%d k=%d", C, ksize);
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(gridn((int64_t)B * Ho * Wo * (C / 4))), dim3(kTN), 0, as_stream(stream), x, ldx, B, H,
                       W, C / 4, ksize, stride, pad, y, ldy, argmax, Ho, Wo);
    return check_launch("maxpool_fwd_kernel");
}

int pp_maxpool2d_bwd(const float* dy, int64_t lddy, const unsigned char* argmax, int B, int H, int W, int C, int ksize, int stride,
                     int pad, float* dx, int64_t lddx, pp_stream_t stream)
{
    if (!dy || !dx || !argmax) return fail(PP_ERR_BAD_ARG, "maxpool bwd: null");
    if (C % 4 != 0) return fail(PP_ERR_UNSUPPORTED, "maxpool bwd: C=%d", C);
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(gridn((int64_t)B * H * W * (C / 4))), dim3(kTN), 0, as_stream(stream), dy, lddy, argmax,
                       B, Ho, Wo, C / 4, ksize, stride, pad, dx, lddx, H, W);
    return check_launch("maxpool_bwd_kernel");
}

}  // extern "C"
