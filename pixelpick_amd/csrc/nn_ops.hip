// nn_ops.hip — bandwidth-bound layers of the PixelPick networks for gfx950, NHWC, fp32.
//
// Everything here is HBM-bound: 16 B/lane coalesced accesses along the channel axis, one pass per
// tensor, deterministic two-stage reductions (no float atomics), no MFMA.
//
//   BatchNorm2d (train + eval)   networks/mobilenet_v2.py:10,39-57; aspp.py:11,56,59; deeplab.py:25;
//                                decoders.py:108,112                                 
//   ReLU / ReLU6 / residual add  mobilenet_v2.py:11,40,50,54,62-63                   (N5)
//   depthwise 3x3 conv           mobilenet_v2.py:38,52 (stride 1/2, dilation, "valid" on the padded map) (N4)
//   fixed_padding                mobilenet_v2.py:15-21                               (N2)
//   bilinear interpolate         deeplab.py:49,55; aspp.py:70; decoders.py:82,101    (N11, N14)
//   AdaptiveAvgPool2d(1)         aspp.py:54                                          (N8)
//   Dropout                      aspp.py:61; decoders.py:110,114                     (own counter-based RNG)
//   cross_entropy(ignore_index)  model.py:116                                        (L2)
//   Adam step                    utils/utils.py:125-141 (torch.optim.Adam semantics) (L4)
#include "pp_common.h"
#include "bn_xchg.h"

namespace pp {

constexpr int kT = 256;

__device__ __forceinline__ float act_fwd(float z, int act)
{
    if (act == 1) return fmaxf(z, 0.0f);
    if (act == 2) return fminf(fmaxf(z, 0.0f), 6.0f);
    return z;
}
// flat element index -> (q, w, h, b) for an array [B][H][W][cq]; 32-bit unsigned divisions whenever the index fits
// (64-bit integer division is emulated in ~100 instructions on this ISA and was most of the depthwise kernels' work)
__device__ __forceinline__ void decode_bhwq(int64_t e, int cq, int Wd, int Hd, int& q, int& w, int& h, int& b)
{
    if (e <= 0xFFFFFFFFll) {
        const unsigned u = (unsigned)e;
        const unsigned t = u / (unsigned)cq;
        q = (int)(u - t * (unsigned)cq);
        const unsigned t2 = t / (unsigned)Wd;
        w = (int)(t - t2 * (unsigned)Wd);
        const unsigned t3 = t2 / (unsigned)Hd;
        h = (int)(t2 - t3 * (unsigned)Hd);
        b = (int)t3;
    } else {
        q = (int)(e % cq);
        int64_t t = e / cq;
        w = (int)(t % Wd); t /= Wd;
        h = (int)(t % Hd);
        b = (int)(t / Hd);
    }
}

// ================================================================================================
// Column reductions over an [M, C] matrix (pixel stride ld): partial sums per row-block, then a finalize.
// Thread mapping: float4 column q = t % cq_blk, row lane ry = t / cq_blk (consecutive threads walk the
// channel axis: coalesced for every C).  MODE 0: (sum x, sum x^2)   MODE 1: (sum g, sum g*xhat) for BN backward.
// ================================================================================================
struct ColReduceGeom {
    int cq;            // C/4
    int cq_blk;        // float4 columns per block (<= 256)
    int rows_per_pass; // 256 / cq_blk
    int64_t rows_per_block;
    int nblk_rows;     // grid.x
    int nblk_cols;     // grid.y
};

static int g_dw_wgrad_blocks = 1024;    // row blocks aimed at by the depthwise weight gradient (pp_debug_set_dw_variant bits 1..)

// cq_blk_max < 256: NARROW column blocks (wide maps get several block columns and 256 / cq_blk row lanes each: the 1/16-resolution
// depthwise weight gradients - 2048 pixels x 384..960 channels - otherwise run one row lane per block and walk their rows in sequence);
// min_passes: rows a thread walks at least (fewer, fatter blocks keep the partial traffic down)
static ColReduceGeom col_geom(int64_t M, int C, int target_blocks = 1024, int cq_blk_max = kT, int min_passes = 4)
{
    ColReduceGeom g;
    g.cq = C / 4;
    g.cq_blk = g.cq < cq_blk_max ? g.cq : cq_blk_max;
    g.rows_per_pass = kT / g.cq_blk;
    g.nblk_cols = (int)cdiv(g.cq, g.cq_blk);
    // measured (profiles/r01_train_step_*): these reductions are latency-bound, more row blocks win even
    // for the small 1/16-resolution maps; the second-stage combine reads the partials with 4 loads in flight
    int64_t want_blocks = target_blocks / g.nblk_cols;
    if (want_blocks < 1) want_blocks = 1;
    int64_t rpb = cdiv(cdiv(M, want_blocks), g.rows_per_pass) * g.rows_per_pass;
    if (rpb < g.rows_per_pass * min_passes) rpb = g.rows_per_pass * min_passes;
    g.rows_per_block = rpb;
    g.nblk_rows = (int)cdiv(M, rpb);
    return g;
}

template <int MODE>
__global__ __launch_bounds__(kT) void col_reduce_kernel(const float* x, const float* dy, const float* yact, int act,
                                                       const float* mean, const float* invstd, int64_t M, int C,
                                                       int64_t ldx, int64_t lddy, int64_t ldya, ColReduceGeom g,
                                                       float* part /*[nblk_rows][2][C]*/)
{
    __shared__ float4 sh[2][kT];
    const int t = threadIdx.x;
    const int ql = t % g.cq_blk, ry = t / g.cq_blk;
    const int q = blockIdx.y * g.cq_blk + ql;
    const bool active = ry < g.rows_per_pass && q < g.cq;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (active) {
        float4 mu = s0, is = s0;
        if (MODE == 1) {
            mu = *reinterpret_cast<const float4*>(mean + q * 4);
            is = *reinterpret_cast<const float4*>(invstd + q * 4);
        }
        const int64_t r0 = (int64_t)blockIdx.x * g.rows_per_block;
        const int64_t r1 = r0 + g.rows_per_block < M ? r0 + g.rows_per_block : M;
        for (int64_t r = r0 + ry; r < r1; r += g.rows_per_pass) {
            const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + q * 4);
            if (MODE == 0) {
                s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w;
                s1.x = fmaf(v.x, v.x, s1.x); s1.y = fmaf(v.y, v.y, s1.y);
                s1.z = fmaf(v.z, v.z, s1.z); s1.w = fmaf(v.w, v.w, s1.w);
            } else {
                float4 gg = *reinterpret_cast<const float4*>(dy + r * lddy + q * 4);
                if (act != 0) {
                    const float4 ya = *reinterpret_cast<const float4*>(yact + r * ldya + q * 4);
                    gg.x *= act_mask(ya.x, act); gg.y *= act_mask(ya.y, act);
                    gg.z *= act_mask(ya.z, act); gg.w *= act_mask(ya.w, act);
                }
                s0.x += gg.x; s0.y += gg.y; s0.z += gg.z; s0.w += gg.w;
                s1.x = fmaf(gg.x, (v.x - mu.x) * is.x, s1.x); s1.y = fmaf(gg.y, (v.y - mu.y) * is.y, s1.y);
                s1.z = fmaf(gg.z, (v.z - mu.z) * is.z, s1.z); s1.w = fmaf(gg.w, (v.w - mu.w) * is.w, s1.w);
            }
        }
    }
    sh[0][t] = s0;
    sh[1][t] = s1;
    __syncthreads();
    if (ry == 0 && q < g.cq) {
        for (int k = 1; k < g.rows_per_pass; ++k) {
            const float4 a = sh[0][k * g.cq_blk + ql], b = sh[1][k * g.cq_blk + ql];
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
        }
        float* p0 = part + ((int64_t)blockIdx.x * 2 + 0) * C + q * 4;
        float* p1 = part + ((int64_t)blockIdx.x * 2 + 1) * C + q * 4;
        *reinterpret_cast<float4*>(p0) = s0;
        *reinterpret_cast<float4*>(p1) = s1;
    }
}

// BN forward finalize: batch mean / biased var (fp64 combine), running-stat update (momentum, unbiased var),
// scale = gamma*invstd, shift = beta - mean*scale.   nn.BatchNorm2d training semantics.
__global__ __launch_bounds__(kT) void bn_finalize_kernel(const float* part, int nblk, int C, double count,
                                                        const float* gamma, const float* beta, float eps,
                                                        float momentum, float* running_mean, float* running_var,
                                                        float* mean, float* invstd, float* scale, float* shift)
{
    __shared__ double shd[kT];
    const int c = blockIdx.x * 8 + (threadIdx.x & 7);
    const double s = lanes32_sum(part, nblk, (int64_t)2 * C, c, c < C, shd);
    const double ss = lanes32_sum(part + C, nblk, (int64_t)2 * C, c, c < C, shd);
    if (c >= C || threadIdx.x >= 8) return;
    const double mu = s / count;
    double var = ss / count - mu * mu;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)mu;
    invstd[c] = is;
    const float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mu * sc;
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mu;
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// The same finalize for partial rows written by a convolution's epilogue (hundreds to thousands of rows): one block per channel
// QUAD, 256 row lanes with 16-byte loads, fp64 lane sums combined by a fixed-order LDS tree.
__global__ __launch_bounds__(kT) void bn_finalize_quad_kernel(const float* part, int nblk, int C, double count, const float* gamma,
                                                             const float* beta, float eps, float momentum, float* running_mean,
                                                             float* running_var, float* mean, float* invstd, float* scale, float* shift)
{
    __shared__ double sh[2][4][kT];
    const int q = blockIdx.x, t = threadIdx.x;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, ss[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = t; b < nblk; b += kT) {
        const float4 a = *reinterpret_cast<const float4*>(part + ((int64_t)b * 2 + 0) * C + q * 4);
        const float4 c = *reinterpret_cast<const float4*>(part + ((int64_t)b * 2 + 1) * C + q * 4);
        s[0] += (double)a.x; s[1] += (double)a.y; s[2] += (double)a.z; s[3] += (double)a.w;
        ss[0] += (double)c.x; ss[1] += (double)c.y; ss[2] += (double)c.z; ss[3] += (double)c.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { sh[0][j][t] = s[j]; sh[1][j][t] = ss[j]; }
    __syncthreads();
    for (int off = kT / 2; off >= 1; off >>= 1) {
        if (t < off) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { sh[0][j][t] += sh[0][j][t + off]; sh[1][j][t] += sh[1][j][t + off]; }
        }
        __syncthreads();
    }
    if (t >= 4) return;
    const int c = q * 4 + t;
    const double mu = sh[0][t][0] / count;
    double var = sh[1][t][0] / count - mu * mu;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)mu;
    invstd[c] = is;
    const float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mu * sc;
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mu;
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// eval-mode BN: scale/shift from running statistics
__global__ __launch_bounds__(kT) void bn_eval_affine_kernel(int C, const float* gamma, const float* beta,
                                                           const float* running_mean, const float* running_var,
                                                           float eps, float* scale, float* shift)
{
    const int c = blockIdx.x * kT + threadIdx.x;
    if (c >= C) return;
    const float is = 1.0f / sqrtf(running_var[c] + eps);
    const float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] - running_mean[c] * sc;
}

// y = act(x*scale + shift [+ res]) ; float4 along channels
__global__ __launch_bounds__(kT) void bn_apply_kernel(const float* x, int64_t ldx, const float* scale,
                                                     const float* shift, const float* res, int64_t ldr, int act,
                                                     float* y, int64_t ldy, int64_t M, int cq)
{
    const int64_t total = M * cq;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
        const int64_t r = total <= 0xFFFFFFFFll ? (int64_t)((unsigned)e / (unsigned)cq) : e / cq;
        const int q = (int)(e - r * cq);
        const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + q * 4);
        const float4 sc = *reinterpret_cast<const float4*>(scale + q * 4);
        const float4 sf = *reinterpret_cast<const float4*>(shift + q * 4);
        float4 o;
        o.x = fmaf(v.x, sc.x, sf.x); o.y = fmaf(v.y, sc.y, sf.y); o.z = fmaf(v.z, sc.z, sf.z); o.w = fmaf(v.w, sc.w, sf.w);
        if (res) {
            const float4 rr = *reinterpret_cast<const float4*>(res + r * ldr + q * 4);
            o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
        }
        o.x = act_fwd(o.x, act); o.y = act_fwd(o.y, act); o.z = act_fwd(o.z, act); o.w = act_fwd(o.w, act);
        *reinterpret_cast<float4*>(y + r * ldy + q * 4) = o;
    }
}

// BN backward finalize: dbeta = sum g, dgamma = sum g*xhat (fixed-order fp64 combine)
__global__ __launch_bounds__(kT) void bn_bwd_finalize_kernel(const float* part, int nblk, int C, float* dgamma, float* dbeta)
{
    __shared__ double shd[kT];
    const int c = blockIdx.x * 8 + (threadIdx.x & 7);
    const double s = lanes32_sum(part, nblk, (int64_t)2 * C, c, c < C, shd);
    const double ss = lanes32_sum(part + C, nblk, (int64_t)2 * C, c, c < C, shd);
    if (c >= C || threadIdx.x >= 8) return;
    dbeta[c] = (float)s;
    dgamma[c] = (float)ss;
}

// dx = gamma*invstd * (g - dbeta/N - xhat*dgamma/N),  g = dy * act'(y).   eval_mode: dx = g*gamma*invstd.
// dres (optional) receives g (gradient of the residual branch, added before the activation).
__global__ __launch_bounds__(kT) void bn_bwd_apply_kernel(const float* x, int64_t ldx, const float* dy, int64_t lddy,
                                                         const float* yact, int64_t ldya, int act, const float* mean,
                                                         const float* invstd, const float* gamma, const float* dgamma,
                                                         const float* dbeta, float inv_count, float* dx, int64_t lddx,
                                                         float* dres, int64_t lddr, int64_t M, int cq)
{
    const int64_t total = M * cq;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
        const int64_t r = total <= 0xFFFFFFFFll ? (int64_t)((unsigned)e / (unsigned)cq) : e / cq;
        const int q = (int)(e - r * cq);
        float4 g = *reinterpret_cast<const float4*>(dy + r * lddy + q * 4);
        if (act != 0) {
            const float4 ya = *reinterpret_cast<const float4*>(yact + r * ldya + q * 4);
            g.x *= act_mask(ya.x, act); g.y *= act_mask(ya.y, act); g.z *= act_mask(ya.z, act); g.w *= act_mask(ya.w, act);
        }
        if (dres) *reinterpret_cast<float4*>(dres + r * lddr + q * 4) = g;
        const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + q * 4);
        const float4 mu = *reinterpret_cast<const float4*>(mean + q * 4);
        const float4 is = *reinterpret_cast<const float4*>(invstd + q * 4);
        const float4 ga = *reinterpret_cast<const float4*>(gamma + q * 4);
        const float4 dg = *reinterpret_cast<const float4*>(dgamma + q * 4);
        const float4 db = *reinterpret_cast<const float4*>(dbeta + q * 4);
        float4 o;
        o.x = ga.x * is.x * (g.x - db.x * inv_count - (v.x - mu.x) * is.x * dg.x * inv_count);
        o.y = ga.y * is.y * (g.y - db.y * inv_count - (v.y - mu.y) * is.y * dg.y * inv_count);
        o.z = ga.z * is.z * (g.z - db.z * inv_count - (v.z - mu.z) * is.z * dg.z * inv_count);
        o.w = ga.w * is.w * (g.w - db.w * inv_count - (v.w - mu.w) * is.w * dg.w * inv_count);
        *reinterpret_cast<float4*>(dx + r * lddx + q * 4) = o;
    }
}

// ================================================================================================
// Single-launch BatchNorm (training) forward and backward.
//
// The three-launch form above (column partials -> finalize -> apply) costs ~17 us forward / ~23 us backward per
// layer on the 1/16-resolution maps where each launch is latency-bound, and DeepLabv3+-MNv2 has 60 BN layers.
// Here one launch does all three: the grid is (channel strip) x (row chunk); a block reduces its chunk of its
// strip, publishes the partial, waits on a per-strip arrival counter (agent-scope release/acquire: the XCD L2s
// are not coherent with each other), then EVERY block of the strip combines the strip's partials in the same
// fixed order in fp64 (bit-identical in all of them, deterministic) and applies the normalisation to its own
// rows, which it re-reads from L2.  All blocks of a launch are co-resident (<= 1024 blocks of 256 threads on
// 256 CUs), so the wait cannot deadlock; R == 1 skips it.  The last block through a strip's counters zeroes them,
// so the caller's `sync` array stays zero between launches (it must not be shared by launches that can overlap).
// ================================================================================================
// counter-based RNG of the dropout kernels (mask = f(seed, flat element index)); also used by the BatchNorm epilogue
__device__ __forceinline__ uint32_t hash_rng(uint64_t seed, uint64_t idx)
{
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}

struct BnFusedGeom {
    int cq;        // C/4
    int bq;        // float4 columns per strip (4..8)
    int nrl;       // row lanes = 256 / bq
    int nstrips;   // cdiv(cq, bq)
    int R;         // row chunks per strip
    int64_t rows_per_chunk;
};

static int g_bn_target_blocks = 0;       // strips x row chunks aimed at (pp_debug_set_bn_target); 0 = one block per CU.  Round 1, in-process:
                                         // 128: 7.77, 192: 7.47, 256-384: 7.30-7.36, 512: 7.36, 768: 7.60, 1024: 7.76 ms/step (384 chosen).
                                         // Round 2: the launches of the head's backward run beside the SegmentHead weight gradient, whose
                                         // 168-VGPR / 48 KiB blocks sit two per CU and leave room for exactly ONE more block per CU; a
                                         // 376-block launch then waits for weight-gradient blocks to retire (246 us instead of ~50 in the
                                         // trace).  192: 6.85, 240-256: 6.69-6.73, 288: 6.74, 320: 6.77, 384: 6.76 ms/step.
static int bn_device_cus()
{
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) {
            (void)hipGetLastError();
            return 256;
        }
        return n;
    }();
    return cus;
}

constexpr int kBnRowCache = 12;                   // rows per thread the single-launch BatchNorm kernels may keep in registers between passes
static int g_bn_row_cache = 1;       // pp_debug_set_bn_bytes_per_block(-1) switches the register-cached variants off (A/B)
static int g_bn_bytes_per_block = 0;   // > 0: large maps get one block per this many bytes of x.  Off: measured neutral in
                                                   // isolation (tools/bn_bench.py: 33.6 MB forward 33.4 us with 384 blocks, 33.2-35.8 us with 640)
                                                   // and in the step (6.84 vs 6.86 ms) - a launch is ~13 us of fixed latency + bytes at ~5 TB/s

// Blocks of the single-launch BatchNorm kernels that can be resident on the device at once (occupancy x CUs), queried
// once per process (one process drives one GPU).  A launch never asks for more than HALF of it: the kernels wait for
// sibling blocks, so all blocks of a launch must become resident while other spin-waiting launches (a second stream, a
// second process on the same GPU) hold theirs - two such launches always fit side by side.  0: not known (no device).
static int bn_fused_capacity();

static BnFusedGeom bn_fused_geom(int64_t M, int C)
{
    BnFusedGeom g;
    g.cq = C / 4;
    g.bq = g.cq < 8 ? g.cq : 8;
    if (g.cq > 8 && g.cq % 8 != 0) {
        for (int b = 7; b >= 4; --b)
            if (g.cq % b == 0) { g.bq = b; break; }
    }
    g.nrl = kT / g.bq;
    g.nstrips = (int)cdiv(g.cq, g.bq);
    int64_t target = g_bn_target_blocks > 0 ? g_bn_target_blocks : bn_device_cus();
    if (g_bn_bytes_per_block > 0) {
        const int64_t by_bytes = M * (int64_t)C * 4 / g_bn_bytes_per_block;
        if (by_bytes > target) target = by_bytes;
    }
    const int cap = bn_fused_capacity();
    int64_t limit = cap > 0 ? cap / 2 : 1024;
    if (limit > 1024) limit = 1024;
    if (target > limit) target = limit;
    int64_t R = target / g.nstrips;
    if (R > 256) R = 256;
    if (R < 1 || (int64_t)g.nstrips * R > 1024) R = 1;
    int64_t rpc = cdiv(cdiv(M, R), g.nrl) * g.nrl;
    g.rows_per_chunk = rpc;
    g.R = (int)cdiv(M, rpc);
    return g;
}

// tree-sum the per-thread (s0,s1) over the row lanes of each column; result valid in row lane 0
__device__ __forceinline__ void rowlane_tree(float4& s0, float4& s1, float4 (*sh)[kT], int rl, int nrl, int bq)
{
    const int t = threadIdx.x;
    sh[0][t] = s0;
    sh[1][t] = s1;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (rl < off && rl + off < nrl) {
            const float4 a = sh[0][t + off * bq], b = sh[1][t + off * bq];
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
            sh[0][t] = s0;
            sh[1][t] = s1;
        }
        __syncthreads();
    }
}

// the element-wise dropout of dropout4_kernel on one float4 (same hash stream: flat index i0 .. i0+3)
__device__ __forceinline__ void drop4(float4& o, uint64_t i0, uint64_t seed, float p, float inv_keep)
{
    o.x = (float)(hash_rng(seed, i0 + 0) >> 8) * (1.0f / 16777216.0f) >= p ? o.x * inv_keep : 0.0f;
    o.y = (float)(hash_rng(seed, i0 + 1) >> 8) * (1.0f / 16777216.0f) >= p ? o.y * inv_keep : 0.0f;
    o.z = (float)(hash_rng(seed, i0 + 2) >> 8) * (1.0f / 16777216.0f) >= p ? o.z * inv_keep : 0.0f;
    o.w = (float)(hash_rng(seed, i0 + 3) >> 8) * (1.0f / 16777216.0f) >= p ? o.w * inv_keep : 0.0f;
}

struct BnFwdArgs {
    const float* x; int64_t ldx; int64_t M; int C;
    const float* gamma; const float* beta; float eps; float momentum;
    float* running_mean; float* running_var; float* mean; float* invstd;
    const float* res; int64_t ldr; int act; float* y; int64_t ldy;
    xword* part; int* sync; BnFusedGeom g;
    float drop_p, drop_inv_keep; uint64_t drop_seed; const uint64_t* drop_seed_dev;   // nn.Dropout after the activation (p = 0: none)
    // DW variant: x is not an input but the OUTPUT of a depthwise 3x3 convolution computed here (mobilenet_v2.py:38,52 -> :39,53)
    const float* dw_in; int64_t dw_ld; const float* dw_w; float* x_out;
    int dw_H, dw_W, dw_Ho, dw_Wo, dw_stride, dw_pad, dw_dil;
    // PARTIALS variant: the statistics were accumulated by the producer (convolution epilogue / split-K reduce):
    const float* stats; int stat_rows;      // [stat_rows][2][C] column sums and sums of squares
    unsigned long long* probe;              // pp_debug_set_bn_probe: [blocks][8] wall-clock stamps (100 MHz) of the kernel's phases, or NULL
};
#define BN_STAMP(i) do { if (a.probe && threadIdx.x == 0) a.probe[(int64_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)

// one output quad of the depthwise 3x3 convolution, same tap order / fma chain as dwconv_fwd_kernel (bit-identical)
__device__ __forceinline__ float4 dw_point(const BnFwdArgs& a, int64_t row, int q, const float4* wreg)
{
    const unsigned ru = (unsigned)row;
    const unsigned t = ru / (unsigned)a.dw_Wo;
    const int ow = (int)(ru - t * (unsigned)a.dw_Wo);
    const unsigned b = t / (unsigned)a.dw_Ho;
    const int oh = (int)(t - b * (unsigned)a.dw_Ho);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int th = 0; th < 3; ++th) {
        const int ih = oh * a.dw_stride - a.dw_pad + th * a.dw_dil;
        if ((unsigned)ih >= (unsigned)a.dw_H) continue;
#pragma unroll
        for (int tw = 0; tw < 3; ++tw) {
            const int iw = ow * a.dw_stride - a.dw_pad + tw * a.dw_dil;
            if ((unsigned)iw >= (unsigned)a.dw_W) continue;
            const float4 v = *reinterpret_cast<const float4*>(a.dw_in + (((int64_t)b * a.dw_H + ih) * a.dw_W + iw) * a.dw_ld + q * 4);
            const float4 ww = wreg[th * 3 + tw];           // the thread's nine weight quads, loaded once (the stores to x_out kept the
            acc.x = fmaf(v.x, ww.x, acc.x); acc.y = fmaf(v.y, ww.y, acc.y);   // compiler from hoisting them: nine more loads per row)
            acc.z = fmaf(v.z, ww.z, acc.z); acc.w = fmaf(v.w, ww.w, acc.w);
        }
    }
    return acc;
}

// MODE 0: statistics + apply in one launch (blocks exchange partials, see above); MODE 1: the same with the depthwise
// convolution computed in the statistics pass; MODE 2: the producer of x already wrote partial statistics - no first pass
// over x, NO exchange between blocks and no co-residency requirement: every block reduces the stat_rows partial rows of its
// channel strip itself (fixed order, fp64) and applies.
// NC > 0 (MODE 0 / 1): a thread's rows of its chunk (at most NC, launcher-checked) stay in registers between the statistics
// pass and the apply pass - same sums in the same order, no second read of x (one L2 round trip less per launch).
template <int MODE, int NC = 0>
__global__ __launch_bounds__(kT) void bn_fused_fwd_kernel(BnFwdArgs a)
{
    constexpr bool DW = MODE == 1;
    static_assert(NC == 0 || (MODE != 2 && NC % 4 == 0), "row cache: the kernels with a statistics pass");
    float4 keep[NC > 0 ? NC : 1];
    __shared__ unsigned sh_tag;
    __shared__ float4 sh[2][kT];
    __shared__ double shd[kT];
    __shared__ double tot[64];
    __shared__ float aff[2][32];
    const BnFusedGeom g = a.g;
    unsigned tag = 0, tag0 = 0;
    BN_STAMP(0);
    if constexpr (MODE != 2) tag0 = tag_issue(a.sync);
    const int t = threadIdx.x;
    const int strip = blockIdx.x % g.nstrips, chunk = blockIdx.x / g.nstrips;
    const int ql = t % g.bq, rl = t / g.bq;
    const int q = strip * g.bq + ql;
    const bool active = rl < g.nrl && q < g.cq;
    const int64_t r0 = (int64_t)chunk * g.rows_per_chunk;
    const int64_t r1 = r0 + g.rows_per_chunk < a.M ? r0 + g.rows_per_chunk : a.M;
    const float* xq = a.x + q * 4;
    const int nch = g.bq * 4, nout = nch * 2;
    if constexpr (MODE == 2) {
        // tot[stat * nch + cl] = sum over the partial rows, thread (sub, o): rows sub, sub + nsub, ... then the subs in order
        const int nsub = kT / nout;
        const int o = t % nout, sub = t / nout;
        const int stat = o / nch, cl = o - stat * nch;
        const int c = strip * nch + cl;
        double s = 0.0;
        if (sub < nsub && c < a.C) {
            const float* p = a.stats + (int64_t)stat * a.C + c;
            int r = sub;
            for (; r + 3 * nsub < a.stat_rows; r += 4 * nsub) {
                const float v0 = p[(int64_t)r * 2 * a.C], v1 = p[(int64_t)(r + nsub) * 2 * a.C];
                const float v2 = p[(int64_t)(r + 2 * nsub) * 2 * a.C], v3 = p[(int64_t)(r + 3 * nsub) * 2 * a.C];
                s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
            }
            for (; r < a.stat_rows; r += nsub) s += (double)p[(int64_t)r * 2 * a.C];
        }
        shd[t] = s;
        __syncthreads();
        if (t < nout) {
            double acc = shd[t];
            for (int k = 1; k < nsub; ++k) acc += shd[k * nout + t];
            tot[t] = acc;
        }
        __syncthreads();
    } else {
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    float4 wreg[DW ? 9 : 1];
    if constexpr (DW) {
        if (active) {
#pragma unroll
            for (int k = 0; k < 9; ++k) wreg[k] = *reinterpret_cast<const float4*>(a.dw_w + k * a.C + q * 4);
        }
    }
    // four rows r, r + nrl, ...: loads, then the accumulation in the order every variant uses
    auto stat_load = [&](int64_t r, float4* v, float* w) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t rr = r + (int64_t)j * g.nrl;
            w[j] = rr < r1 ? 1.0f : 0.0f;
            if constexpr (DW) {
                v[j] = dw_point(a, rr < r1 ? rr : r1 - 1, q, wreg);
                if (rr < r1) *reinterpret_cast<float4*>(a.x_out + rr * a.ldx + q * 4) = v[j];   // BN's input, kept for backward
            } else {
                v[j] = *reinterpret_cast<const float4*>(xq + (rr < r1 ? rr : r1 - 1) * a.ldx);
            }
        }
    };
    auto stat_acc = [&](const float4* v, const float* w) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 u = make_float4(v[j].x * w[j], v[j].y * w[j], v[j].z * w[j], v[j].w * w[j]);
            s0.x += u.x; s0.y += u.y; s0.z += u.z; s0.w += u.w;
            s1.x = fmaf(u.x, u.x, s1.x); s1.y = fmaf(u.y, u.y, s1.y);
            s1.z = fmaf(u.z, u.z, s1.z); s1.w = fmaf(u.w, u.w, s1.w);
        }
    };
    if (active) {
        if constexpr (NC > 0) {
#pragma unroll
            for (int it = 0; it < NC / 4; ++it) {
                const int64_t r = r0 + rl + (int64_t)it * g.nrl * 4;
                if (r < r1) {
                    float w[4];
                    stat_load(r, &keep[it * 4], w);
                    stat_acc(&keep[it * 4], w);
                }
            }
        } else if constexpr (DW) {
            for (int64_t r = r0 + rl; r < r1; r += (int64_t)g.nrl * 4) {
                float4 v[4];
                float w[4];
                stat_load(r, v, w);
                stat_acc(v, w);
            }
        } else {
            // large maps run ONE block per CU (beside the weight-gradient blocks): eight rows of loads in flight per thread, or the
            // launch is bound by its own memory-level parallelism (256 blocks x 4 loads x 1 KiB = 1 MB in flight chip-wide)
            for (int64_t r = r0 + rl; r < r1; r += (int64_t)g.nrl * 8) {
                float4 va[4], vb[4];
                float wa[4], wb[4];
                stat_load(r, va, wa);
                stat_load(r + (int64_t)g.nrl * 4, vb, wb);       // rows past r1: weight 0, clamped address
                stat_acc(va, wa);
                stat_acc(vb, wb);
            }
        }
    }
    BN_STAMP(1);
    tag_share(tag0, &sh_tag);
    rowlane_tree(s0, s1, sh, rl, g.nrl, g.bq);
    tag = sh_tag;
    BN_STAMP(2);
    if (rl == 0) {
        publish_partial(a.part + ((int64_t)strip * g.R + chunk) * nout + ql * 4, nch, s0, s1, tag);
    }
    BN_STAMP(3);
    strip_combine(a.part, strip, g.R, nout, tag, shd, tot);
    BN_STAMP(4);
    launch_done(a.sync);
    }
    if (t < nch) {
        const int c = strip * nch + t;
        if (c < a.C) {
            const double count = (double)a.M;
            const double mu = tot[t] / count;
            double var = tot[nch + t] / count - mu * mu;
            if (var < 0.0) var = 0.0;
            const float is = (float)(1.0 / sqrt(var + (double)a.eps));
            const float sc = a.gamma[c] * is;
            aff[0][t] = sc;
            aff[1][t] = a.beta[c] - (float)mu * sc;
            if (chunk == 0) {
                a.mean[c] = (float)mu;
                a.invstd[c] = is;
                if (a.running_mean) {
                    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
                    a.running_mean[c] = (1.0f - a.momentum) * a.running_mean[c] + a.momentum * (float)mu;
                    a.running_var[c] = (1.0f - a.momentum) * a.running_var[c] + a.momentum * (float)unbiased;
                }
            }
        }
    }
    __syncthreads();
    if (!active) return;
    const float4 sc = *reinterpret_cast<const float4*>(&aff[0][ql * 4]);
    const float4 sf = *reinterpret_cast<const float4*>(&aff[1][ql * 4]);
    float* yq = a.y + q * 4;
    const float* rq = a.res ? a.res + q * 4 : nullptr;
    const int act = a.act;
    const bool drop = a.drop_p > 0.0f;
    uint64_t dseed = a.drop_seed;
    if (drop && a.drop_seed_dev) dseed += *a.drop_seed_dev * 0x9E3779B97F4A7C15ull;      // as dropout4_kernel
    if constexpr (NC > 0) {
#pragma unroll
        for (int e = 0; e < NC; ++e) {
            const int64_t r = r0 + rl + (int64_t)e * g.nrl;
            if (r >= r1) break;
            const float4 va = keep[e];
            float4 oa;
            oa.x = fmaf(va.x, sc.x, sf.x); oa.y = fmaf(va.y, sc.y, sf.y); oa.z = fmaf(va.z, sc.z, sf.z); oa.w = fmaf(va.w, sc.w, sf.w);
            if (rq) {
                const float4 ra = *reinterpret_cast<const float4*>(rq + r * a.ldr);
                oa.x += ra.x; oa.y += ra.y; oa.z += ra.z; oa.w += ra.w;
            }
            oa.x = act_fwd(oa.x, act); oa.y = act_fwd(oa.y, act); oa.z = act_fwd(oa.z, act); oa.w = act_fwd(oa.w, act);
            if (drop) drop4(oa, (uint64_t)(r * g.cq + q) * 4, dseed, a.drop_p, a.drop_inv_keep);
            *reinterpret_cast<float4*>(yq + r * a.ldy) = oa;
        }
        BN_STAMP(5);
        return;
    }
    for (int64_t r = r0 + rl; r < r1; r += (int64_t)g.nrl * 4) {
        float4 v[4], rv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t rr = r + (int64_t)j * g.nrl;
            const int64_t rc = rr < r1 ? rr : r;
            v[j] = *reinterpret_cast<const float4*>(xq + rc * a.ldx);
            rv[j] = rq ? *reinterpret_cast<const float4*>(rq + rc * a.ldr) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t rr = r + (int64_t)j * g.nrl;
            if (rr >= r1) break;
            float4 o;
            o.x = fmaf(v[j].x, sc.x, sf.x); o.y = fmaf(v[j].y, sc.y, sf.y); o.z = fmaf(v[j].z, sc.z, sf.z); o.w = fmaf(v[j].w, sc.w, sf.w);
            if (rq) { o.x += rv[j].x; o.y += rv[j].y; o.z += rv[j].z; o.w += rv[j].w; }
            o.x = act_fwd(o.x, act); o.y = act_fwd(o.y, act); o.z = act_fwd(o.z, act); o.w = act_fwd(o.w, act);
            if (drop) drop4(o, (uint64_t)(rr * g.cq + q) * 4, dseed, a.drop_p, a.drop_inv_keep);
            *reinterpret_cast<float4*>(yq + rr * a.ldy) = o;
        }
    }
    BN_STAMP(5);
}

struct BnBwdArgs {
    const float* x; int64_t ldx; const float* dy; int64_t lddy; const float* yact; int64_t ldya; int act;
    int64_t M; int C; const float* mean; const float* invstd; const float* gamma; float* dgamma; float* dbeta;
    float* dx; int64_t lddx; float* dres; int64_t lddr; xword* part; int* sync; BnFusedGeom g;
    float gscale;     // 1/(1-p) of a dropout fused after the activation (its mask is y_act == 0), else 1
    const float* beta_mask;   // non-NULL (and yact NULL): the activation mask is recomputed from x, z = x*scale + shift
    const unsigned char* rowflag;   // SPARSE variant: rowflag[r] == 0 promises that row r of dy is all zeros (never read then)
};

// NC > 0: a thread's (at most NC, launcher-checked) rows of x and of the masked, scaled dy stay in registers between the
// reduction pass and the dx pass (same values, same order; the mask source is not read twice either).
// SPARSE (NC == 0 only): the gradient is zero except in the rows whose flag is set - the loss of a sparsely labelled batch
// (model.py:113-119: 20 labelled pixels per image, ignore_index elsewhere) leaves <= 4 x 80 non-zero rows of 32768 behind the
// classifier.  The statistics pass visits only those rows, in the order the dense kernel adds them (the skipped terms are exact
// zeros: the sums are bit-equal), the dx pass reads dy and the mask source only there: x is read once and dy almost never.
__device__ __attribute__((aligned(16))) float g_bn_zero4[4] = {0.f, 0.f, 0.f, 0.f};
template <int NC, bool SPARSE = false>
__global__ __launch_bounds__(kT) void bn_fused_bwd_kernel(BnBwdArgs a)
{
    static_assert(!SPARSE || NC == 0, "row flags: the uncached variant");
    __shared__ unsigned sh_tag;
    __shared__ float4 sh[2][kT];
    __shared__ double shd[kT];
    __shared__ double tot[64];
    __shared__ float red[2][32];
    static_assert(NC % 2 == 0, "row cache: pairs of rows");
    float4 keepx[NC > 0 ? NC : 1], keepu[NC > 0 ? NC : 1];
    const BnFusedGeom g = a.g;
    const unsigned tag0 = tag_issue(a.sync);
    const int t = threadIdx.x;
    const int strip = blockIdx.x % g.nstrips, chunk = blockIdx.x / g.nstrips;
    const int ql = t % g.bq, rl = t / g.bq;
    const int q = strip * g.bq + ql;
    const bool active = rl < g.nrl && q < g.cq;
    const int64_t r0 = (int64_t)chunk * g.rows_per_chunk;
    const int64_t r1 = r0 + g.rows_per_chunk < a.M ? r0 + g.rows_per_chunk : a.M;
    const float* xq = a.x + q * 4;
    const float* gq = a.dy + q * 4;
    const float* aq = a.yact ? a.yact + q * 4 : nullptr;
    // mask source: the saved activation output, or (no residual, no dropout) the forward's own z = fma(x, scale, shift)
    // recomputed from x, which is read anyway: one tensor less in both passes
    const bool remask = aq == nullptr && a.beta_mask != nullptr && a.act != 0;
    const int act = (aq || remask) ? a.act : 0;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, mu = s0, is = s0, zsc = s0, zsf = s0;
    if (active) {
        mu = *reinterpret_cast<const float4*>(a.mean + q * 4);
        is = *reinterpret_cast<const float4*>(a.invstd + q * 4);
        if (remask) {
            const float4 gm = *reinterpret_cast<const float4*>(a.gamma + q * 4), be = *reinterpret_cast<const float4*>(a.beta_mask + q * 4);
            zsc = make_float4(gm.x * is.x, gm.y * is.y, gm.z * is.z, gm.w * is.w);
            zsf = make_float4(be.x - mu.x * zsc.x, be.y - mu.y * zsc.y, be.z - mu.z * zsc.z, be.w - mu.w * zsc.w);
        }
        // two rows r, r + nrl: loads, then the accumulation in the order every variant uses; vk / uk (row cache) receive x and
        // mask * dy * gscale
        auto red_load = [&](int64_t r, float4* v, float4* gg, float4* ya, float* w) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int64_t rr = r + (int64_t)j * g.nrl;
                w[j] = rr < r1 ? 1.0f : 0.0f;
                const int64_t rc = rr < r1 ? rr : r1 - 1;
                v[j] = *reinterpret_cast<const float4*>(xq + rc * a.ldx);
                gg[j] = *reinterpret_cast<const float4*>(gq + rc * a.lddy);
                ya[j] = (act != 0 && !remask) ? *reinterpret_cast<const float4*>(aq + rc * a.ldya) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto red_acc = [&](const float4* v, const float4* gg, float4* ya, const float* w, float4* vk, float4* uk) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float4 u = gg[j];
                if (remask)
                    ya[j] = make_float4(fmaf(v[j].x, zsc.x, zsf.x), fmaf(v[j].y, zsc.y, zsf.y), fmaf(v[j].z, zsc.z, zsf.z),
                                        fmaf(v[j].w, zsc.w, zsf.w));
                if (act != 0) {
                    u.x *= act_mask(ya[j].x, act); u.y *= act_mask(ya[j].y, act);
                    u.z *= act_mask(ya[j].z, act); u.w *= act_mask(ya[j].w, act);
                }
                if constexpr (NC > 0) {
                    vk[j] = v[j];
                    uk[j] = make_float4(u.x * a.gscale, u.y * a.gscale, u.z * a.gscale, u.w * a.gscale);   // what the dx pass computes
                }
                const float ws_ = w[j] * a.gscale;
                u.x *= ws_; u.y *= ws_; u.z *= ws_; u.w *= ws_;
                s0.x += u.x; s0.y += u.y; s0.z += u.z; s0.w += u.w;
                s1.x = fmaf(u.x, (v[j].x - mu.x) * is.x, s1.x); s1.y = fmaf(u.y, (v[j].y - mu.y) * is.y, s1.y);
                s1.z = fmaf(u.z, (v[j].z - mu.z) * is.z, s1.z); s1.w = fmaf(u.w, (v[j].w - mu.w) * is.w, s1.w);
            }
        };
        if constexpr (NC > 0) {
#pragma unroll
            for (int it = 0; it < NC / 2; ++it) {
                const int64_t r = r0 + rl + (int64_t)it * g.nrl * 2;
                if (r < r1) {
                    float4 v[2], gg[2], ya[2];
                    float w[2];
                    red_load(r, v, gg, ya, w);
                    red_acc(v, gg, ya, w, &keepx[it * 2], &keepu[it * 2]);
                }
            }
        } else {
            if constexpr (SPARSE) {
                for (int64_t rb = r0 + rl; rb < r1; rb += (int64_t)g.nrl * 8) {   // flags of eight rows at a time (independent loads)
                  unsigned char fl8[8];
#pragma unroll
                  for (int j = 0; j < 8; ++j) {
                      const int64_t rj = rb + (int64_t)j * g.nrl;
                      fl8[j] = rj < r1 ? a.rowflag[rj] : (unsigned char)0;
                  }
#pragma unroll
                  for (int j = 0; j < 8; ++j) {                           // the dense order: r, r + nrl, r + 2 nrl, ...
                    if (fl8[j] == 0) continue;
                    const int64_t r = rb + (int64_t)j * g.nrl;
                    float4 v = *reinterpret_cast<const float4*>(xq + r * a.ldx);
                    float4 u = *reinterpret_cast<const float4*>(gq + r * a.lddy);
                    if (act != 0) {
                        const float4 ya = remask ? make_float4(fmaf(v.x, zsc.x, zsf.x), fmaf(v.y, zsc.y, zsf.y), fmaf(v.z, zsc.z, zsf.z),
                                                               fmaf(v.w, zsc.w, zsf.w))
                                                 : *reinterpret_cast<const float4*>(aq + r * a.ldya);
                        u.x *= act_mask(ya.x, act); u.y *= act_mask(ya.y, act); u.z *= act_mask(ya.z, act); u.w *= act_mask(ya.w, act);
                    }
                    const float ws_ = 1.0f * a.gscale;
                    u.x *= ws_; u.y *= ws_; u.z *= ws_; u.w *= ws_;
                    s0.x += u.x; s0.y += u.y; s0.z += u.z; s0.w += u.w;
                    s1.x = fmaf(u.x, (v.x - mu.x) * is.x, s1.x); s1.y = fmaf(u.y, (v.y - mu.y) * is.y, s1.y);
                    s1.z = fmaf(u.z, (v.z - mu.z) * is.z, s1.z); s1.w = fmaf(u.w, (v.w - mu.w) * is.w, s1.w);
                  }
                }
            } else
            // large maps, one block per CU: four rows (8-12 float4 loads) in flight per thread (see the forward kernel)
            for (int64_t r = r0 + rl; r < r1; r += (int64_t)g.nrl * 4) {
                float4 va[2], ga[2], yaa[2], vb[2], gb[2], yab[2];
                float wa[2], wb[2];
                red_load(r, va, ga, yaa, wa);
                red_load(r + (int64_t)g.nrl * 2, vb, gb, yab, wb);   // rows past r1: weight 0, clamped address
                red_acc(va, ga, yaa, wa, nullptr, nullptr);
                red_acc(vb, gb, yab, wb, nullptr, nullptr);
            }
        }
    }
    tag_share(tag0, &sh_tag);
    rowlane_tree(s0, s1, sh, rl, g.nrl, g.bq);
    const unsigned tag = sh_tag;
    const int nch = g.bq * 4, nout = nch * 2;
    if (rl == 0) {
        publish_partial(a.part + ((int64_t)strip * g.R + chunk) * nout + ql * 4, nch, s0, s1, tag);
    }
    strip_combine(a.part, strip, g.R, nout, tag, shd, tot);
    launch_done(a.sync);
    if (t < nch) {
        const int c = strip * nch + t;
        const float db = (float)tot[t], dg = (float)tot[nch + t];
        red[0][t] = db;
        red[1][t] = dg;
        if (c < a.C && chunk == 0) {
            a.dbeta[c] = db;
            a.dgamma[c] = dg;
        }
    }
    __syncthreads();
    if (!active) return;
    const float inv_count = 1.0f / (float)a.M;
    const float4 db = *reinterpret_cast<const float4*>(&red[0][ql * 4]);
    const float4 dg = *reinterpret_cast<const float4*>(&red[1][ql * 4]);
    const float4 ga = *reinterpret_cast<const float4*>(a.gamma + q * 4);
    float* dxq = a.dx + q * 4;
    float* drq = a.dres ? a.dres + q * 4 : nullptr;
    if constexpr (NC > 0) {
#pragma unroll
        for (int e = 0; e < NC; ++e) {
            const int64_t r = r0 + rl + (int64_t)e * g.nrl;
            if (r >= r1) break;
            const float4 u = keepu[e], v = keepx[e];
            if (drq) *reinterpret_cast<float4*>(drq + r * a.lddr) = u;
            float4 o;
            o.x = bn_dx(u.x, v.x, mu.x, is.x, ga.x, db.x, dg.x, inv_count);
            o.y = bn_dx(u.y, v.y, mu.y, is.y, ga.y, db.y, dg.y, inv_count);
            o.z = bn_dx(u.z, v.z, mu.z, is.z, ga.z, db.z, dg.z, inv_count);
            o.w = bn_dx(u.w, v.w, mu.w, is.w, ga.w, db.w, dg.w, inv_count);
            *reinterpret_cast<float4*>(dxq + r * a.lddx) = o;
        }
        return;
    }
    // rows in flight per thread in the dx pass: the sparse form has nothing but this pass to hide its (cold) read of x behind
    // and runs one block per CU - eight rows (four measured 51 us on the 33.6 MB head map: 1.7 TB/s)
    constexpr int RW = SPARSE ? 8 : 4;
    unsigned char fnext[RW] = {};
    if constexpr (SPARSE) {
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int64_t rr = r0 + rl + (int64_t)j * g.nrl;
            fnext[j] = rr < r1 ? a.rowflag[rr] : (unsigned char)0;
        }
    }
    for (int64_t r = r0 + rl; r < r1; r += (int64_t)g.nrl * RW) {
        float4 uu[RW], vv[RW], yy[RW];
        unsigned char fcur[RW];
#pragma unroll
        for (int j = 0; j < RW; ++j) fcur[j] = fnext[j];
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int64_t rr = r + (int64_t)j * g.nrl;
            const int64_t rc = rr < r1 ? rr : r;
            if constexpr (SPARSE) {
                // unflagged rows: dy (and the mask source) come from a zero word - the load stays unconditional, the row is not read
                const bool fl = rr < r1 && fcur[j] != 0;
                {   // the flags of the next group fly with this group's loads
                    const int64_t rn = rr + (int64_t)g.nrl * RW;
                    fnext[j] = rn < r1 ? a.rowflag[rn] : (unsigned char)0;
                }
                uu[j] = *reinterpret_cast<const float4*>(fl ? gq + rc * a.lddy : g_bn_zero4);
                yy[j] = (act != 0 && !remask) ? *reinterpret_cast<const float4*>(fl ? aq + rc * a.ldya : g_bn_zero4) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                uu[j] = *reinterpret_cast<const float4*>(gq + rc * a.lddy);
                yy[j] = (act != 0 && !remask) ? *reinterpret_cast<const float4*>(aq + rc * a.ldya) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            vv[j] = *reinterpret_cast<const float4*>(xq + rc * a.ldx);
        }
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int64_t rr = r + (int64_t)j * g.nrl;
            if (rr >= r1) break;
            float4 u = uu[j];
            const float4 v = vv[j];
            if (act != 0) {
                const float4 ya = remask ? make_float4(fmaf(v.x, zsc.x, zsf.x), fmaf(v.y, zsc.y, zsf.y), fmaf(v.z, zsc.z, zsf.z),
                                                       fmaf(v.w, zsc.w, zsf.w))
                                         : yy[j];
                u.x *= act_mask(ya.x, act); u.y *= act_mask(ya.y, act); u.z *= act_mask(ya.z, act); u.w *= act_mask(ya.w, act);
            }
            u.x *= a.gscale; u.y *= a.gscale; u.z *= a.gscale; u.w *= a.gscale;
            if (drq) *reinterpret_cast<float4*>(drq + rr * a.lddr) = u;
            float4 o;
            o.x = bn_dx(u.x, v.x, mu.x, is.x, ga.x, db.x, dg.x, inv_count);
            o.y = bn_dx(u.y, v.y, mu.y, is.y, ga.y, db.y, dg.y, inv_count);
            o.z = bn_dx(u.z, v.z, mu.z, is.z, ga.z, db.z, dg.z, inv_count);
            o.w = bn_dx(u.w, v.w, mu.w, is.w, ga.w, db.w, dg.w, inv_count);
            *reinterpret_cast<float4*>(dxq + rr * a.lddx) = o;
        }
    }
}

static int bn_fused_capacity()
{
    static const int cap = [] {
        int dev = 0, cus = 0, a = 0, b = 0, c = 0;
        if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
        int d = 0, e = 0, f = 0, sp = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, bn_fused_fwd_kernel<0, 0>, kT, 0) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, bn_fused_fwd_kernel<1, 0>, kT, 0) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&c, bn_fused_bwd_kernel<0>, kT, 0) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&sp, (bn_fused_bwd_kernel<0, true>), kT, 0) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&d, bn_fused_fwd_kernel<0, kBnRowCache>, kT, 0) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&e, bn_fused_bwd_kernel<kBnRowCache>, kT, 0) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&f, bn_fused_fwd_kernel<1, kBnRowCache>, kT, 0) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        int per = a < b ? a : b;
        per = per < c ? per : c;
        per = per < d ? per : d;
        per = per < e ? per : e;
        per = per < f ? per : f;
        per = per < sp ? per : sp;
        return per * cus;
    }();
    return reserve_scaled(cap, bn_device_cus());      // minus the CUs set aside for a resident communication kernel
}

// ================================================================================================
// depthwise 3x3 (weights [3][3][C]); generic stride / dilation / padding.
// ================================================================================================
__global__ __launch_bounds__(kT) void dwconv_fwd_kernel(const float* x, int64_t ldx, int B, int H, int W, int cq,
                                                       const float* w, int stride, int pad, int dil, float* y,
                                                       int64_t ldy, int Ho, int Wo, Epilogue epi)
{
    const int C = cq * 4;
    const int64_t total = (int64_t)B * Ho * Wo * cq;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
        int q, ow, oh, b;
        decode_bhwq(e, cq, Wo, Ho, q, ow, oh, b);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // all nine loads are UNCONDITIONAL (from a clamped pixel, zeroed afterwards where the tap falls outside): behind a branch per
        // tap the loads of a row waited for the row before - three dependent round trips per output (profiles/r06_dw_layers.txt)
        float4 v[9];
#pragma unroll
        for (int th = 0; th < 3; ++th) {
            const int ih = oh * stride - pad + th * dil;
            const bool rok = (unsigned)ih < (unsigned)H;
#pragma unroll
            for (int tw = 0; tw < 3; ++tw) {
                const int iw = ow * stride - pad + tw * dil;
                const bool ok = rok && (unsigned)iw < (unsigned)W;
                const float4 t = *reinterpret_cast<const float4*>(x + (((int64_t)b * H + (ok ? ih : 0)) * W + (ok ? iw : 0)) * ldx + q * 4);
                v[th * 3 + tw] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4 ww = *reinterpret_cast<const float4*>(w + t * C + q * 4);
            acc.x = fmaf(v[t].x, ww.x, acc.x); acc.y = fmaf(v[t].y, ww.y, acc.y);
            acc.z = fmaf(v[t].z, ww.z, acc.z); acc.w = fmaf(v[t].w, ww.w, acc.w);
        }
        const int64_t row = ((int64_t)b * Ho + oh) * Wo + ow;
        if (epi.gamma) {       // inference: folded eval-mode BatchNorm (same arithmetic as bn_eval_affine + bn_apply)
            const float4 g = *reinterpret_cast<const float4*>(epi.gamma + q * 4), be = *reinterpret_cast<const float4*>(epi.beta + q * 4);
            const float4 mu = *reinterpret_cast<const float4*>(epi.mean + q * 4), va = *reinterpret_cast<const float4*>(epi.var + q * 4);
            float sc;
            sc = g.x * (1.0f / sqrtf(va.x + epi.eps)); acc.x = fmaf(acc.x, sc, be.x - mu.x * sc);
            sc = g.y * (1.0f / sqrtf(va.y + epi.eps)); acc.y = fmaf(acc.y, sc, be.y - mu.y * sc);
            sc = g.z * (1.0f / sqrtf(va.z + epi.eps)); acc.z = fmaf(acc.z, sc, be.z - mu.z * sc);
            sc = g.w * (1.0f / sqrtf(va.w + epi.eps)); acc.w = fmaf(acc.w, sc, be.w - mu.w * sc);
        }
        if (epi.res) {
            const float4 rr = *reinterpret_cast<const float4*>(epi.res + row * epi.ldr + q * 4);
            acc.x += rr.x; acc.y += rr.y; acc.z += rr.z; acc.w += rr.w;
        }
        acc.x = epi_act(acc.x, epi.act); acc.y = epi_act(acc.y, epi.act); acc.z = epi_act(acc.z, epi.act); acc.w = epi_act(acc.w, epi.act);
        *reinterpret_cast<float4*>(y + row * ldy + q * 4) = acc;
    }
}

// Stride 1, dilation 1 (13 of MobileNetV2's 17 depthwise layers): one thread produces FOUR neighbouring outputs of a
// row, so the 3x6 input window and the 9 weights are loaded once for them (27 float4 loads per 4 outputs instead of
// 72).  FLIP == false: forward (taps read x at (oh - pad + th, ow - pad + tw)); FLIP == true: backward-data of the
// same layer (dx(ih,iw) = sum dy(ih + pad - th, iw + pad - tw) w[th][tw], i.e. the forward with the weights
// mirrored and padding 2 - pad).
template <bool FLIP, int DIL = 1>
__global__ __launch_bounds__(kT) void dwconv_s1_x4_kernel(const float* x, int64_t ldx, int B, int H, int W, int cq,
                                                         const float* w, int pad, float* y, int64_t ldy, int Ho, int Wo,
                                                         Epilogue epi)
{
    // DIL = 2: the dilated last block of the backbone (features.17, mobilenet_v2.py:99-106): a 3 x 8 window for four outputs, same tap
    // order and fma chain as the generic kernel it replaces there (bit-identical)
    constexpr int NW = 4 + 2 * DIL;
    const int C = cq * 4;
    const int wq = (Wo + 3) / 4;
    const int64_t total = (int64_t)B * Ho * wq * cq;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
        int q, owq, oh, b;
        decode_bhwq(e, cq, wq, Ho, q, owq, oh, b);
        const int ow0 = owq * 4;
        float4 acc[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int th = 0; th < 3; ++th) {
            const int ih = oh - pad + th * DIL;
            if ((unsigned)ih >= (unsigned)H) continue;
            const float* row = x + ((int64_t)b * H + ih) * W * ldx + q * 4;
            float4 v[NW];
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                const int iw = ow0 - pad + j;
                v[j] = (unsigned)iw < (unsigned)W ? *reinterpret_cast<const float4*>(row + (int64_t)iw * ldx)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int tw = 0; tw < 3; ++tw) {
                const int wi = FLIP ? (2 - th) * 3 + (2 - tw) : th * 3 + tw;
                const float4 ww = *reinterpret_cast<const float4*>(w + wi * C + q * 4);
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    acc[o].x = fmaf(v[o + tw * DIL].x, ww.x, acc[o].x); acc[o].y = fmaf(v[o + tw * DIL].y, ww.y, acc[o].y);
                    acc[o].z = fmaf(v[o + tw * DIL].z, ww.z, acc[o].z); acc[o].w = fmaf(v[o + tw * DIL].w, ww.w, acc[o].w);
                }
            }
        }
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sf = make_float4(0.f, 0.f, 0.f, 0.f);
        if (epi.gamma) {
            const float4 g = *reinterpret_cast<const float4*>(epi.gamma + q * 4), be = *reinterpret_cast<const float4*>(epi.beta + q * 4);
            const float4 mu = *reinterpret_cast<const float4*>(epi.mean + q * 4), va = *reinterpret_cast<const float4*>(epi.var + q * 4);
            sc.x = g.x * (1.0f / sqrtf(va.x + epi.eps)); sf.x = be.x - mu.x * sc.x;
            sc.y = g.y * (1.0f / sqrtf(va.y + epi.eps)); sf.y = be.y - mu.y * sc.y;
            sc.z = g.z * (1.0f / sqrtf(va.z + epi.eps)); sf.z = be.z - mu.z * sc.z;
            sc.w = g.w * (1.0f / sqrtf(va.w + epi.eps)); sf.w = be.w - mu.w * sc.w;
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            if (ow0 + o >= Wo) break;
            const int64_t r = ((int64_t)b * Ho + oh) * Wo + ow0 + o;
            float4 a = acc[o];
            if (epi.gamma) {
                a.x = fmaf(a.x, sc.x, sf.x); a.y = fmaf(a.y, sc.y, sf.y); a.z = fmaf(a.z, sc.z, sf.z); a.w = fmaf(a.w, sc.w, sf.w);
            }
            if (epi.res) {
                const float4 rr = *reinterpret_cast<const float4*>(epi.res + r * epi.ldr + q * 4);
                a.x += rr.x; a.y += rr.y; a.z += rr.z; a.w += rr.w;
            }
            a.x = epi_act(a.x, epi.act); a.y = epi_act(a.y, epi.act); a.z = epi_act(a.z, epi.act); a.w = epi_act(a.w, epi.act);
            *reinterpret_cast<float4*>(y + r * ldy + q * 4) = a;
        }
    }
}

// Depthwise forward BETWEEN two training BatchNorms (mobilenet_v2.py:48-56: pw -> BN -> ReLU6 -> dw -> BN -> ReLU6 -> pw):
//   * in_scale / in_shift / in_act (optional): the producer's BatchNorm + activation applied WHERE THE INPUT IS LOADED,
//     v = act(fma(x, scale, shift)) - the arithmetic of bn_apply_kernel, so the sums equal those over the materialised tensor
//     bit for bit; taps that fall into the zero padding stay zero (the padding is applied to the activated tensor);
//   * stats (optional): column sums / sums of squares of the outputs this block stores -> stats[blockIdx.x][2][C] (the partial
//     rows bn_finalize_kernel combines), so the BatchNorm BEHIND this convolution needs no statistics pass either.
// Thread map as col_reduce_kernel (a thread keeps ONE channel quad, its sums stay in registers; consecutive threads walk the
// channel axis).  X4: one work item = four neighbouring outputs of a row (stride 1, dilation 1; 18 loads per item).
template <bool X4>
__global__ __launch_bounds__(kT) void dwconv_fwd_fused_kernel(const float* x, int64_t ldx, int B, int H, int W, int cq, const float* w,
                                                             int stride, int pad, int dil, const float* in_scale, const float* in_shift,
                                                             int in_act, float* y, int64_t ldy, int Ho, int Wo, ColReduceGeom g,
                                                             float* stats)
{
    __shared__ float4 sh[2][kT];
    const int C = cq * 4;
    const int t = threadIdx.x;
    const int ql = t % g.cq_blk, ry = t / g.cq_blk;
    const int q = blockIdx.y * g.cq_blk + ql;
    const bool active = ry < g.rows_per_pass && q < cq;
    const int wq = X4 ? (Wo + 3) / 4 : Wo;
    const int64_t items = (int64_t)B * Ho * wq;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    if (active) {
        float4 ww[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) ww[k] = *reinterpret_cast<const float4*>(w + k * C + q * 4);
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sf = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool aff = in_scale != nullptr;
        if (aff) {
            sc = *reinterpret_cast<const float4*>(in_scale + q * 4);
            sf = *reinterpret_cast<const float4*>(in_shift + q * 4);
        }
        auto ld = [&](const float* ptr) -> float4 {
            float4 v = *reinterpret_cast<const float4*>(ptr);
            if (aff) {
                v.x = act_fwd(fmaf(v.x, sc.x, sf.x), in_act); v.y = act_fwd(fmaf(v.y, sc.y, sf.y), in_act);
                v.z = act_fwd(fmaf(v.z, sc.z, sf.z), in_act); v.w = act_fwd(fmaf(v.w, sc.w, sf.w), in_act);
            }
            return v;
        };
        const int64_t r0 = (int64_t)blockIdx.x * g.rows_per_block;
        const int64_t r1 = r0 + g.rows_per_block < items ? r0 + g.rows_per_block : items;
        for (int64_t r = r0 + ry; r < r1; r += g.rows_per_pass) {
            const unsigned ru = (unsigned)r;                         // items < 2^31 (checked on the host)
            const unsigned tt = ru / (unsigned)wq;
            const int owq = (int)(ru - tt * (unsigned)wq);
            const int b = (int)(tt / (unsigned)Ho);
            const int oh = (int)(tt - (unsigned)b * (unsigned)Ho);
            if constexpr (X4) {
                const int ow0 = owq * 4;
                float4 acc[4];
#pragma unroll
                for (int o = 0; o < 4; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int th = 0; th < 3; ++th) {
                    const int ih = oh - pad + th;
                    if ((unsigned)ih >= (unsigned)H) continue;
                    const float* row = x + ((int64_t)b * H + ih) * W * ldx + q * 4;
                    float4 v[6];
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        const int iw = ow0 - pad + j;
                        v[j] = (unsigned)iw < (unsigned)W ? ld(row + (int64_t)iw * ldx) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int tw = 0; tw < 3; ++tw) {
                        const float4 wk = ww[th * 3 + tw];
#pragma unroll
                        for (int o = 0; o < 4; ++o) {
                            acc[o].x = fmaf(v[o + tw].x, wk.x, acc[o].x); acc[o].y = fmaf(v[o + tw].y, wk.y, acc[o].y);
                            acc[o].z = fmaf(v[o + tw].z, wk.z, acc[o].z); acc[o].w = fmaf(v[o + tw].w, wk.w, acc[o].w);
                        }
                    }
                }
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    if (ow0 + o >= Wo) break;
                    const int64_t ro = ((int64_t)b * Ho + oh) * Wo + ow0 + o;
                    *reinterpret_cast<float4*>(y + ro * ldy + q * 4) = acc[o];
                    s1.x += acc[o].x; s1.y += acc[o].y; s1.z += acc[o].z; s1.w += acc[o].w;
                    s2.x = fmaf(acc[o].x, acc[o].x, s2.x); s2.y = fmaf(acc[o].y, acc[o].y, s2.y);
                    s2.z = fmaf(acc[o].z, acc[o].z, s2.z); s2.w = fmaf(acc[o].w, acc[o].w, s2.w);
                }
            } else {
                const int ow = owq;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int th = 0; th < 3; ++th) {
                    const int ih = oh * stride - pad + th * dil;
                    if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
                    for (int tw = 0; tw < 3; ++tw) {
                        const int iw = ow * stride - pad + tw * dil;
                        if ((unsigned)iw >= (unsigned)W) continue;
                        const float4 v = ld(x + (((int64_t)b * H + ih) * W + iw) * ldx + q * 4);
                        const float4 wk = ww[th * 3 + tw];
                        acc.x = fmaf(v.x, wk.x, acc.x); acc.y = fmaf(v.y, wk.y, acc.y);
                        acc.z = fmaf(v.z, wk.z, acc.z); acc.w = fmaf(v.w, wk.w, acc.w);
                    }
                }
                const int64_t ro = ((int64_t)b * Ho + oh) * Wo + ow;
                *reinterpret_cast<float4*>(y + ro * ldy + q * 4) = acc;
                s1.x += acc.x; s1.y += acc.y; s1.z += acc.z; s1.w += acc.w;
                s2.x = fmaf(acc.x, acc.x, s2.x); s2.y = fmaf(acc.y, acc.y, s2.y);
                s2.z = fmaf(acc.z, acc.z, s2.z); s2.w = fmaf(acc.w, acc.w, s2.w);
            }
        }
    }
    if (!stats) return;
    sh[0][t] = s1;
    sh[1][t] = s2;
    __syncthreads();
    if (ry == 0 && q < cq) {
        for (int k = 1; k < g.rows_per_pass; ++k) {            // fixed order over the row lanes
            const float4 a = sh[0][k * g.cq_blk + ql], b2 = sh[1][k * g.cq_blk + ql];
            s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
            s2.x += b2.x; s2.y += b2.y; s2.z += b2.z; s2.w += b2.w;
        }
        *reinterpret_cast<float4*>(stats + ((int64_t)blockIdx.x * 2 + 0) * C + q * 4) = s1;
        *reinterpret_cast<float4*>(stats + ((int64_t)blockIdx.x * 2 + 1) * C + q * 4) = s2;
    }
}

// dx(ih,iw) = sum_t dy((ih + pad - th*dil)/stride, ...) * w[t]  where divisible
__global__ __launch_bounds__(kT) void dwconv_bwd_data_kernel(const float* dy, int64_t lddy, int B, int Ho, int Wo, int cq,
                                                            const float* w, int stride, int pad, int dil, float* dx,
                                                            int64_t lddx, int H, int W)
{
    const int C = cq * 4;
    const int64_t total = (int64_t)B * H * W * cq;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
        int q, iw, ih, b;
        decode_bhwq(e, cq, W, H, q, iw, ih, b);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // nine UNCONDITIONAL loads (clamped pixel, zeroed where the tap is dead) instead of a branch in front of each
        float4 g[9];
#pragma unroll
        for (int th = 0; th < 3; ++th) {
            const int nh = ih + pad - th * dil;
            const int oh = nh / stride;
            const bool rok = nh >= 0 && oh * stride == nh && oh < Ho;
#pragma unroll
            for (int tw = 0; tw < 3; ++tw) {
                const int nw = iw + pad - tw * dil;
                const int ow = nw / stride;
                const bool ok = rok && nw >= 0 && ow * stride == nw && ow < Wo;
                const float4 t = *reinterpret_cast<const float4*>(dy + (((int64_t)b * Ho + (ok ? oh : 0)) * Wo + (ok ? ow : 0)) * lddy + q * 4);
                g[th * 3 + tw] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4 ww = *reinterpret_cast<const float4*>(w + t * C + q * 4);
            acc.x = fmaf(g[t].x, ww.x, acc.x); acc.y = fmaf(g[t].y, ww.y, acc.y);
            acc.z = fmaf(g[t].z, ww.z, acc.z); acc.w = fmaf(g[t].w, ww.w, acc.w);
        }
        *reinterpret_cast<float4*>(dx + (((int64_t)b * H + ih) * W + iw) * lddx + q * 4) = acc;
    }
}

// Stride 2, dilation 1 (the four strided depthwise layers of MobileNetV2): in the generic kernel above the lanes of a wave disagree
// about which taps are live (the parity of ih / iw) - every lane walks all nine taps' branches, each load behind a branch (41 us for
// the 51 MB dx of the first one).  Here a thread produces the 2 x 2 block of dx pixels with ih + pad in {2 ph, 2 ph + 1}, iw + pad in
// {2 pw, 2 pw + 1}: it needs the four dy pixels (ph - 1 .. ph) x (pw - 1 .. pw) and all nine weights, no branches around the loads,
// and adds each pixel's taps in the generic kernel's order (th, tw ascending): the results are bit-identical.
__global__ __launch_bounds__(kT) void dwconv_s2_bwd_data_kernel(const float* dy, int64_t lddy, int B, int Ho, int Wo, int cq, const float* w,
                                                               int pad, float* dx, int64_t lddx, int H, int W)
{
    const int C = cq * 4;
    const int PH = (H + pad + 1) / 2 + 1, PW = (W + pad + 1) / 2 + 1;       // blocks in the padded coordinate (covers ih + pad up to H + pad - 1)
    const int64_t total = (int64_t)B * PH * PW * cq;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
        int q, pw, ph, b;
        decode_bhwq(e, cq, PW, PH, q, pw, ph, b);
        float4 g[2][2];                                      // g[a][c] = dy(ph - a, pw - c), zero outside
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int oh = ph - a, ow = pw - c;
                const bool ok = (unsigned)oh < (unsigned)Ho && (unsigned)ow < (unsigned)Wo;
                const float4 v = *reinterpret_cast<const float4*>(dy + (ok ? (((int64_t)b * Ho + oh) * Wo + ow) * lddy : 0) + q * 4);
                g[a][c] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        float4 ww[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) ww[i] = *reinterpret_cast<const float4*>(w + i * C + q * 4);
        // dx pixel (ih, iw) = (2 ph + dr - pad, 2 pw + dc - pad); its live taps: th = dr + 2 a (a = 0, 1 while th <= 2), dy row ph - a
#pragma unroll
        for (int dr = 0; dr < 2; ++dr)
#pragma unroll
            for (int dc = 0; dc < 2; ++dc) {
                const int ih = 2 * ph + dr - pad, iw = 2 * pw + dc - pad;
                if ((unsigned)ih >= (unsigned)H || (unsigned)iw >= (unsigned)W) continue;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int th = dr + 2 * a;
                    if (th > 2) continue;
                    const bool rok = (unsigned)(ph - a) < (unsigned)Ho;
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const int tw = dc + 2 * c;
                        if (tw > 2) continue;
                        if (!rok || (unsigned)(pw - c) >= (unsigned)Wo) continue;       // the generic kernel skips these taps too
                        const float4 gv = g[a][c], wv = ww[th * 3 + tw];
                        acc.x = fmaf(gv.x, wv.x, acc.x); acc.y = fmaf(gv.y, wv.y, acc.y);
                        acc.z = fmaf(gv.z, wv.z, acc.z); acc.w = fmaf(gv.w, wv.w, acc.w);
                    }
                }
                *reinterpret_cast<float4*>(dx + (((int64_t)b * H + ih) * W + iw) * lddx + q * 4) = acc;
            }
    }
}

// dw[t][c] = sum_{b,oh,ow} x[..]*dy[..]: per row-block partials [nblk][9][C], then fixed-order finalize
__global__ __launch_bounds__(kT) void dwconv_bwd_weight_kernel(const float* x, int64_t ldx, int B, int H, int W, int cq,
                                                              const float* dy, int64_t lddy, int Ho, int Wo, int stride,
                                                              int pad, int dil, ColReduceGeom g, float* part,
                                                              const float* in_scale, const float* in_shift, int in_act)
{
    __shared__ float4 sh[kT];
    const int C = cq * 4;
    const int t = threadIdx.x;
    const int ql = t % g.cq_blk, ry = t / g.cq_blk;
    const int q = blockIdx.y * g.cq_blk + ql;
    const bool active = ry < g.rows_per_pass && q < cq;
    const int64_t M = (int64_t)B * Ho * Wo;
    float4 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
        const bool aff = in_scale != nullptr;
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sf = make_float4(0.f, 0.f, 0.f, 0.f);
        if (aff) { sc = *reinterpret_cast<const float4*>(in_scale + q * 4); sf = *reinterpret_cast<const float4*>(in_shift + q * 4); }
        const int64_t r0 = (int64_t)blockIdx.x * g.rows_per_block;
        const int64_t r1 = r0 + g.rows_per_block < M ? r0 + g.rows_per_block : M;
        for (int64_t r = r0 + ry; r < r1; r += g.rows_per_pass) {
            const unsigned ru = (unsigned)r;                 // M < 2^31 (checked on the host)
            const unsigned tt = ru / (unsigned)Wo;
            const int ow = (int)(ru - tt * (unsigned)Wo);
            const int b = (int)(tt / (unsigned)Ho);
            const int oh = (int)(tt - (unsigned)b * (unsigned)Ho);
            const float4 gg = *reinterpret_cast<const float4*>(dy + r * lddy + q * 4);
#pragma unroll
            for (int th = 0; th < 3; ++th) {
                const int ih = oh * stride - pad + th * dil;
                if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
                for (int tw = 0; tw < 3; ++tw) {
                    const int iw = ow * stride - pad + tw * dil;
                    if ((unsigned)iw >= (unsigned)W) continue;
                    float4 v = *reinterpret_cast<const float4*>(x + (((int64_t)b * H + ih) * W + iw) * ldx + q * 4);
                    if (aff) {               // x is act(bn(raw)) of the producer, applied on load (see dwconv_fwd_fused_kernel)
                        v.x = act_fwd(fmaf(v.x, sc.x, sf.x), in_act); v.y = act_fwd(fmaf(v.y, sc.y, sf.y), in_act);
                        v.z = act_fwd(fmaf(v.z, sc.z, sf.z), in_act); v.w = act_fwd(fmaf(v.w, sc.w, sf.w), in_act);
                    }
                    float4& a = acc[th * 3 + tw];
                    a.x = fmaf(v.x, gg.x, a.x); a.y = fmaf(v.y, gg.y, a.y);
                    a.z = fmaf(v.z, gg.z, a.z); a.w = fmaf(v.w, gg.w, a.w);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        sh[t] = acc[k];
        __syncthreads();
        if (ry == 0 && q < cq) {
            float4 s = acc[k];
            for (int j = 1; j < g.rows_per_pass; ++j) {
                const float4 a = sh[j * g.cq_blk + ql];
                s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            }
            *reinterpret_cast<float4*>(part + ((int64_t)blockIdx.x * 9 + k) * C + q * 4) = s;
        }
        __syncthreads();
    }
}

// Stride-1, dilation-1 variant: one work item = FOUR consecutive output pixels of one image row (Wo % 4 == 0) x one
// channel quad.  The four 3x3 windows overlap in a 3 x 6 input patch, so an item costs 18 + 4 float4 loads instead of 40,
// all in flight at once; fewer, fatter row blocks then keep the partial traffic ([blocks][9][C]) below the input size
// (the one-pixel kernel needed 512 row blocks on a 2048-pixel map to hide its latency: 17.7 MB of partials for 7.8 MB of x).
__global__ __launch_bounds__(kT) void dwconv_bwd_weight_x4_kernel(const float* x, int64_t ldx, int B, int H, int W, int cq,
                                                                 const float* dy, int64_t lddy, int Ho, int Wo, int pad,
                                                                 ColReduceGeom g, float* part,
                                                                 const float* in_scale, const float* in_shift, int in_act)
{
    __shared__ float4 sh[kT];
    const int C = cq * 4;
    const int t = threadIdx.x;
    const int ql = t % g.cq_blk, ry = t / g.cq_blk;
    const int q = blockIdx.y * g.cq_blk + ql;
    const bool active = ry < g.rows_per_pass && q < cq;
    const int wq = Wo / 4;
    const int64_t M4 = (int64_t)B * Ho * wq;            // work items
    float4 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
        const bool aff = in_scale != nullptr;
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sf = make_float4(0.f, 0.f, 0.f, 0.f);
        if (aff) { sc = *reinterpret_cast<const float4*>(in_scale + q * 4); sf = *reinterpret_cast<const float4*>(in_shift + q * 4); }
        const int64_t r0 = (int64_t)blockIdx.x * g.rows_per_block;
        const int64_t r1 = r0 + g.rows_per_block < M4 ? r0 + g.rows_per_block : M4;
        for (int64_t r = r0 + ry; r < r1; r += g.rows_per_pass) {
            const unsigned ru = (unsigned)r;
            const unsigned tt = ru / (unsigned)wq;
            const int ow0 = (int)(ru - tt * (unsigned)wq) * 4;
            const int b = (int)(tt / (unsigned)Ho);
            const int oh = (int)(tt - (unsigned)b * (unsigned)Ho);
            const float* gp = dy + (((int64_t)b * Ho + oh) * Wo + ow0) * lddy + q * 4;
            float4 gg[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) gg[o] = *reinterpret_cast<const float4*>(gp + (int64_t)o * lddy);
#pragma unroll
            for (int th = 0; th < 3; ++th) {
                const int ih = oh - pad + th;
                if ((unsigned)ih >= (unsigned)H) continue;
                const float* row = x + ((int64_t)b * H + ih) * W * ldx + q * 4;
                float4 v[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int iw = ow0 - pad + j;
                    v[j] = (unsigned)iw < (unsigned)W ? *reinterpret_cast<const float4*>(row + (int64_t)iw * ldx)
                                                      : make_float4(0.f, 0.f, 0.f, 0.f);
                    if (aff && (unsigned)iw < (unsigned)W) {
                        v[j].x = act_fwd(fmaf(v[j].x, sc.x, sf.x), in_act); v[j].y = act_fwd(fmaf(v[j].y, sc.y, sf.y), in_act);
                        v[j].z = act_fwd(fmaf(v[j].z, sc.z, sf.z), in_act); v[j].w = act_fwd(fmaf(v[j].w, sc.w, sf.w), in_act);
                    }
                }
#pragma unroll
                for (int tw = 0; tw < 3; ++tw) {
                    float4& a = acc[th * 3 + tw];
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        a.x = fmaf(v[o + tw].x, gg[o].x, a.x); a.y = fmaf(v[o + tw].y, gg[o].y, a.y);
                        a.z = fmaf(v[o + tw].z, gg[o].z, a.z); a.w = fmaf(v[o + tw].w, gg[o].w, a.w);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        sh[t] = acc[k];
        __syncthreads();
        if (ry == 0 && q < cq) {
            float4 s = acc[k];
            for (int j = 1; j < g.rows_per_pass; ++j) {
                const float4 a = sh[j * g.cq_blk + ql];
                s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            }
            *reinterpret_cast<float4*>(part + ((int64_t)blockIdx.x * 9 + k) * C + q * 4) = s;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(kT) void sum_partials_kernel(const float* part, int nblk, int64_t n, float* out, float mul)
{
    __shared__ double shd[kT];
    const int64_t i = (int64_t)blockIdx.x * 8 + (threadIdx.x & 7);
    const double s = lanes32_sum(part, nblk, n, i, i < n, shd);
    if (i >= n || threadIdx.x >= 8) return;
    out[i] = (float)(s * (double)mul);
}

// ================================================================================================
// zero padding / cropping (mobilenet_v2.py:15-21 fixed_padding and its adjoint)
// ================================================================================================
__global__ __launch_bounds__(kT) void pad_kernel(const float* x, int64_t ldx, int B, int H, int W, int cq, int pt, int pl,
                                                float* y, int64_t ldy, int Hp, int Wp)
{
    const int64_t total = (int64_t)B * Hp * Wp * cq;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
        const int q = (int)(e % cq);
        int64_t t = e / cq;
        const int pw = (int)(t % Wp); t /= Wp;
        const int ph = (int)(t % Hp);
        const int b = (int)(t / Hp);
        const int ih = ph - pt, iw = pw - pl;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
            v = *reinterpret_cast<const float4*>(x + (((int64_t)b * H + ih) * W + iw) * ldx + q * 4);
        *reinterpret_cast<float4*>(y + (((int64_t)b * Hp + ph) * Wp + pw) * ldy + q * 4) = v;
    }
}

// y = crop(x_padded) [+ add]
__global__ __launch_bounds__(kT) void crop_kernel(const float* xp, int64_t ldxp, int B, int Hp, int Wp, int cq, int pt, int pl,
                                                 const float* add, int64_t ldadd, float* y, int64_t ldy, int H, int W)
{
    const int64_t total = (int64_t)B * H * W * cq;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
        const int q = (int)(e % cq);
        int64_t t = e / cq;
        const int iw = (int)(t % W); t /= W;
        const int ih = (int)(t % H);
        const int b = (int)(t / H);
        float4 v = *reinterpret_cast<const float4*>(xp + (((int64_t)b * Hp + ih + pt) * Wp + iw + pl) * ldxp + q * 4);
        const int64_t r = ((int64_t)b * H + ih) * W + iw;
        if (add) {
            const float4 a = *reinterpret_cast<const float4*>(add + r * ldadd + q * 4);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        *reinterpret_cast<float4*>(y + r * ldy + q * 4) = v;
    }
}

// ================================================================================================
// bilinear interpolation, torch semantics (upsample_bilinear2d): source index computed in fp32 as
//   align_corners:  src = scale*dst,               scale = (in-1)/(out-1)   (0 if out == 1)
//   otherwise:      src = max(scale*(dst+0.5)-0.5, 0),  scale = in/out (or 1/scale_factor)
// ================================================================================================
// (Lerp / lerp_src / bilerp live in pp_common.h: the fused low-resolution acquisition kernel shares them bit for bit)

// NHWC -> NHWC (channel slices allowed) or NHWC -> NCHW (out_nchw: y is [B,C,Ho,Wo] contiguous)
template <bool OUT_NCHW>
__global__ __launch_bounds__(kT) void bilinear_fwd_kernel(const float* x, int64_t ldx, int B, int H, int W, int C,
                                                         float* y, int64_t ldy, int Ho, int Wo, float sh, float sw,
                                                         int align)
{
    if constexpr (!OUT_NCHW) {
        const int cq = C / 4;
        const int64_t total = (int64_t)B * Ho * Wo * cq;
        for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
            int q, ow, oh, b;
            if (total < (1ll << 31)) {                  // 32-bit index arithmetic (three 64-bit divisions per float4 cost more than the loads)
                const unsigned eu = (unsigned)e, t1 = eu / (unsigned)cq, t2 = t1 / (unsigned)Wo;
                q = (int)(eu - t1 * (unsigned)cq);
                ow = (int)(t1 - t2 * (unsigned)Wo);
                b = (int)(t2 / (unsigned)Ho);
                oh = (int)(t2 - (unsigned)b * (unsigned)Ho);
            } else {
                q = (int)(e % cq);
                int64_t t = e / cq;
                ow = (int)(t % Wo); t /= Wo;
                oh = (int)(t % Ho);
                b = (int)(t / Ho);
            }
            const Lerp lh = lerp_src(oh, H, sh, align), lw = lerp_src(ow, W, sw, align);
            const float* base = x + (int64_t)b * H * W * ldx + q * 4;
            const float4 v00 = *reinterpret_cast<const float4*>(base + ((int64_t)lh.i0 * W + lw.i0) * ldx);
            const float4 v01 = *reinterpret_cast<const float4*>(base + ((int64_t)lh.i0 * W + lw.i1) * ldx);
            const float4 v10 = *reinterpret_cast<const float4*>(base + ((int64_t)lh.i1 * W + lw.i0) * ldx);
            const float4 v11 = *reinterpret_cast<const float4*>(base + ((int64_t)lh.i1 * W + lw.i1) * ldx);
            float4 o;
            o.x = bilerp(lh.l0, lh.l1, lw.l0, lw.l1, v00.x, v01.x, v10.x, v11.x);
            o.y = bilerp(lh.l0, lh.l1, lw.l0, lw.l1, v00.y, v01.y, v10.y, v11.y);
            o.z = bilerp(lh.l0, lh.l1, lw.l0, lw.l1, v00.z, v01.z, v10.z, v11.z);
            o.w = bilerp(lh.l0, lh.l1, lw.l0, lw.l1, v00.w, v01.w, v10.w, v11.w);
            *reinterpret_cast<float4*>(y + (((int64_t)b * Ho + oh) * Wo + ow) * ldy + q * 4) = o;
        }
    } else {
        // one thread per output pixel (b,c,oh,ow): consecutive threads walk ow -> coalesced NCHW stores;
        // the 4 source pixels are shared by ~scale^2 neighbours (L1/L2 hits)
        const int64_t total = (int64_t)B * C * Ho * Wo;
        for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
            const int ow = (int)(e % Wo);
            int64_t t = e / Wo;
            const int oh = (int)(t % Ho); t /= Ho;
            const int c = (int)(t % C);
            const int b = (int)(t / C);
            const Lerp lh = lerp_src(oh, H, sh, align), lw = lerp_src(ow, W, sw, align);
            const float* base = x + (int64_t)b * H * W * ldx + c;
            const float v00 = base[((int64_t)lh.i0 * W + lw.i0) * ldx], v01 = base[((int64_t)lh.i0 * W + lw.i1) * ldx];
            const float v10 = base[((int64_t)lh.i1 * W + lw.i0) * ldx], v11 = base[((int64_t)lh.i1 * W + lw.i1) * ldx];
            y[e] = bilerp(lh.l0, lh.l1, lw.l0, lw.l1, v00, v01, v10, v11);
        }
    }
}

// Backward as a GATHER (deterministic, no atomics): input pixel (ih,iw) collects from every output pixel
// whose i0 or i1 equals it.  Candidate output rows: a conservative window around ih/scale.
__device__ __forceinline__ void out_window(int i, int in, int out, float scale, int align, int& lo, int& hi)
{
    // outputs o with i0(o) in {i-1, i}:  src in [i-1, i+1)
    float inv = scale > 0.0f ? 1.0f / scale : 0.0f;
    float a = ((float)i - 1.0f) * inv, b = ((float)i + 1.0f) * inv;
    if (!align) { a -= 0.5f; b += 0.5f; }
    lo = (int)floorf(a) - 1;
    hi = (int)ceilf(b) + 1;
    if (lo < 0) lo = 0;
    if (hi > out - 1) hi = out - 1;
    if (scale == 0.0f) { lo = 0; hi = out - 1; }
}

// NHWC, C % 4 == 0: one thread per (input pixel, 4 channels); the row / column weights of the candidate window are
// separable, so they are evaluated once per thread (<= 2*kMaxWin lerp evaluations) instead of once per tap.
constexpr int kMaxWin = 24;
__global__ __launch_bounds__(kT) void bilinear_bwd4_kernel(const float* dy, int64_t lddy, int B, int Ho, int Wo, int cq,
                                                          float* dx, int64_t lddx, int H, int W, float sh, float sw, int align)
{
    const int64_t total = (int64_t)B * H * W * cq;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
        const int q = (int)(e % cq);
        int64_t t = e / cq;
        const int iw = (int)(t % W); t /= W;
        const int ih = (int)(t % H);
        const int b = (int)(t / H);
        int h0, h1, w0, w1;
        out_window(ih, H, Ho, sh, align, h0, h1);
        out_window(iw, W, Wo, sw, align, w0, w1);
        float ww[kMaxWin];               // statically indexed everywhere (fully unrolled loops): stays in registers
        const int nw = w1 - w0 + 1;
#pragma unroll
        for (int k = 0; k < kMaxWin; ++k) {
            float v = 0.0f;
            if (k < nw) {
                const Lerp lw = lerp_src(w0 + k, W, sw, align);
                if (lw.i0 == iw) v += lw.l0;
                if (lw.i1 == iw) v += lw.l1;
            }
            ww[k] = v;
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int oh = h0; oh <= h1; ++oh) {
            const Lerp lh = lerp_src(oh, H, sh, align);
            float wh = 0.0f;
            if (lh.i0 == ih) wh += lh.l0;
            if (lh.i1 == ih) wh += lh.l1;
            if (wh == 0.0f) continue;
            float4 row = make_float4(0.f, 0.f, 0.f, 0.f);
            const float* base = dy + (((int64_t)b * Ho + oh) * Wo + w0) * lddy + q * 4;
#pragma unroll
            for (int k = 0; k < kMaxWin; ++k) {
                const float wk = ww[k];
                if (wk == 0.0f) continue;    // also covers k >= nw
                const float4 g = *reinterpret_cast<const float4*>(base + (int64_t)k * lddy);
                row.x = fmaf(wk, g.x, row.x); row.y = fmaf(wk, g.y, row.y); row.z = fmaf(wk, g.z, row.z); row.w = fmaf(wk, g.w, row.w);
            }
            acc.x = fmaf(wh, row.x, acc.x); acc.y = fmaf(wh, row.y, acc.y); acc.z = fmaf(wh, row.z, acc.z); acc.w = fmaf(wh, row.w, acc.w);
        }
        *reinterpret_cast<float4*>(dx + (((int64_t)b * H + ih) * W + iw) * lddx + q * 4) = acc;
    }
}

// Same gather with the window ROWS spread over the block: block = (input pixel, group of QB float4 columns),
// thread (r, ql) accumulates window rows h0+r, h0+r+NR, ..; the NR row sums are combined in fixed order through LDS.
// The one-thread-per-pixel form above walks ~49 taps serially with only B*H*W*C/4 threads (131 K for the ASPP
// upsample: 144 us for 33 MB); this form has ~8x the loads in flight.
__global__ __launch_bounds__(kT) void bilinear_bwd4r_kernel(const float* dy, int64_t lddy, int B, int Ho, int Wo, int cq,
                                                           float* dx, int64_t lddx, int H, int W, float sh, float sw,
                                                           int align, int QB, int NR)
{
    __shared__ float4 red[kT];
    const int t = threadIdx.x;
    const int ql = t % QB, r = t / QB;
    const int nqb = (cq + QB - 1) / QB;
    const int qb = blockIdx.x % nqb;
    int64_t pix = blockIdx.x / nqb;
    const int iw = (int)(pix % W); pix /= W;
    const int ih = (int)(pix % H);
    const int b = (int)(pix / H);
    const int q = qb * QB + ql;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < NR && q < cq) {
        int h0, h1, w0, w1;
        out_window(ih, H, Ho, sh, align, h0, h1);
        out_window(iw, W, Wo, sw, align, w0, w1);
        float ww[kMaxWin];
        const int nw = w1 - w0 + 1;
#pragma unroll
        for (int k = 0; k < kMaxWin; ++k) {
            float v = 0.0f;
            if (k < nw) {
                const Lerp lw = lerp_src(w0 + k, W, sw, align);
                if (lw.i0 == iw) v += lw.l0;
                if (lw.i1 == iw) v += lw.l1;
            }
            ww[k] = v;
        }
        for (int oh = h0 + r; oh <= h1; oh += NR) {
            const Lerp lh = lerp_src(oh, H, sh, align);
            float wh = 0.0f;
            if (lh.i0 == ih) wh += lh.l0;
            if (lh.i1 == ih) wh += lh.l1;
            if (wh == 0.0f) continue;
            float4 row = make_float4(0.f, 0.f, 0.f, 0.f);
            const float* base = dy + (((int64_t)b * Ho + oh) * Wo + w0) * lddy + q * 4;
#pragma unroll
            for (int k = 0; k < kMaxWin; ++k) {
                const float wk = ww[k];
                if (wk == 0.0f) continue;
                const float4 g = *reinterpret_cast<const float4*>(base + (int64_t)k * lddy);
                row.x = fmaf(wk, g.x, row.x); row.y = fmaf(wk, g.y, row.y); row.z = fmaf(wk, g.z, row.z); row.w = fmaf(wk, g.w, row.w);
            }
            acc.x = fmaf(wh, row.x, acc.x); acc.y = fmaf(wh, row.y, acc.y); acc.z = fmaf(wh, row.z, acc.z); acc.w = fmaf(wh, row.w, acc.w);
        }
    }
    red[t] = acc;
    __syncthreads();
    if (r == 0 && q < cq) {
        for (int k = 1; k < NR; ++k) {
            const float4 a = red[k * QB + ql];
            acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
        }
        *reinterpret_cast<float4*>(dx + (((int64_t)b * H + ih) * W + iw) * lddx + q * 4) = acc;
    }
}

// Exact x2, align_corners=False (every upsample of the FPN decoder, decoders.py:82,101): the gather has the closed
// form  dx[i] = 0.25 dy[2i-1] + 0.75 dy[2i] + 0.75 dy[2i+1] + 0.25 dy[2i+2]  per axis, with the weight of a tap that
// falls outside folded into its clamped neighbour (dy[0] and dy[2H-1] count 1.0).  16 float4 loads per thread, no
// window search (the generic kernel spends most of its issue slots skipping empty taps: 246 us for 268 MB).
__global__ __launch_bounds__(kT) void bilinear_up2_bwd4_kernel(const float* dy, int64_t lddy, int B, int cq, float* dx,
                                                              int64_t lddx, int H, int W)
{
    const int Ho = 2 * H, Wo = 2 * W;
    const int64_t total = (int64_t)B * H * W * cq;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
        const int q = (int)(e % cq);
        int64_t t = e / cq;
        const int iw = (int)(t % W); t /= W;
        const int ih = (int)(t % H);
        const int b = (int)(t / H);
        // taps o = 2i-1 .. 2i+2 and their weights for input index i (size n): out-of-range taps get weight 0,
        // the first / last output row counts fully
        float wh[4], ww[4];
        int oh[4], ow[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int o = 2 * ih - 1 + k;
            oh[k] = o < 0 ? 0 : (o > Ho - 1 ? Ho - 1 : o);
            float v = (k == 0 || k == 3) ? 0.25f : 0.75f;
            if (o < 0 || o > Ho - 1) v = 0.0f;
            if ((k == 1 && ih == 0) || (k == 2 && ih == H - 1)) v = 1.0f;
            wh[k] = v;
            const int p = 2 * iw - 1 + k;
            ow[k] = p < 0 ? 0 : (p > Wo - 1 ? Wo - 1 : p);
            float u = (k == 0 || k == 3) ? 0.25f : 0.75f;
            if (p < 0 || p > Wo - 1) u = 0.0f;
            if ((k == 1 && iw == 0) || (k == 2 && iw == W - 1)) u = 1.0f;
            ww[k] = u;
        }
        const float* base = dy + (int64_t)b * Ho * Wo * lddy + q * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float4 row = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 g = *reinterpret_cast<const float4*>(base + ((int64_t)oh[r] * Wo + ow[k]) * lddy);
                row.x = fmaf(ww[k], g.x, row.x); row.y = fmaf(ww[k], g.y, row.y);
                row.z = fmaf(ww[k], g.z, row.z); row.w = fmaf(ww[k], g.w, row.w);
            }
            acc.x = fmaf(wh[r], row.x, acc.x); acc.y = fmaf(wh[r], row.y, acc.y);
            acc.z = fmaf(wh[r], row.z, acc.z); acc.w = fmaf(wh[r], row.w, acc.w);
        }
        *reinterpret_cast<float4*>(dx + (((int64_t)b * H + ih) * W + iw) * lddx + q * 4) = acc;
    }
}

// dY in NCHW (the logits gradient, deeplab.py:55 / model.py:116) -> dX NHWC: one thread per (b, c, ih, iw) with iw
// fastest, so neighbouring lanes read neighbouring windows of the SAME dY plane (the per-element form below walks
// c fastest: every lane in a different plane, 4-byte accesses 512 KB apart).
__global__ __launch_bounds__(kT) void bilinear_bwd_planes_kernel(const float* dy, int B, int Ho, int Wo, int C, float* dx,
                                                                int64_t lddx, int H, int W, float sh, float sw, int align)
{
    const int64_t total = (int64_t)B * C * H * W;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
        const int iw = (int)(e % W);
        int64_t t = e / W;
        const int ih = (int)(t % H); t /= H;
        const int c = (int)(t % C);
        const int b = (int)(t / C);
        int h0, h1, w0, w1;
        out_window(ih, H, Ho, sh, align, h0, h1);
        out_window(iw, W, Wo, sw, align, w0, w1);
        float ww[kMaxWin];
        const int nw = w1 - w0 + 1;
#pragma unroll
        for (int k = 0; k < kMaxWin; ++k) {
            float v = 0.0f;
            if (k < nw) {
                const Lerp lw = lerp_src(w0 + k, W, sw, align);
                if (lw.i0 == iw) v += lw.l0;
                if (lw.i1 == iw) v += lw.l1;
            }
            ww[k] = v;
        }
        const float* plane = dy + ((int64_t)b * C + c) * Ho * Wo;
        float acc = 0.0f;
        for (int oh = h0; oh <= h1; ++oh) {
            const Lerp lh = lerp_src(oh, H, sh, align);
            float wh = 0.0f;
            if (lh.i0 == ih) wh += lh.l0;
            if (lh.i1 == ih) wh += lh.l1;
            if (wh == 0.0f) continue;
            const float* base = plane + (int64_t)oh * Wo + w0;
            float row = 0.0f;
#pragma unroll
            for (int k = 0; k < kMaxWin; ++k) {
                const float wk = ww[k];
                if (wk == 0.0f) continue;
                row = fmaf(wk, base[k], row);
            }
            acc = fmaf(wh, row, acc);
        }
        dx[(((int64_t)b * H + ih) * W + iw) * lddx + c] = acc;
    }
}

template <bool DY_NCHW>
__global__ __launch_bounds__(kT) void bilinear_bwd_kernel(const float* dy, int64_t lddy, int B, int Ho, int Wo, int C,
                                                         float* dx, int64_t lddx, int H, int W, float sh, float sw,
                                                         int align)
{
    const int64_t total = (int64_t)B * H * W * C;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
        const int c = (int)(e % C);
        int64_t t = e / C;
        const int iw = (int)(t % W); t /= W;
        const int ih = (int)(t % H);
        const int b = (int)(t / H);
        int h0, h1, w0, w1;
        out_window(ih, H, Ho, sh, align, h0, h1);
        out_window(iw, W, Wo, sw, align, w0, w1);
        float acc = 0.0f;
        for (int oh = h0; oh <= h1; ++oh) {
            const Lerp lh = lerp_src(oh, H, sh, align);
            float wh = 0.0f;
            if (lh.i0 == ih) wh += lh.l0;
            if (lh.i1 == ih) wh += lh.l1;
            if (wh == 0.0f) continue;
            float row = 0.0f;
            for (int ow = w0; ow <= w1; ++ow) {
                const Lerp lw = lerp_src(ow, W, sw, align);
                float ww = 0.0f;
                if (lw.i0 == iw) ww += lw.l0;
                if (lw.i1 == iw) ww += lw.l1;
                if (ww == 0.0f) continue;
                const float g = DY_NCHW ? dy[(((int64_t)b * C + c) * Ho + oh) * Wo + ow]
                                        : dy[(((int64_t)b * Ho + oh) * Wo + ow) * lddy + c];
                row = fmaf(ww, g, row);
            }
            acc = fmaf(wh, row, acc);
        }
        dx[(((int64_t)b * H + ih) * W + iw) * lddx + c] = acc;
    }
}

// ================================================================================================
// per-image spatial mean (AdaptiveAvgPool2d(1)) / column sum, and its adjoint (broadcast)
// ================================================================================================
// out[b][c] = mul * sum_p x[b][p][c]   ; one block per (image, 64-channel group), 4 row lanes
__global__ __launch_bounds__(kT) void image_colsum_kernel(const float* x, int64_t ldx, int64_t P, int C, float mul,
                                                         float* out, int64_t ldo)
{
    __shared__ float sh[4][64];
    const int b = blockIdx.y;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ry = threadIdx.x >> 6;
    float s = 0.0f;
    if (c < C) {
        const float* base = x + (int64_t)b * P * ldx + c;
        int64_t p = ry;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (; p + 28 < P; p += 32) {          // 8 rows in flight per lane (one row per trip exposes a full load latency)
            const float v0 = base[p * ldx], v1 = base[(p + 4) * ldx], v2 = base[(p + 8) * ldx], v3 = base[(p + 12) * ldx];
            const float v4 = base[(p + 16) * ldx], v5 = base[(p + 20) * ldx], v6 = base[(p + 24) * ldx], v7 = base[(p + 28) * ldx];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3;
            s0 += v4; s1 += v5; s2 += v6; s3 += v7;
        }
        s = (s0 + s1) + (s2 + s3);
        for (; p < P; p += 4) s += base[p * ldx];
    }
    sh[ry][threadIdx.x & 63] = s;
    __syncthreads();
    if (ry == 0 && c < C) {
        const int l = threadIdx.x;
        out[(int64_t)b * ldo + c] = ((sh[0][l] + sh[1][l]) + (sh[2][l] + sh[3][l])) * mul;
    }
}

// y[b][p][c] = mul * v[b][c]
__global__ __launch_bounds__(kT) void image_broadcast_kernel(const float* v, int64_t ldv, int B, int64_t P, int cq, float mul,
                                                            float* y, int64_t ldy)
{
    const int64_t total = (int64_t)B * P * cq;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
        const int q = (int)(e % cq);
        const int64_t r = e / cq;
        const int b = (int)(r / P);
        float4 a = *reinterpret_cast<const float4*>(v + (int64_t)b * ldv + q * 4);
        a.x *= mul; a.y *= mul; a.z *= mul; a.w *= mul;
        *reinterpret_cast<float4*>(y + r * ldy + q * 4) = a;
    }
}

// ================================================================================================
// dropout: counter-based RNG (SplitMix64-style hash of (seed, flat element index)); the mask is regenerated
// in the backward pass from the same seed, never stored.
// ================================================================================================
// float4 variant (C % 4 == 0): same per-element hash stream as the scalar kernel (mask = f(seed, flat index))
__global__ __launch_bounds__(kT) void dropout4_kernel(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t M, int cq,
                                                     float p, float inv_keep, uint64_t seed, const uint64_t* seed_dev)
{
    if (seed_dev) seed += *seed_dev * 0x9E3779B97F4A7C15ull;   // per-step base seed read at run time (hipGraph replay)
    const int64_t total = M * cq;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
        const int64_t r = total <= 0xFFFFFFFFll ? (int64_t)((unsigned)e / (unsigned)cq) : e / cq;
        const int q = (int)(e - r * cq);
        const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + q * 4);
        const uint64_t i0 = (uint64_t)(r * cq + q) * 4;
        float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float u = (float)(hash_rng(seed, i0 + j) >> 8) * (1.0f / 16777216.0f);
            o[j] = u >= p ? o[j] * inv_keep : 0.0f;
        }
        *reinterpret_cast<float4*>(y + r * ldy + q * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// nn.Dropout2d (mobilenet_v2.py:114-115,127,133-134): ONE draw per (sample, channel) zeroes the whole feature map of that
// channel; same counter-based hash as the element dropout, indexed by b*C + c.
__global__ __launch_bounds__(kT) void dropout2d_kernel(const float* x, int64_t ldx, float* y, int64_t ldy, int B, int64_t P, int C,
                                                      float p, float inv_keep, uint64_t seed, const uint64_t* seed_dev)
{
    if (seed_dev) seed += *seed_dev * 0x9E3779B97F4A7C15ull;
    const int64_t total = (int64_t)B * P * C;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
        const int64_t r = e / C;                 // pixel row b*P + pix
        const int c = (int)(e - r * C);
        const int64_t b = r / P;
        const float u = (float)(hash_rng(seed, (uint64_t)(b * C + c)) >> 8) * (1.0f / 16777216.0f);
        y[r * ldy + c] = u >= p ? x[r * ldx + c] * inv_keep : 0.0f;
    }
}

__global__ __launch_bounds__(kT) void dropout_kernel(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t M, int C,
                                                    float p, float inv_keep, uint64_t seed, const uint64_t* seed_dev)
{
    if (seed_dev) seed += *seed_dev * 0x9E3779B97F4A7C15ull;
    const int64_t total = M * C;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
        const int64_t r = e / C;
        const int c = (int)(e - r * C);
        const float u = (float)(hash_rng(seed, (uint64_t)e) >> 8) * (1.0f / 16777216.0f);
        y[r * ldy + c] = u >= p ? x[r * ldx + c] * inv_keep : 0.0f;
    }
}

// ================================================================================================
// cross entropy with ignore_index on NCHW logits (model.py:116): loss = mean over labelled pixels of
// (logsumexp - x[target]); dlogits = (softmax - onehot)/N at labelled pixels, 0 elsewhere.
// ================================================================================================
__global__ __launch_bounds__(kT) void ce_partial_kernel(const float* logits, const int64_t* target, int B, int C, int64_t HW,
                                                       int64_t sB, int64_t sC, int ignore_index, float* part /*[nblk][2]*/)
{
    __shared__ float sh[2][kT];
    float ls = 0.0f, cnt = 0.0f;
    const int64_t total = (int64_t)B * HW;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
        const int64_t tg = target[e];
        if (tg == ignore_index) continue;
        const int64_t b = e / HW, pix = e - b * HW;
        const float* px = logits + b * sB + pix;
        float m = px[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, px[c * sC]);
        float S = 0.0f;
        for (int c = 0; c < C; ++c) S += expf(px[c * sC] - m);
        const float xt = (tg >= 0 && tg < C) ? px[tg * sC] : 0.0f;
        ls += (m + logf(S)) - xt;
        cnt += 1.0f;
    }
    sh[0][threadIdx.x] = ls;
    sh[1][threadIdx.x] = cnt;
    __syncthreads();
    for (int s = kT / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + s];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[blockIdx.x * 2] = sh[0][0]; part[blockIdx.x * 2 + 1] = sh[1][0]; }
}

// dlogits (NCHW contiguous) = grad_scale * (softmax - onehot) / N at labelled pixels, 0 elsewhere
__global__ __launch_bounds__(kT) void ce_bwd_kernel(const float* logits, const int64_t* target, int B, int C, int64_t HW,
                                                   int64_t sB, int64_t sC, int ignore_index, const float* count,
                                                   const float* grad_out, float* dlogits)
{
    const int64_t total = (int64_t)B * HW;
    const float gs = (grad_out ? *grad_out : 1.0f) / *count;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
        const int64_t tg = target[e];
        const int64_t b = e / HW, pix = e - b * HW;
        float* dst = dlogits + (b * C) * HW + pix;
        if (tg == ignore_index) {
            for (int c = 0; c < C; ++c) dst[c * HW] = 0.0f;
            continue;
        }
        const float* px = logits + b * sB + pix;
        float m = px[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, px[c * sC]);
        float S = 0.0f;
        for (int c = 0; c < C; ++c) S += expf(px[c * sC] - m);
        const float inv = 1.0f / S;
        for (int c = 0; c < C; ++c) {
            const float pr = expf(px[c * sC] - m) * inv;
            dst[c * HW] = gs * (pr - (c == tg ? 1.0f : 0.0f));
        }
    }
}

// ================================================================================================
// The same loss taken straight from the LOW-resolution classifier output (model.py:113-121 for DeepLab):
//   logits = F.interpolate(low, size=(H,W), 'bilinear', align_corners)   deeplab.py:55-56
//   loss   = F.cross_entropy(logits, target, ignore_index)               model.py:116
//   dlow   = d loss / d low                                              (autograd through both)
// PixelPick labels 10-100 pixels per image (80 of 524 288 at B=4), so neither the full-size logits
// (40 MB), nor their gradient (40 MB of zeros) need to exist: the forward scans the labels and interpolates the class
// vector only where a label is, the backward GATHERS per low-resolution pixel from the labelled pixels in its
// footprint (fixed order, no atomics: bitwise reproducible).
// ================================================================================================
template <int CMAX, bool EXACT>
__device__ __forceinline__ void lowres_class_vector(const float* base, int64_t ldx, int w, const Lerp& lh, const Lerp& lw, int C,
                                                    float (&x)[CMAX])
{
    const float* p00 = base + ((int64_t)lh.i0 * w + lw.i0) * ldx;
    const float* p01 = base + ((int64_t)lh.i0 * w + lw.i1) * ldx;
    const float* p10 = base + ((int64_t)lh.i1 * w + lw.i0) * ldx;
    const float* p11 = base + ((int64_t)lh.i1 * w + lw.i1) * ldx;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
        if (EXACT || c < C) x[c] = bilerp(lh.l0, lh.l1, lw.l0, lw.l1, p00[c], p01[c], p10[c], p11[c]);
}

template <int CMAX, bool EXACT>
__global__ __launch_bounds__(kT) void ce_lowres_partial_kernel(const float* low, int64_t ldx, int B, int C, int h, int w, int H,
                                                              int W, float sh, float sw, int align, const int64_t* target,
                                                              int ignore_index, float* part /*[nblk][2]*/)
{
    __shared__ float shm[2][kT];
    float ls = 0.0f, cnt = 0.0f;
    const int64_t HW = (int64_t)H * W, total = (int64_t)B * HW;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
        const int64_t tg = target[e];
        if (tg == ignore_index) continue;
        const int64_t b = e / HW, pix = e - b * HW;
        const int Y = (int)(pix / W), X = (int)(pix - (int64_t)Y * W);
        float x[CMAX];
        lowres_class_vector<CMAX, EXACT>(low + b * h * w * ldx, ldx, w, lerp_src(Y, h, sh, align), lerp_src(X, w, sw, align), C, x);
        float m = x[0], xt = 0.0f;
#pragma unroll
        for (int c = 1; c < CMAX; ++c)
            if (EXACT || c < C) m = fmaxf(m, x[c]);
        float S = 0.0f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (EXACT || c < C) {
                S += expf(x[c] - m);
                xt = (c == tg) ? x[c] : xt;
            }
        ls += (m + logf(S)) - xt;
        cnt += 1.0f;
    }
    shm[0][threadIdx.x] = ls;
    shm[1][threadIdx.x] = cnt;
    __syncthreads();
    for (int s = kT / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            shm[0][threadIdx.x] += shm[0][threadIdx.x + s];
            shm[1][threadIdx.x] += shm[1][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[blockIdx.x * 2] = shm[0][0]; part[blockIdx.x * 2 + 1] = shm[1][0]; }
}

// one 64-lane wave, fixed order: lane l adds part[l], part[l+64], ... in double, then a fixed shuffle tree
__global__ __launch_bounds__(64) void ce_finalize_wave_kernel(const float* part, int nblk, float* loss, float* count)
{
    double s = 0.0, n = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) { s += (double)part[b * 2]; n += (double)part[b * 2 + 1]; }
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, 64);
        n += __shfl_down(n, off, 64);
    }
    if (threadIdx.x == 0) {
        *count = (float)n;
        *loss = (float)(s / n);   // 0/0 = NaN when no pixel is labelled, like F.cross_entropy
    }
}

// One thread per low-resolution pixel: walk the output pixels that interpolate from it, and for the (rare) labelled
// ones add  weight * (softmax - onehot).  dlow [B,h,w,lddx] is fully written (zeros included).
template <int CMAX, bool EXACT>
__global__ __launch_bounds__(kT) void ce_lowres_bwd_kernel(const float* low, int64_t ldx, int B, int C, int h, int w, int H,
                                                          int W, float sh, float sw, int align, const int64_t* target,
                                                          int ignore_index, const float* count, const float* grad_out,
                                                          float* dlow, int64_t lddx)
{
    const int64_t total = (int64_t)B * h * w;
    const int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (e >= total) return;
    const int c0 = (int)(e % w);
    const int64_t t = e / w;
    const int r0 = (int)(t % h);
    const int b = (int)(t / h);
    const float gs = (grad_out ? *grad_out : 1.0f) / *count;
    float acc[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) acc[c] = 0.0f;
    int ylo, yhi, xlo, xhi;
    out_window(r0, h, H, sh, align, ylo, yhi);
    out_window(c0, w, W, sw, align, xlo, xhi);
    const float* base = low + (int64_t)b * h * w * ldx;
    const int64_t* tb = target + (int64_t)b * H * W;
    for (int Y = ylo; Y <= yhi; ++Y) {
        const Lerp lh = lerp_src(Y, h, sh, align);
        const float wh = (lh.i0 == r0 ? lh.l0 : 0.0f) + (lh.i1 == r0 ? lh.l1 : 0.0f);
        if (lh.i0 != r0 && lh.i1 != r0) continue;
        const int64_t* trow = tb + (int64_t)Y * W;
        for (int X = xlo; X <= xhi; ++X) {
            const int64_t tg = trow[X];
            if (tg == ignore_index) continue;
            const Lerp lw = lerp_src(X, w, sw, align);
            if (lw.i0 != c0 && lw.i1 != c0) continue;
            const float ww = (lw.i0 == c0 ? lw.l0 : 0.0f) + (lw.i1 == c0 ? lw.l1 : 0.0f);
            float x[CMAX];
            lowres_class_vector<CMAX, EXACT>(base, ldx, w, lh, lw, C, x);
            float m = x[0];
#pragma unroll
            for (int c = 1; c < CMAX; ++c)
                if (EXACT || c < C) m = fmaxf(m, x[c]);
            float S = 0.0f;
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (EXACT || c < C) {
                    x[c] = expf(x[c] - m);
                    S += x[c];
                }
            const float inv = 1.0f / S, wgt = wh * ww;
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (EXACT || c < C) acc[c] = fmaf(wgt, gs * (x[c] * inv - (c == tg ? 1.0f : 0.0f)), acc[c]);
        }
    }
    float* dst = dlow + e * lddx;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
        if (EXACT || c < C) dst[c] = acc[c];
}

// ================================================================================================
// Adam on flat buffers, torch.optim.Adam semantics (L2 weight decay, bias correction), two lr segments
// (utils/utils.py:125-141: backbone/encoder at lr/10).
// ================================================================================================
__global__ __launch_bounds__(kT) void adam_kernel(float* p, const float* g, float* m, float* v, int64_t n, int64_t n_split,
                                                 float lr_a, float lr_b, float beta1, float beta2, float eps, float wd,
                                                 float bc1, float bc2_sqrt, float grad_scale, const float* hyper_dev)
{
    if (hyper_dev) {   // [lr_a, lr_b, bc1, bc2_sqrt] read at run time: the launch can sit in a replayed hipGraph
        lr_a = hyper_dev[0]; lr_b = hyper_dev[1]; bc1 = hyper_dev[2]; bc2_sqrt = hyper_dev[3];
    }
    for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kT) {
        const float lr = i < n_split ? lr_a : lr_b;
        float grad = g[i] * grad_scale;
        const float par = p[i];
        grad = fmaf(wd, par, grad);
        const float mm = m[i] + (grad - m[i]) * (1.0f - beta1);
        const float vv = fmaf(grad * grad, 1.0f - beta2, v[i] * beta2);
        m[i] = mm;
        v[i] = vv;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        p[i] = par - (lr / bc1) * (mm / denom);
    }
}

// SGD with momentum on flat buffers, torch.optim.SGD semantics (L2 weight decay added to the gradient, dampening 0,
// no Nesterov; the momentum buffer of the FIRST step is the gradient itself), two lr segments
// (utils/utils.py:208-270: voc and the SGD variant use lr 1e-3 for the backbone/encoder and 1e-2 for the rest).
__global__ __launch_bounds__(kT) void sgd_kernel(float* p, const float* g, float* buf, int64_t n, int64_t n_split, float lr_a,
                                                float lr_b, float momentum, float wd, int first, float grad_scale,
                                                const float* hyper_dev)
{
    if (hyper_dev) { lr_a = hyper_dev[0]; lr_b = hyper_dev[1]; }
    for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kT) {
        const float par = p[i];
        float d = fmaf(wd, par, g[i] * grad_scale);
        if (momentum != 0.0f) {
            const float b = first ? d : fmaf(momentum, buf[i], d);
            buf[i] = b;
            d = b;
        }
        p[i] = par - (i < n_split ? lr_a : lr_b) * d;
    }
}

// NCHW -> NHWC (network input, 3 channels) ; generic small-C transpose
__global__ __launch_bounds__(kT) void nchw_to_nhwc_kernel(const float* x, int B, int C, int64_t HW, float* y, int64_t ldy)
{
    const int64_t total = (int64_t)B * HW * C;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
        const int c = (int)(e % C);
        const int64_t r = e / C;
        const int64_t b = r / HW, pix = r - b * HW;
        y[r * ldy + c] = x[(b * C + c) * HW + pix];
    }
}

// y = a + b  (gradient accumulation where a tensor has several consumers)
__global__ __launch_bounds__(kT) void add2d_kernel(const float* a, int64_t lda, const float* b, int64_t ldb, float* y,
                                                  int64_t ldy, int64_t M, int C)
{
    const int64_t total = M * C;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
        const int64_t r = e / C;
        const int c = (int)(e - r * C);
        y[r * ldy + c] = a[r * lda + c] + b[r * ldb + c];
    }
}

// float4 form (C, every ld multiples of 4, 16-byte aligned): the FPN top-down sums (decoders.py:36-55,95-99: 33.6 MB operands) ran
// at a fifth of the bandwidth in the scalar kernel (a 64-bit division and three 4-byte accesses per element: 86 us for 100 MB)
__global__ __launch_bounds__(kT) void add2d_v4_kernel(const float* a, int64_t lda, const float* b, int64_t ldb, float* y,
                                                     int64_t ldy, int64_t M, int cq, int flat)
{
    const int64_t total = M * cq;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
        int64_t oa = e * 4, ob = e * 4, oy = e * 4;
        if (!flat) {
            const int64_t r = total < (1ll << 31) ? (int64_t)((unsigned)e / (unsigned)cq) : e / cq;
            const int64_t c = (e - r * cq) * 4;
            oa = r * lda + c; ob = r * ldb + c; oy = r * ldy + c;
        }
        const float4 u = *reinterpret_cast<const float4*>(a + oa), v = *reinterpret_cast<const float4*>(b + ob);
        *reinterpret_cast<float4*>(y + oy) = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    }
}

// NHWC (pixel stride ldx) -> NCHW contiguous; consecutive threads walk the pixel axis (coalesced stores)
__global__ __launch_bounds__(kT) void nhwc_to_nchw_kernel(const float* x, int64_t ldx, int B, int C, int64_t HW, float* y)
{
    const int64_t total = (int64_t)B * C * HW;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
        const int64_t pix = e % HW;
        const int64_t t = e / HW;
        const int c = (int)(t % C);
        const int64_t b = t / C;
        y[e] = x[(b * HW + pix) * ldx + c];
    }
}

static int g_bil_sep = 1;   // separable bilinear backward for >= x3 up-sampling (bit 8 of pp_debug_set_dw_variant switches it off)
static int g_dw_wgrad_x4 = 1, g_dw_wgrad_x4_blocks = 256;   // four-pixel items for the stride-1 depthwise weight gradient
static int g_dw_wgrad_cq_blk = 0, g_dw_wgrad_passes = 0;      // column-block width / least rows per thread (pp_debug_set_dw_variant bits 13-17); 0 = by map size
static int g_dw_x4 = 1;     // pp_debug_set_dw_variant(1) switches the 4-outputs-per-thread depthwise kernels off (A/B)
static int g_dw_s2 = 1;     // bit 20: the 2 x 2-block backward-data kernel of the stride-2 layers off (A/B)

static inline unsigned grid_for(int64_t total)
{
    int64_t b = cdiv(total, kT);
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

static int need_c4(int C, const char* what)
{
    if (C < 4 || C % 4 != 0) return fail(PP_ERR_UNSUPPORTED, "%s: C=%d must be a positive multiple of 4", what, C);
    return PP_OK;
}

}  // namespace pp

using namespace pp;

extern "C" {

#ifdef PP_DEBUG_KNOBS
void pp_debug_set_dw_variant(int v)
{
    g_dw_x4 = (v & 1) ? 0 : 1;
    g_dw_s2 = (v & 1048576) ? 0 : 1;                        // bit 20: 2 x 2-block backward-data kernel of the stride-2 layers off (A/B)
    g_bil_sep = (v & 256) ? 0 : 1;
    g_dw_wgrad_x4 = (v & 512) ? 0 : 1;                      // bit 9: four-pixel depthwise weight-gradient kernel off
    { const int s4 = (v >> 10) & 7; g_dw_wgrad_x4_blocks = s4 == 1 ? 128 : s4 == 2 ? 512 : s4 == 3 ? 1024 : s4 == 4 ? 64 : 256; }
    { const int cb = (v >> 13) & 7; g_dw_wgrad_cq_blk = cb == 1 ? 8 : cb == 2 ? 16 : cb == 3 ? 32 : cb == 4 ? 64 : cb == 5 ? 128 : cb == 6 ? kT : 0; }
    { const int mp = (v >> 16) & 3; g_dw_wgrad_passes = mp == 1 ? 1 : mp == 2 ? 2 : mp == 3 ? 4 : 0; }
    const int sel = (v >> 1) & 7;                 // 0: default, 1: 512, 2: 256, 3: 128, 4: 2048 row blocks for the weight gradient
    g_dw_wgrad_blocks = sel == 1 ? 512 : sel == 2 ? 256 : sel == 3 ? 128 : sel == 4 ? 2048 : 1024;
}
#endif
#ifdef PP_DEBUG_KNOBS
void pp_debug_set_bn_target(int blocks) { g_bn_target_blocks = blocks > 0 ? (blocks > 1024 ? 1024 : blocks) : 0; }
#endif
#ifdef PP_DEBUG_KNOBS
void pp_debug_set_bn_bytes_per_block(int bytes)
{
    g_bn_bytes_per_block = bytes > 0 ? bytes : 0;
    g_bn_row_cache = bytes == -1 ? 0 : 1;          // -1: the register-cached variants off (A/B)
}
#endif
static thread_local unsigned long long* g_bn_probe = nullptr;
#ifdef PP_DEBUG_KNOBS
void pp_debug_set_bn_probe(void* device_buffer) { g_bn_probe = reinterpret_cast<unsigned long long*>(device_buffer); }
#endif
int pp_bn_fused_capacity(void) { return bn_fused_capacity(); }

// ---- batch norm -----------------------------------------------------------------------------------
size_t pp_colreduce_workspace_bytes(int64_t M, int C)
{
    if (M < 1 || C < 4) return 256;
    ColReduceGeom g = col_geom(M, C);
    size_t a = (size_t)g.nblk_rows * 2 * C * 4;
    size_t b = (size_t)(M < 2048 ? M : 2048) * 9 * C * 4;           // depthwise weight-gradient partials: no geometry has more than 2048 row blocks
    return align_up(a > b ? a : b, 256);
}

int pp_bn_train_fwd(const float* x, int64_t ldx, int64_t M, int C, const float* gamma, const float* beta, float eps,
                    float momentum, float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                    float* shift, void* workspace, size_t ws_bytes, pp_stream_t stream)
{
    if (!x || !gamma || !beta || !mean || !invstd || !scale || !shift) return fail(PP_ERR_BAD_ARG, "bn_train_fwd: null");
    if (int rc = need_c4(C, "bn_train_fwd")) return rc;
    if (M < 1 || ldx % 4 != 0) return fail(PP_ERR_BAD_ARG, "bn_train_fwd: bad M/ld");
    ColReduceGeom g = col_geom(M, C);
    if (!workspace || ws_bytes < (size_t)g.nblk_rows * 2 * C * 4) return fail(PP_ERR_WORKSPACE, "bn_train_fwd: workspace");
    hipStream_t st = as_stream(stream);
    float* part = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL((col_reduce_kernel<0>), dim3(g.nblk_rows, g.nblk_cols), dim3(kT), 0, st, x, (const float*)nullptr,
                       (const float*)nullptr, 0, (const float*)nullptr, (const float*)nullptr, M, C, ldx, (int64_t)0,
                       (int64_t)0, g, part);
    if (int rc = check_launch("col_reduce_kernel<0>")) return rc;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)cdiv(C, 8)), dim3(kT), 0, st, part, g.nblk_rows, C, (double)M,
                       gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift);
    return check_launch("bn_finalize_kernel");
}

size_t pp_bn_fused_workspace_bytes(int64_t M, int C)
{
    if (M < 1 || C < 4) return 256;
    BnFusedGeom g = bn_fused_geom(M, C);
    return align_up((size_t)g.nstrips * g.R * g.bq * 8 * 8, 256);     // one 64-bit {tag, value} word per partial
}

int pp_bn_fused_rows_cached(int64_t M, int C)
{
    if (M < 1 || C < 4 || C % 4) return 0;
    BnFusedGeom g = bn_fused_geom(M, C);
    return (g_bn_row_cache && g.rows_per_chunk <= (int64_t)g.nrl * kBnRowCache) ? 1 : 0;
}

size_t pp_bn_fused_sync_ints(int C) { (void)C; return 64; }      // [0] launch epoch, [1] blocks done; one 256-byte line

static int bn_fused_check(const char* what, int64_t M, int C, const BnFusedGeom& g, const void* workspace, size_t ws_bytes,
                          const int32_t* sync, size_t sync_ints)
{
    if (M < 1) return fail(PP_ERR_BAD_ARG, "%s: M < 1", what);
    if (!workspace || ws_bytes < (size_t)g.nstrips * g.R * g.bq * 8 * 8) return fail(PP_ERR_WORKSPACE, "%s: workspace", what);
    if ((reinterpret_cast<uintptr_t>(workspace) & 7) != 0) return fail(PP_ERR_BAD_ARG, "%s: workspace must be 8-byte aligned", what);
    if (!sync || sync_ints < 2) return fail(PP_ERR_WORKSPACE, "%s: sync array needs 2 ints", what);
    (void)C;
    return PP_OK;
}

int pp_bn_train_fwd_fused(const float* x, int64_t ldx, int64_t M, int C, const float* gamma, const float* beta, float eps,
                          float momentum, float* running_mean, float* running_var, float* mean, float* invstd,
                          const float* residual, int64_t ldr, int act, float drop_p, uint64_t drop_seed,
                          const uint64_t* drop_seed_dev, float* y, int64_t ldy, void* workspace,
                          size_t ws_bytes, int32_t* sync, size_t sync_ints, pp_stream_t stream)
{
    if (drop_p < 0.0f || drop_p >= 1.0f) return fail(PP_ERR_BAD_ARG, "bn_train_fwd_fused: dropout p=%f outside [0,1)", (double)drop_p);
    if (drop_p > 0.0f && act == 2) return fail(PP_ERR_UNSUPPORTED, "bn_train_fwd_fused: dropout after ReLU6 is not fusable");
    if (!x || !gamma || !beta || !mean || !invstd || !y) return fail(PP_ERR_BAD_ARG, "bn_train_fwd_fused: null");
    if (int rc = need_c4(C, "bn_train_fwd_fused")) return rc;
    if (ldx % 4 || ldy % 4 || (residual && ldr % 4)) return fail(PP_ERR_BAD_ARG, "bn_train_fwd_fused: ld must be multiples of 4");
    BnFusedGeom g = bn_fused_geom(M, C);
    if (int rc = bn_fused_check("bn_train_fwd_fused", M, C, g, workspace, ws_bytes, sync, sync_ints)) return rc;
    BnFwdArgs a{x, ldx, M, C, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, residual, ldr, act, y, ldy,
                reinterpret_cast<xword*>(workspace), sync, g, drop_p, 1.0f / (1.0f - drop_p), drop_seed, drop_seed_dev,
                nullptr, 0, nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, nullptr, 0, g_bn_probe};
    if (g_bn_row_cache && g.rows_per_chunk <= (int64_t)g.nrl * kBnRowCache)
        hipLaunchKernelGGL((bn_fused_fwd_kernel<0, kBnRowCache>), dim3((unsigned)(g.nstrips * g.R)), dim3(kT), 0, as_stream(stream), a);
    else
        hipLaunchKernelGGL((bn_fused_fwd_kernel<0, 0>), dim3((unsigned)(g.nstrips * g.R)), dim3(kT), 0, as_stream(stream), a);
    return check_launch("bn_fused_fwd_kernel");
}

int pp_bn_train_fwd_partials(const float* x, int64_t ldx, int64_t M, int C, const float* stats, int64_t stat_rows, const float* gamma,
                             const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* mean,
                             float* invstd, const float* residual, int64_t ldr, int act, float drop_p, uint64_t drop_seed,
                             const uint64_t* drop_seed_dev, float* y, int64_t ldy, pp_stream_t stream)
{
    if (drop_p < 0.0f || drop_p >= 1.0f) return fail(PP_ERR_BAD_ARG, "bn_train_fwd_partials: dropout p=%f outside [0,1)", (double)drop_p);
    if (drop_p > 0.0f && act == 2) return fail(PP_ERR_UNSUPPORTED, "bn_train_fwd_partials: dropout after ReLU6 is not fusable");
    if (!x || !gamma || !beta || !mean || !invstd || !y || !stats) return fail(PP_ERR_BAD_ARG, "bn_train_fwd_partials: null");
    if (M < 1 || stat_rows < 1 || stat_rows > 0x7FFFFFFFll) return fail(PP_ERR_BAD_ARG, "bn_train_fwd_partials: M / stat_rows");
    if (int rc = need_c4(C, "bn_train_fwd_partials")) return rc;
    if (ldx % 4 || ldy % 4 || (residual && ldr % 4)) return fail(PP_ERR_BAD_ARG, "bn_train_fwd_partials: ld must be multiples of 4");
    BnFusedGeom g = bn_fused_geom(M, C);
    BnFwdArgs a{x, ldx, M, C, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, residual, ldr, act, y, ldy,
                nullptr, nullptr, g, drop_p, 1.0f / (1.0f - drop_p), drop_seed, drop_seed_dev,
                nullptr, 0, nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, stats, (int)stat_rows};
    hipLaunchKernelGGL((bn_fused_fwd_kernel<2, 0>), dim3((unsigned)(g.nstrips * g.R)), dim3(kT), 0, as_stream(stream), a);
    return check_launch("bn_fused_fwd_kernel<partials>");
}

int pp_dwconv3x3_bn_train_fwd_fused(const float* in, int64_t ld_in, int B, int H, int W, int C, const float* w, int stride, int pad,
                                    int dil, float* x_out, int64_t ldx, const float* gamma, const float* beta, float eps,
                                    float momentum, float* running_mean, float* running_var, float* mean, float* invstd,
                                    const float* residual, int64_t ldr, int act, float* y, int64_t ldy, void* workspace,
                                    size_t ws_bytes, int32_t* sync, size_t sync_ints, pp_stream_t stream)
{
    if (!in || !w || !x_out || !gamma || !beta || !mean || !invstd || !y) return fail(PP_ERR_BAD_ARG, "dwconv_bn_train_fwd_fused: null");
    if (int rc = need_c4(C, "dwconv_bn_train_fwd_fused")) return rc;
    if (ld_in % 4 || ldx % 4 || ldy % 4 || (residual && ldr % 4)) return fail(PP_ERR_BAD_ARG, "dwconv_bn_train_fwd_fused: ld must be multiples of 4");
    if (stride < 1 || dil < 1) return fail(PP_ERR_BAD_ARG, "dwconv_bn_train_fwd_fused: stride / dilation");
    const int Ho = (H + 2 * pad - 2 * dil - 1) / stride + 1, Wo = (W + 2 * pad - 2 * dil - 1) / stride + 1;
    if (Ho < 1 || Wo < 1) return fail(PP_ERR_BAD_ARG, "dwconv_bn_train_fwd_fused: empty output");
    const int64_t M = (int64_t)B * Ho * Wo;
    if (M > 0x7FFFFFFFll) return fail(PP_ERR_UNSUPPORTED, "dwconv_bn_train_fwd_fused: more than 2^31 output pixels");
    BnFusedGeom g = bn_fused_geom(M, C);
    if (int rc = bn_fused_check("dwconv_bn_train_fwd_fused", M, C, g, workspace, ws_bytes, sync, sync_ints)) return rc;
    BnFwdArgs a{x_out, ldx, M, C, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, residual, ldr, act, y, ldy,
                reinterpret_cast<xword*>(workspace), sync, g, 0.0f, 1.0f, 0ull, nullptr,
                in, ld_in, w, x_out, H, W, Ho, Wo, stride, pad, dil, nullptr, 0};
    if (g_bn_row_cache && g.rows_per_chunk <= (int64_t)g.nrl * kBnRowCache)
        hipLaunchKernelGGL((bn_fused_fwd_kernel<1, kBnRowCache>), dim3((unsigned)(g.nstrips * g.R)), dim3(kT), 0, as_stream(stream), a);
    else
        hipLaunchKernelGGL((bn_fused_fwd_kernel<1, 0>), dim3((unsigned)(g.nstrips * g.R)), dim3(kT), 0, as_stream(stream), a);
    return check_launch("bn_fused_fwd_kernel<dw>");
}

static int bn_bwd_fused_impl(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* y_act, int64_t ldya, int act,
                             int64_t M, int C, const float* mean, const float* invstd, const float* gamma, float* dgamma,
                             float* dbeta, float* dx, int64_t lddx, float* dres, int64_t lddr, float grad_scale, const float* beta,
                             void* workspace, size_t ws_bytes, int32_t* sync, size_t sync_ints, pp_stream_t stream,
                             const unsigned char* row_flags);

int pp_bn_bwd_fused(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* y_act, int64_t ldya, int act,
                    int64_t M, int C, const float* mean, const float* invstd, const float* gamma, float* dgamma,
                    float* dbeta, float* dx, int64_t lddx, float* dres, int64_t lddr, float grad_scale, const float* beta,
                    void* workspace, size_t ws_bytes, int32_t* sync, size_t sync_ints, pp_stream_t stream)
{
    return bn_bwd_fused_impl(x, ldx, dy, lddy, y_act, ldya, act, M, C, mean, invstd, gamma, dgamma, dbeta, dx, lddx, dres, lddr, grad_scale,
                             beta, workspace, ws_bytes, sync, sync_ints, stream, nullptr);
}

int pp_bn_bwd_fused_sparse(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* y_act, int64_t ldya, int act,
                         int64_t M, int C, const float* mean, const float* invstd, const float* gamma, float* dgamma,
                         float* dbeta, float* dx, int64_t lddx, float* dres, int64_t lddr, float grad_scale, const float* beta,
                         void* workspace, size_t ws_bytes, int32_t* sync, size_t sync_ints, const unsigned char* row_flags,
                         pp_stream_t stream)
{
    return bn_bwd_fused_impl(x, ldx, dy, lddy, y_act, ldya, act, M, C, mean, invstd, gamma, dgamma, dbeta, dx, lddx, dres, lddr, grad_scale,
                             beta, workspace, ws_bytes, sync, sync_ints, stream, row_flags);
}

__global__ __launch_bounds__(256) void row_flags_kernel(const float* dy, int64_t ld, int64_t M, int C, unsigned char* flags)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= M) return;
    const float* p = dy + r * ld;
    bool nz = false;
    for (int c = 0; c < C; ++c) nz = nz || (p[c] != 0.0f);
    flags[r] = nz ? 1 : 0;                                    // exactly 0 / 1 (conv1x1_bwd_data_rows_kernel relies on it)
}

int pp_row_flags(const float* dy, int64_t lddy, int64_t M, int C, unsigned char* flags, pp_stream_t stream)
{
    if (!dy || !flags || M < 1 || C < 1) return fail(PP_ERR_BAD_ARG, "row_flags: bad argument");
    hipLaunchKernelGGL(row_flags_kernel, dim3((unsigned)cdiv(M, 256)), dim3(256), 0, as_stream(stream), dy, lddy, M, C, flags);
    return check_launch("row_flags_kernel");
}

// Backward-data of a pointwise convolution whose output gradient is zero outside the flagged rows (the classifier behind a sparsely
// labelled loss, decoders.py:120: Conv2d(256, n_classes, 1)): unflagged rows of dx are written as zeros, flagged rows are
// dx[r][c] = sum_k dy[r][k] W[c][k] (k ascending).  16 rows per block; a row's flag is block-uniform.
__global__ __launch_bounds__(256) void conv1x1_bwd_data_rows_kernel(const float* dy, int64_t lddy, int64_t M, int Cout, const float* w, int Cin,
                                                                     const unsigned char* flags, float* dx, int64_t lddx)
{
    __shared__ unsigned char fl[16];
    const int t = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 16;
    const int cq = Cin / 4;
    if (t < 16) fl[t] = (r0 + t < M) ? flags[r0 + t] : (unsigned char)2;       // 2: past the end (neither zeroed nor computed)
    __syncthreads();
    for (int e = t; e < 16 * cq; e += 256) {                 // zeros for the unflagged rows (float4, coalesced)
        const int rr = e / cq, q = e - rr * cq;
        if (fl[rr] == 0) *reinterpret_cast<float4*>(dx + (r0 + rr) * lddx + q * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int rr = 0; rr < 16; ++rr) {
        const int64_t r = r0 + rr;
        if (fl[rr] != 1) continue;
        const float* g = dy + r * lddy;
        for (int c = t; c < Cin; c += 256) {
            const float* wc = w + (int64_t)c * Cout;
            float acc = 0.0f;
            for (int k = 0; k < Cout; ++k) acc = fmaf(g[k], wc[k], acc);
            dx[r * lddx + c] = acc;
        }
    }
}

int pp_conv1x1_bwd_data_sparse(const float* dy, int64_t lddy, int64_t M, int Cout, const float* w, int Cin, const unsigned char* row_flags,
                             float* dx, int64_t lddx, pp_stream_t stream)
{
    if (!dy || !w || !row_flags || !dx || M < 1 || Cout < 1 || Cin < 1) return fail(PP_ERR_BAD_ARG, "conv1x1_bwd_data_rows: bad argument");
    if (Cin % 4 != 0 || lddx % 4 != 0 || (reinterpret_cast<uintptr_t>(dx) & 15) != 0)
        return fail(PP_ERR_BAD_ARG, "conv1x1_bwd_data_rows: Cin and lddx must be multiples of 4, dx 16-byte aligned");
    hipLaunchKernelGGL(conv1x1_bwd_data_rows_kernel, dim3((unsigned)cdiv(M, 16)), dim3(256), 0, as_stream(stream), dy, lddy, M, Cout, w, Cin,
                       row_flags, dx, lddx);
    return check_launch("conv1x1_bwd_data_rows_kernel");
}

// Weight (and bias) gradient of the same pointwise convolution: dw[c][k] = sum over the FLAGGED rows r, ascending, of x[r][c] dy[r][k]
// (db[k] = sum dy[r][k]) - 80 rows of 32768 (DeepLab's classifier) or of 524288 (FPNSeg's, at full resolution: the dense kernels read
// 310 MB for them, 412 us).  Fixed partition, fixed order: a block owns kSwRows consecutive rows, lists its flagged ones in row
// order (wave ballots), accumulates them one after the other - thread t owns elements t, t + 256, ... of the [Cin][Cout] (+ [Cout])
// result - and writes a slice of partial sums only if it saw any row; the reduce adds the non-empty slices in block order.
constexpr int kSwRows = 512, kSwAcc = 32;          // rows per block; accumulators per thread (Cin * Cout + Cout <= 8192)
__global__ __launch_bounds__(256) void conv1x1_wgrad_rows_partial_kernel(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t M, int Cin,
                                                                          int Cout, const unsigned char* flags, int with_bias, float* part, int* cnt)
{
    __shared__ int list[kSwRows];
    __shared__ int wcount[4], nlist;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * kSwRows;
    if (t == 0) nlist = 0;
    __syncthreads();
    for (int base = 0; base < kSwRows; base += 256) {              // ordered compaction: rows base .. base + 255 of this block
        const int64_t r = r0 + base + t;
        const bool f = r < M && flags[r] != 0;
        const unsigned long long bal = __ballot(f);
        if (lane == 0) wcount[wave] = (int)__popcll(bal);
        __syncthreads();
        int before = nlist;
        for (int w = 0; w < wave; ++w) before += wcount[w];
        if (f) list[before + (int)__popcll(bal & ((1ull << lane) - 1ull))] = base + t;
        __syncthreads();
        if (t == 0) nlist += wcount[0] + wcount[1] + wcount[2] + wcount[3];
        __syncthreads();
    }
    const int n = nlist;
    if (t == 0) cnt[blockIdx.x] = n;
    if (n == 0) return;
    const int cn = Cin * Cout, E = cn + (with_bias ? Cout : 0);
    float acc[kSwAcc];
    int ec[kSwAcc], ek[kSwAcc];                                    // element -> (input channel | -1 for the bias, output channel)
#pragma unroll
    for (int j = 0; j < kSwAcc; ++j) {
        acc[j] = 0.0f;
        const int e = t + 256 * j;
        if (e < cn) { ec[j] = e / Cout; ek[j] = e - ec[j] * Cout; }
        else { ec[j] = -1; ek[j] = e - cn; }
    }
    for (int i = 0; i < n; ++i) {
        const int64_t r = r0 + list[i];
        const float* xr = x + r * ldx;
        const float* g = dy + r * lddy;
#pragma unroll
        for (int j = 0; j < kSwAcc; ++j) {
            if (t + 256 * j < E) {
                const float gv = g[ek[j]];
                acc[j] = ec[j] >= 0 ? fmaf(xr[ec[j]], gv, acc[j]) : acc[j] + gv;
            }
        }
    }
    float* dst = part + (int64_t)blockIdx.x * E;
#pragma unroll
    for (int j = 0; j < kSwAcc; ++j)
        if (t + 256 * j < E) dst[t + 256 * j] = acc[j];
}

__global__ __launch_bounds__(256) void conv1x1_wgrad_rows_reduce_kernel(const float* part, const int* cnt, int G, int E, int cn, float* dw, float* db)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    float s = 0.0f;
    for (int b = 0; b < G; ++b)
        if (cnt[b]) s += part[(int64_t)b * E + e];                  // block order: deterministic
    if (e < cn) dw[e] = s;
    else db[e - cn] = s;
}

size_t pp_conv1x1_bwd_weight_sparse_workspace_bytes(int64_t M, int Cin, int Cout)
{
    if (M < 1 || Cin < 1 || Cout < 1 || (int64_t)Cin * Cout + Cout > 256 * kSwAcc) return 0;
    const int64_t G = cdiv(M, kSwRows);
    return align_up((size_t)G * ((size_t)Cin * Cout + Cout) * 4, 256) + align_up((size_t)G * 4, 256);
}

int pp_conv1x1_bwd_weight_sparse(const float* x, int64_t ldx, int64_t M, int Cin, const float* dy, int64_t lddy, int Cout,
                                 const unsigned char* row_flags, float* dw, float* dbias, void* workspace, size_t ws_bytes, pp_stream_t stream)
{
    if (!x || !dy || !row_flags || !dw || M < 1 || Cin < 1 || Cout < 1) return fail(PP_ERR_BAD_ARG, "conv1x1_bwd_weight_sparse: bad argument");
    const size_t need = pp_conv1x1_bwd_weight_sparse_workspace_bytes(M, Cin, Cout);
    if (need == 0) return fail(PP_ERR_UNSUPPORTED, "conv1x1_bwd_weight_sparse: Cin * Cout + Cout = %lld > %d", (long long)Cin * Cout + Cout, 256 * kSwAcc);
    if (!workspace || ws_bytes < need) return fail(PP_ERR_WORKSPACE, "conv1x1_bwd_weight_sparse: workspace %zu B < required %zu B", ws_bytes, need);
    const int64_t G = cdiv(M, kSwRows);
    if (G > 0x7FFFFFFF) return fail(PP_ERR_UNSUPPORTED, "conv1x1_bwd_weight_sparse: too many rows");
    const int cn = Cin * Cout, E = cn + (dbias ? Cout : 0);
    float* part = reinterpret_cast<float*>(workspace);
    int* cnt = reinterpret_cast<int*>(reinterpret_cast<char*>(workspace) + align_up((size_t)G * ((size_t)cn + Cout) * 4, 256));
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(conv1x1_wgrad_rows_partial_kernel, dim3((unsigned)G), dim3(256), 0, st, x, ldx, dy, lddy, M, Cin, Cout, row_flags,
                       dbias ? 1 : 0, part, cnt);
    if (int rc = check_launch("conv1x1_wgrad_rows_partial_kernel")) return rc;
    hipLaunchKernelGGL(conv1x1_wgrad_rows_reduce_kernel, dim3((unsigned)cdiv(E, 256)), dim3(256), 0, st, (const float*)part, (const int*)cnt, (int)G, E, cn,
                       dw, dbias);
    return check_launch("conv1x1_wgrad_rows_reduce_kernel");
}

static int bn_bwd_fused_impl(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* y_act, int64_t ldya, int act,
                             int64_t M, int C, const float* mean, const float* invstd, const float* gamma, float* dgamma,
                             float* dbeta, float* dx, int64_t lddx, float* dres, int64_t lddr, float grad_scale, const float* beta,
                             void* workspace, size_t ws_bytes, int32_t* sync, size_t sync_ints, pp_stream_t stream,
                             const unsigned char* row_flags)
{
    if (!x || !dy || !mean || !invstd || !gamma || !dgamma || !dbeta || !dx) return fail(PP_ERR_BAD_ARG, "bn_bwd_fused: null");
    if (act != 0 && !y_act && !beta) return fail(PP_ERR_BAD_ARG, "bn_bwd_fused: the mask needs the activation output or beta");
    if (!y_act && act != 0 && (dres || grad_scale != 1.0f))
        return fail(PP_ERR_BAD_ARG, "bn_bwd_fused: the mask can only be recomputed from x without residual / dropout");
    if (int rc = need_c4(C, "bn_bwd_fused")) return rc;
    if (ldx % 4 || lddy % 4 || lddx % 4 || (y_act && ldya % 4) || (dres && lddr % 4))
        return fail(PP_ERR_BAD_ARG, "bn_bwd_fused: ld must be multiples of 4");
    BnFusedGeom g = bn_fused_geom(M, C);
    if (int rc = bn_fused_check("bn_bwd_fused", M, C, g, workspace, ws_bytes, sync, sync_ints)) return rc;
    BnBwdArgs a{x, ldx, dy, lddy, y_act, ldya, act, M, C, mean, invstd, gamma, dgamma, dbeta, dx, lddx, dres, lddr,
                reinterpret_cast<xword*>(workspace), sync, g, grad_scale, y_act ? nullptr : beta, row_flags};
    if (g_bn_row_cache && g.rows_per_chunk <= (int64_t)g.nrl * kBnRowCache)
        hipLaunchKernelGGL(bn_fused_bwd_kernel<kBnRowCache>, dim3((unsigned)(g.nstrips * g.R)), dim3(kT), 0, as_stream(stream), a);
    else if (row_flags)
        hipLaunchKernelGGL((bn_fused_bwd_kernel<0, true>), dim3((unsigned)(g.nstrips * g.R)), dim3(kT), 0, as_stream(stream), a);
    else
        hipLaunchKernelGGL(bn_fused_bwd_kernel<0>, dim3((unsigned)(g.nstrips * g.R)), dim3(kT), 0, as_stream(stream), a);
    return check_launch("bn_fused_bwd_kernel");
}

int pp_bn_eval_affine(int C, const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                      float eps, float* scale, float* shift, pp_stream_t stream)
{
    if (!gamma || !beta || !running_mean || !running_var || !scale || !shift) return fail(PP_ERR_BAD_ARG, "bn_eval_affine: null");
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((unsigned)cdiv(C, kT)), dim3(kT), 0, as_stream(stream), C, gamma, beta,
                       running_mean, running_var, eps, scale, shift);
    return check_launch("bn_eval_affine_kernel");
}

int pp_scale_shift_act(const float* x, int64_t ldx, int64_t M, int C, const float* scale, const float* shift,
                       const float* residual, int64_t ldr, int act, float* y, int64_t ldy, pp_stream_t stream)
{
    if (!x || !scale || !shift || !y) return fail(PP_ERR_BAD_ARG, "scale_shift_act: null");
    if (int rc = need_c4(C, "scale_shift_act")) return rc;
    if (ldx % 4 || ldy % 4 || (residual && ldr % 4)) return fail(PP_ERR_BAD_ARG, "scale_shift_act: ld must be multiples of 4");
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(M * (C / 4))), dim3(kT), 0, as_stream(stream), x, ldx, scale, shift,
                       residual, ldr, act, y, ldy, M, C / 4);
    return check_launch("bn_apply_kernel");
}

int pp_bn_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, const float* y_act, int64_t ldya, int act,
              int64_t M, int C, const float* mean, const float* invstd, const float* gamma, float* dgamma, float* dbeta,
              float* dx, int64_t lddx, float* dres, int64_t lddr, void* workspace, size_t ws_bytes, pp_stream_t stream)
{
    if (!x || !dy || !mean || !invstd || !gamma || !dgamma || !dbeta || !dx) return fail(PP_ERR_BAD_ARG, "bn_bwd: null");
    if (act != 0 && !y_act) return fail(PP_ERR_BAD_ARG, "bn_bwd: activation output needed for the mask");
    if (int rc = need_c4(C, "bn_bwd")) return rc;
    ColReduceGeom g = col_geom(M, C);
    if (!workspace || ws_bytes < (size_t)g.nblk_rows * 2 * C * 4) return fail(PP_ERR_WORKSPACE, "bn_bwd: workspace");
    hipStream_t st = as_stream(stream);
    float* part = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL((col_reduce_kernel<1>), dim3(g.nblk_rows, g.nblk_cols), dim3(kT), 0, st, x, dy, y_act, act, mean,
                       invstd, M, C, ldx, lddy, ldya, g, part);
    if (int rc = check_launch("col_reduce_kernel<1>")) return rc;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)cdiv(C, 8)), dim3(kT), 0, st, part, g.nblk_rows, C, dgamma, dbeta);
    if (int rc = check_launch("bn_bwd_finalize_kernel")) return rc;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(M * (C / 4))), dim3(kT), 0, st, x, ldx, dy, lddy, y_act, ldya, act,
                       mean, invstd, gamma, dgamma, dbeta, 1.0f / (float)M, dx, lddx, dres, lddr, M, C / 4);
    return check_launch("bn_bwd_apply_kernel");
}

// ---- depthwise 3x3 ---------------------------------------------------------------------------------
static int dwconv_fwd_impl(const float* x, int64_t ldx, int B, int H, int W, int C, const float* w, int stride, int pad, int dil,
                           float* y, int64_t ldy, const Epilogue& epi, pp_stream_t stream)
{
    if (!x || !w || !y) return fail(PP_ERR_BAD_ARG, "dwconv fwd: null");
    if (int rc = need_c4(C, "dwconv fwd")) return rc;
    const int Ho = (H + 2 * pad - 2 * dil - 1) / stride + 1, Wo = (W + 2 * pad - 2 * dil - 1) / stride + 1;
    if (Ho < 1 || Wo < 1) return fail(PP_ERR_BAD_ARG, "dwconv fwd: empty output");
    if (stride == 1 && dil == 1 && g_dw_x4) {
        hipLaunchKernelGGL((dwconv_s1_x4_kernel<false>), dim3(grid_for((int64_t)B * Ho * ((Wo + 3) / 4) * (C / 4))), dim3(kT), 0,
                           as_stream(stream), x, ldx, B, H, W, C / 4, w, pad, y, ldy, Ho, Wo, epi);
        return check_launch("dwconv_s1_x4_kernel");
    }
    if (stride == 1 && dil == 2 && g_dw_x4) {
        hipLaunchKernelGGL((dwconv_s1_x4_kernel<false, 2>), dim3(grid_for((int64_t)B * Ho * ((Wo + 3) / 4) * (C / 4))), dim3(kT), 0,
                           as_stream(stream), x, ldx, B, H, W, C / 4, w, pad, y, ldy, Ho, Wo, epi);
        return check_launch("dwconv_s1_x4_kernel<2>");
    }
    hipLaunchKernelGGL(dwconv_fwd_kernel, dim3(grid_for((int64_t)B * Ho * Wo * (C / 4))), dim3(kT), 0, as_stream(stream), x,
                       ldx, B, H, W, C / 4, w, stride, pad, dil, y, ldy, Ho, Wo, epi);
    return check_launch("dwconv_fwd_kernel");
}

int pp_dwconv3x3_fwd(const float* x, int64_t ldx, int B, int H, int W, int C, const float* w, int stride, int pad, int dil,
                     float* y, int64_t ldy, pp_stream_t stream)
{
    return dwconv_fwd_impl(x, ldx, B, H, W, C, w, stride, pad, dil, y, ldy, Epilogue{}, stream);
}

int pp_dwconv3x3_fwd_bn_act(const float* x, int64_t ldx, int B, int H, int W, int C, const float* w, int stride, int pad,
                            int dil, const float* gamma, const float* beta, const float* running_mean,
                            const float* running_var, float eps, const float* residual, int64_t ldr, int act, float* y,
                            int64_t ldy, pp_stream_t stream)
{
    if (gamma && (!beta || !running_mean || !running_var)) return fail(PP_ERR_BAD_ARG, "dwconv fwd_bn_act: incomplete BatchNorm");
    if (act < 0 || act > 2) return fail(PP_ERR_BAD_ARG, "dwconv fwd_bn_act: act %d", act);
    if (ldy % 4 || (residual && ldr % 4)) return fail(PP_ERR_BAD_ARG, "dwconv fwd_bn_act: ld must be multiples of 4");
    Epilogue e{gamma, beta, running_mean, running_var, eps, residual, ldr, act};
    return dwconv_fwd_impl(x, ldx, B, H, W, C, w, stride, pad, dil, y, ldy, e, stream);
}

int pp_dwconv3x3_bwd_data(const float* dy, int64_t lddy, int B, int H, int W, int C, const float* w, int stride, int pad,
                          int dil, float* dx, int64_t lddx, pp_stream_t stream)
{
    if (!dy || !w || !dx) return fail(PP_ERR_BAD_ARG, "dwconv bwd_data: null");
    if (int rc = need_c4(C, "dwconv bwd_data")) return rc;
    const int Ho = (H + 2 * pad - 2 * dil - 1) / stride + 1, Wo = (W + 2 * pad - 2 * dil - 1) / stride + 1;
    if (stride == 1 && dil == 1 && g_dw_x4) {
        hipLaunchKernelGGL((dwconv_s1_x4_kernel<true>), dim3(grid_for((int64_t)B * H * ((W + 3) / 4) * (C / 4))), dim3(kT), 0,
                           as_stream(stream), dy, lddy, B, Ho, Wo, C / 4, w, 2 - pad, dx, lddx, H, W, Epilogue{});
        return check_launch("dwconv_s1_x4_kernel");
    }
    if (stride == 1 && dil == 2 && g_dw_x4) {          // the forward with mirrored weights and padding 2 * dil - pad
        hipLaunchKernelGGL((dwconv_s1_x4_kernel<true, 2>), dim3(grid_for((int64_t)B * H * ((W + 3) / 4) * (C / 4))), dim3(kT), 0,
                           as_stream(stream), dy, lddy, B, Ho, Wo, C / 4, w, 4 - pad, dx, lddx, H, W, Epilogue{});
        return check_launch("dwconv_s1_x4_kernel<2>");
    }
    if (stride == 2 && dil == 1 && pad >= 0 && pad <= 2 && g_dw_s2) {
        const int64_t items = (int64_t)B * ((H + pad + 1) / 2 + 1) * ((W + pad + 1) / 2 + 1) * (C / 4);
        hipLaunchKernelGGL(dwconv_s2_bwd_data_kernel, dim3(grid_for(items)), dim3(kT), 0, as_stream(stream), dy, lddy, B, Ho, Wo, C / 4, w, pad,
                           dx, lddx, H, W);
        return check_launch("dwconv_s2_bwd_data_kernel");
    }
    hipLaunchKernelGGL(dwconv_bwd_data_kernel, dim3(grid_for((int64_t)B * H * W * (C / 4))), dim3(kT), 0, as_stream(stream),
                       dy, lddy, B, Ho, Wo, C / 4, w, stride, pad, dil, dx, lddx, H, W);
    return check_launch("dwconv_bwd_data_kernel");
}

static int dwconv_bwd_weight_impl(const float* x, int64_t ldx, int B, int H, int W, int C, const float* dy, int64_t lddy,
                                  int stride, int pad, int dil, float* dw, void* workspace, size_t ws_bytes, pp_stream_t stream,
                                  const float* in_scale, const float* in_shift, int in_act, pp_reduce_job* job = nullptr)
{
    if (job) job->kind = 0;
    if (!x || !dy || !dw) return fail(PP_ERR_BAD_ARG, "dwconv bwd_weight: null");
    if (int rc = need_c4(C, "dwconv bwd_weight")) return rc;
    const int Ho = (H + 2 * pad - 2 * dil - 1) / stride + 1, Wo = (W + 2 * pad - 2 * dil - 1) / stride + 1;
    const int64_t M = (int64_t)B * Ho * Wo;
    if (M > 0x7FFFFFFFll) return fail(PP_ERR_UNSUPPORTED, "dwconv bwd_weight: more than 2^31 output pixels");
    // (the geometry must not depend on how large the caller's workspace happens to be: the summation order would)
    if (!workspace || ws_bytes < pp_colreduce_workspace_bytes(M, C)) return fail(PP_ERR_WORKSPACE, "dwconv bwd_weight: workspace");
    // maps of <= 8192 pixels (1/8 and 1/16 resolution, 192..960 channels): 32-quad column blocks x 8 row lanes, one item per
    // thread - 13.0-17.6 -> 10.1-12.1 us per call (tools/dw_wgrad_bench.py); the large maps keep full-width blocks (b3: 23 vs 27 us)
    const int cq_blk = g_dw_wgrad_cq_blk ? g_dw_wgrad_cq_blk : (M <= 8192 ? 32 : kT);
    const int passes = g_dw_wgrad_passes ? g_dw_wgrad_passes : (M <= 8192 ? 1 : 4);
    hipStream_t st = as_stream(stream);
    float* part = reinterpret_cast<float*>(workspace);
    if (stride == 1 && dil == 1 && Wo % 4 == 0 && g_dw_wgrad_x4) {
        // items of four pixels; the same workspace bound holds (never more row blocks than the one-pixel geometry)
        ColReduceGeom g4 = col_geom(M / 4, C, g_dw_wgrad_x4_blocks, cq_blk, passes);
        if (g4.nblk_rows <= 2048) {
            hipLaunchKernelGGL(dwconv_bwd_weight_x4_kernel, dim3(g4.nblk_rows, g4.nblk_cols), dim3(kT), 0, st, x, ldx, B, H, W, C / 4,
                               dy, lddy, Ho, Wo, pad, g4, part, in_scale, in_shift, in_act);
            if (int rc = check_launch("dwconv_bwd_weight_x4_kernel")) return rc;
            if (job) { job->part = part; job->dst = dw; job->cn = (int64_t)9 * C; job->splits = g4.nblk_rows; job->ntaps = 1; job->kind = 3; return PP_OK; }
            hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)cdiv((int64_t)9 * C, 8)), dim3(kT), 0, st, part, g4.nblk_rows,
                               (int64_t)9 * C, dw, 1.0f);
            return check_launch("sum_partials_kernel");
        }
    }
    ColReduceGeom g = col_geom(M, C, g_dw_wgrad_blocks, cq_blk, passes);
    if (g.nblk_rows > 2048) return fail(PP_ERR_UNSUPPORTED, "dwconv bwd_weight: geometry");
    hipLaunchKernelGGL(dwconv_bwd_weight_kernel, dim3(g.nblk_rows, g.nblk_cols), dim3(kT), 0, st, x, ldx, B, H, W, C / 4, dy,
                       lddy, Ho, Wo, stride, pad, dil, g, part, in_scale, in_shift, in_act);
    if (int rc = check_launch("dwconv_bwd_weight_kernel")) return rc;
    if (job) { job->part = part; job->dst = dw; job->cn = (int64_t)9 * C; job->splits = g.nblk_rows; job->ntaps = 1; job->kind = 3; return PP_OK; }
    hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)cdiv(9 * C, 8)), dim3(kT), 0, st, part, g.nblk_rows,
                       (int64_t)9 * C, dw, 1.0f);
    return check_launch("sum_partials_kernel");
}

int pp_dwconv3x3_bwd_weight(const float* x, int64_t ldx, int B, int H, int W, int C, const float* dy, int64_t lddy,
                            int stride, int pad, int dil, float* dw, void* workspace, size_t ws_bytes, pp_stream_t stream)
{
    return dwconv_bwd_weight_impl(x, ldx, B, H, W, C, dy, lddy, stride, pad, dil, dw, workspace, ws_bytes, stream, nullptr, nullptr, 0);
}

int pp_dwconv3x3_bwd_weight_partials(const float* x, int64_t ldx, int B, int H, int W, int C, const float* dy, int64_t lddy,
                                     int stride, int pad, int dil, float* dw, void* workspace, size_t ws_bytes, pp_reduce_job* job,
                                     pp_stream_t stream)
{
    if (!job) return fail(PP_ERR_BAD_ARG, "dwconv bwd_weight_partials: job is NULL");
    return dwconv_bwd_weight_impl(x, ldx, B, H, W, C, dy, lddy, stride, pad, dil, dw, workspace, ws_bytes, stream, nullptr, nullptr, 0, job);
}

int pp_dwconv3x3_bwd_weight_affine_in(const float* x_raw, int64_t ldx, int B, int H, int W, int C, const float* in_scale,
                                      const float* in_shift, int in_act, const float* dy, int64_t lddy, int stride, int pad, int dil,
                                      float* dw, void* workspace, size_t ws_bytes, pp_stream_t stream)
{
    if (!in_scale || !in_shift || in_act < 0 || in_act > 2) return fail(PP_ERR_BAD_ARG, "dwconv bwd_weight: input affine");
    return dwconv_bwd_weight_impl(x_raw, ldx, B, H, W, C, dy, lddy, stride, pad, dil, dw, workspace, ws_bytes, stream, in_scale, in_shift,
                                  in_act);
}

/* Geometry of the fused depthwise forward: work items (outputs, or groups of four along a row) and the column-reduce grid. */
static ColReduceGeom dw_fused_geom(int B, int Ho, int Wo, int C, int stride, int dil, bool* x4)
{
    // These kernels are latency-bound (18 / 9 dependent-free loads per item, then the arithmetic): ONE item per thread as long as
    // that gives at most ~2048 row blocks (= partial statistics rows), more items per thread only beyond.
    *x4 = stride == 1 && dil == 1 && g_dw_x4;
    const int64_t items = (int64_t)B * Ho * (*x4 ? (Wo + 3) / 4 : Wo);
    ColReduceGeom g;
    g.cq = C / 4;
    g.cq_blk = g.cq < kT ? g.cq : kT;
    g.rows_per_pass = kT / g.cq_blk;
    g.nblk_cols = (int)cdiv(g.cq, g.cq_blk);
    const int64_t passes = cdiv(items, g.rows_per_pass);
    const int64_t u = cdiv(passes, 2048);
    g.rows_per_block = (int64_t)g.rows_per_pass * u;
    g.nblk_rows = (int)cdiv(items, g.rows_per_block);
    return g;
}

int64_t pp_dwconv3x3_fwd_stats_rows(int B, int H, int W, int C, int stride, int pad, int dil)
{
    if (B < 1 || H < 1 || W < 1 || C < 4 || C % 4 || stride < 1 || dil < 1 || pad < 0) return 0;
    const int Ho = (H + 2 * pad - 2 * dil - 1) / stride + 1, Wo = (W + 2 * pad - 2 * dil - 1) / stride + 1;
    if (Ho < 1 || Wo < 1) return 0;
    bool x4;
    return dw_fused_geom(B, Ho, Wo, C, stride, dil, &x4).nblk_rows;
}

int pp_dwconv3x3_fwd_fused(const float* x, int64_t ldx, int B, int H, int W, int C, const float* w, int stride, int pad, int dil,
                           const float* in_scale, const float* in_shift, int in_act, float* y, int64_t ldy, float* stats,
                           size_t stats_floats, pp_stream_t stream)
{
    if (!x || !w || !y) return fail(PP_ERR_BAD_ARG, "dwconv fwd_fused: null");
    if ((in_scale == nullptr) != (in_shift == nullptr) || in_act < 0 || in_act > 2) return fail(PP_ERR_BAD_ARG, "dwconv fwd_fused: input affine");
    if (int rc = need_c4(C, "dwconv fwd_fused")) return rc;
    if (ldx % 4 || ldy % 4) return fail(PP_ERR_BAD_ARG, "dwconv fwd_fused: ld must be multiples of 4");
    const int Ho = (H + 2 * pad - 2 * dil - 1) / stride + 1, Wo = (W + 2 * pad - 2 * dil - 1) / stride + 1;
    if (Ho < 1 || Wo < 1) return fail(PP_ERR_BAD_ARG, "dwconv fwd_fused: empty output");
    if ((int64_t)B * Ho * Wo > 0x7FFFFFFFll) return fail(PP_ERR_UNSUPPORTED, "dwconv fwd_fused: more than 2^31 output pixels");
    bool x4;
    const ColReduceGeom g = dw_fused_geom(B, Ho, Wo, C, stride, dil, &x4);
    if (stats && stats_floats < (size_t)g.nblk_rows * 2 * C) return fail(PP_ERR_WORKSPACE, "dwconv fwd_fused: statistics buffer too small");
    const dim3 grid(g.nblk_rows, g.nblk_cols);
    if (x4) hipLaunchKernelGGL((dwconv_fwd_fused_kernel<true>), grid, dim3(kT), 0, as_stream(stream), x, ldx, B, H, W, C / 4, w, stride, pad,
                               dil, in_scale, in_shift, in_act, y, ldy, Ho, Wo, g, stats);
    else    hipLaunchKernelGGL((dwconv_fwd_fused_kernel<false>), grid, dim3(kT), 0, as_stream(stream), x, ldx, B, H, W, C / 4, w, stride,
                               pad, dil, in_scale, in_shift, in_act, y, ldy, Ho, Wo, g, stats);
    return check_launch("dwconv_fwd_fused_kernel");
}

/* Combine partial column statistics ([rows][2][C]: sum, sum of squares - what pp_conv2d_fwd_stats / pp_dwconv3x3_fwd_fused wrote)
 * into the training BatchNorm's batch mean / invstd, the folded scale / shift its consumers apply on load, and the running
 * statistics update (nn.BatchNorm2d training semantics; fixed-order fp64 combine). */
int pp_bn_finalize_partials(const float* stats, int64_t rows, int64_t M, int C, const float* gamma, const float* beta, float eps,
                            float momentum, float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                            float* shift, pp_stream_t stream)
{
    if (!stats || !gamma || !beta || !mean || !invstd || !scale || !shift) return fail(PP_ERR_BAD_ARG, "bn_finalize_partials: null");
    if (rows < 1 || rows > 0x7FFFFFFFll || M < 1 || C < 1) return fail(PP_ERR_BAD_ARG, "bn_finalize_partials: shape");
    if (C % 4 == 0 && (reinterpret_cast<uintptr_t>(stats) & 15) == 0) {
        hipLaunchKernelGGL(bn_finalize_quad_kernel, dim3((unsigned)(C / 4)), dim3(kT), 0, as_stream(stream), stats, (int)rows, C, (double)M,
                           gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift);
        return check_launch("bn_finalize_quad_kernel");
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)cdiv(C, 8)), dim3(kT), 0, as_stream(stream), stats, (int)rows, C, (double)M,
                       gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift);
    return check_launch("bn_finalize_kernel");
}

// ---- padding ------------------------------------------------------------------------------------------
int pp_pad2d(const float* x, int64_t ldx, int B, int H, int W, int C, int pad_top, int pad_left, int Hp, int Wp, float* y,
             int64_t ldy, pp_stream_t stream)
{
    if (!x || !y) return fail(PP_ERR_BAD_ARG, "pad2d: null");
    if (int rc = need_c4(C, "pad2d")) return rc;
    hipLaunchKernelGGL(pad_kernel, dim3(grid_for((int64_t)B * Hp * Wp * (C / 4))), dim3(kT), 0, as_stream(stream), x, ldx, B,
                       H, W, C / 4, pad_top, pad_left, y, ldy, Hp, Wp);
    return check_launch("pad_kernel");
}

int pp_crop2d_add(const float* xp, int64_t ldxp, int B, int Hp, int Wp, int C, int pad_top, int pad_left, const float* add,
                  int64_t ldadd, float* y, int64_t ldy, int H, int W, pp_stream_t stream)
{
    if (!xp || !y) return fail(PP_ERR_BAD_ARG, "crop2d_add: null");
    if (int rc = need_c4(C, "crop2d_add")) return rc;
    hipLaunchKernelGGL(crop_kernel, dim3(grid_for((int64_t)B * H * W * (C / 4))), dim3(kT), 0, as_stream(stream), xp, ldxp, B,
                       Hp, Wp, C / 4, pad_top, pad_left, add, ldadd, y, ldy, H, W);
    return check_launch("crop_kernel");
}

// ---- bilinear ---------------------------------------------------------------------------------------------
// Separable backward for large up-sampling factors (the ASPP x4 upsample of deeplab.py:49: 64x128 -> 16x32, 256 channels):
//   dx = Rh^T (dy Rw)   as two gathers - along W into tmp[B,Ho,W,C], then along H - instead of one thread walking a
// (2/s+2)^2 window (10x10 float4 loads per thread, 131 K threads: 145 us for 33.5 MB of dy).  Fixed order, no atomics.
__global__ __launch_bounds__(kT) void bilinear_bwd_sepw_kernel(const float* dy, int64_t lddy, int B, int Ho, int Wo, int cq,
                                                              float* tmp, int W, float sw, int align)
{
    const int64_t total = (int64_t)B * Ho * W * cq;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
        const int q = (int)(e % cq);
        int64_t t = e / cq;
        const int iw = (int)(t % W);
        const int64_t row = t / W;                     // b*Ho + oh
        int lo, hi;
        out_window(iw, W, Wo, sw, align, lo, hi);
        const float* src = dy + row * Wo * lddy + q * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ow = lo; ow <= hi; ++ow) {
            const Lerp l = lerp_src(ow, W, sw, align);
            const float wgt = (l.i0 == iw ? l.l0 : 0.0f) + (l.i1 == iw ? l.l1 : 0.0f);
            if (wgt != 0.0f) {
                const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)ow * lddy);
                acc.x = fmaf(wgt, v.x, acc.x); acc.y = fmaf(wgt, v.y, acc.y); acc.z = fmaf(wgt, v.z, acc.z); acc.w = fmaf(wgt, v.w, acc.w);
            }
        }
        *reinterpret_cast<float4*>(tmp + e * 4) = acc;
    }
}

__global__ __launch_bounds__(kT) void bilinear_bwd_seph_kernel(const float* tmp, int B, int Ho, int W, int cq, float* dx,
                                                              int64_t lddx, int H, float sh, int align)
{
    const int64_t total = (int64_t)B * H * W * cq;
    for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kT) {
        const int q = (int)(e % cq);
        int64_t t = e / cq;
        const int iw = (int)(t % W); t /= W;
        const int ih = (int)(t % H);
        const int b = (int)(t / H);
        int lo, hi;
        out_window(ih, H, Ho, sh, align, lo, hi);
        const float* src = tmp + (((int64_t)b * Ho) * W + iw) * cq * 4 + q * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int oh = lo; oh <= hi; ++oh) {
            const Lerp l = lerp_src(oh, H, sh, align);
            const float wgt = (l.i0 == ih ? l.l0 : 0.0f) + (l.i1 == ih ? l.l1 : 0.0f);
            if (wgt != 0.0f) {
                const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)oh * W * cq * 4);
                acc.x = fmaf(wgt, v.x, acc.x); acc.y = fmaf(wgt, v.y, acc.y); acc.z = fmaf(wgt, v.z, acc.z); acc.w = fmaf(wgt, v.w, acc.w);
            }
        }
        *reinterpret_cast<float4*>(dx + (((int64_t)b * H + ih) * W + iw) * lddx + q * 4) = acc;
    }
}

static void bil_scales(int H, int W, int Ho, int Wo, int align, float scale_h, float scale_w, float& sh, float& sw)
{
    if (align) {
        sh = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.0f;
        sw = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.0f;
    } else {
        sh = scale_h > 0.0f ? 1.0f / scale_h : (float)H / (float)Ho;
        sw = scale_w > 0.0f ? 1.0f / scale_w : (float)W / (float)Wo;
    }
}

int pp_bilinear_fwd(const float* x, int64_t ldx, int B, int H, int W, int C, float* y, int64_t ldy, int Ho, int Wo,
                    int align_corners, float scale_h, float scale_w, int out_nchw, pp_stream_t stream)
{
    if (!x || !y) return fail(PP_ERR_BAD_ARG, "bilinear fwd: null");
    float sh, sw;
    bil_scales(H, W, Ho, Wo, align_corners, scale_h, scale_w, sh, sw);
    hipStream_t st = as_stream(stream);
    if (out_nchw) {
        hipLaunchKernelGGL((bilinear_fwd_kernel<true>), dim3(grid_for((int64_t)B * C * Ho * Wo)), dim3(kT), 0, st, x, ldx, B, H,
                           W, C, y, ldy, Ho, Wo, sh, sw, align_corners);
    } else {
        if (int rc = need_c4(C, "bilinear fwd (NHWC out)")) return rc;
        hipLaunchKernelGGL((bilinear_fwd_kernel<false>), dim3(grid_for((int64_t)B * Ho * Wo * (C / 4))), dim3(kT), 0, st, x, ldx,
                           B, H, W, C, y, ldy, Ho, Wo, sh, sw, align_corners);
    }
    return check_launch("bilinear_fwd_kernel");
}

size_t pp_bilinear_bwd_workspace_bytes(int B, int Ho, int W, int C)
{
    if (B < 1 || Ho < 1 || W < 1 || C < 1) return 0;
    return align_up((size_t)B * Ho * W * C * 4, 256);
}

int pp_bilinear_bwd(const float* dy, int64_t lddy, int B, int Ho, int Wo, int C, float* dx, int64_t lddx, int H, int W,
                    int align_corners, float scale_h, float scale_w, int dy_nchw, void* workspace, size_t ws_bytes,
                    pp_stream_t stream)
{
    if (!dy || !dx) return fail(PP_ERR_BAD_ARG, "bilinear bwd: null");
    float sh, sw;
    bil_scales(H, W, Ho, Wo, align_corners, scale_h, scale_w, sh, sw);
    hipStream_t st = as_stream(stream);
    const bool win_ok = sh > 0.0f && sw > 0.0f && (int)(2.0f / sw) + 6 <= kMaxWin;   // candidate columns fit the weight table
    const bool up2_special = !align_corners && Ho == 2 * H && Wo == 2 * W && H >= 2 && W >= 2 &&
                             (scale_h == 0.0f || scale_h == 2.0f) && (scale_w == 0.0f || scale_w == 2.0f);
    if (!dy_nchw && C % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && sh > 0.0f && sw > 0.0f && Ho >= 3 * H && Wo >= 3 * W &&
        !up2_special && workspace && ws_bytes >= pp_bilinear_bwd_workspace_bytes(B, Ho, W, C) && g_bil_sep) {
        float* tmp = reinterpret_cast<float*>(workspace);
        const int cq = C / 4;
        hipLaunchKernelGGL(bilinear_bwd_sepw_kernel, dim3(grid_for((int64_t)B * Ho * W * cq)), dim3(kT), 0, st, dy, lddy, B, Ho, Wo, cq,
                           tmp, W, sw, align_corners);
        if (int rc = check_launch("bilinear_bwd_sepw_kernel")) return rc;
        hipLaunchKernelGGL(bilinear_bwd_seph_kernel, dim3(grid_for((int64_t)B * H * W * cq)), dim3(kT), 0, st, tmp, B, Ho, W, cq, dx,
                           lddx, H, sh, align_corners);
        return check_launch("bilinear_bwd_seph_kernel");
    }
    if (dy_nchw && win_ok) {
        hipLaunchKernelGGL(bilinear_bwd_planes_kernel, dim3(grid_for((int64_t)B * H * W * C)), dim3(kT), 0, st, dy, B, Ho, Wo, C,
                           dx, lddx, H, W, sh, sw, align_corners);
    } else if (dy_nchw) {
        hipLaunchKernelGGL((bilinear_bwd_kernel<true>), dim3(grid_for((int64_t)B * H * W * C)), dim3(kT), 0, st, dy, lddy, B, Ho,
                           Wo, C, dx, lddx, H, W, sh, sw, align_corners);
    } else if (C % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && !align_corners && Ho == 2 * H && Wo == 2 * W && H >= 2 && W >= 2 &&
               (scale_h == 0.0f || scale_h == 2.0f) && (scale_w == 0.0f || scale_w == 2.0f)) {
        hipLaunchKernelGGL(bilinear_up2_bwd4_kernel, dim3(grid_for((int64_t)B * H * W * (C / 4))), dim3(kT), 0, st, dy, lddy, B, C / 4,
                           dx, lddx, H, W);
    } else if (C % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && win_ok) {
        const int cq = C / 4;
        const int QB = cq >= 32 ? 32 : (cq >= 16 ? 16 : (cq >= 8 ? 8 : cq));
        const int NR = kT / QB;
        const int64_t nblk = (int64_t)B * H * W * cdiv(cq, QB);
        const int hwin = (int)(2.0f / sh) + 6;
        // row-split pays when the one-thread-per-(pixel, 4 channels) grid cannot fill the chip (ASPP x4 upsample: 131 K
        // threads); on the FPN decoder's x2 upsamples at 128x256 (1 M+ threads, 4-row windows) it is 3x slower
        if (hwin >= 4 && (int64_t)B * H * W * cq <= (1 << 19))
            hipLaunchKernelGGL(bilinear_bwd4r_kernel, dim3((unsigned)nblk), dim3(kT), 0, st, dy, lddy, B, Ho, Wo, cq, dx, lddx, H, W,
                               sh, sw, align_corners, QB, NR);
        else
            hipLaunchKernelGGL(bilinear_bwd4_kernel, dim3(grid_for((int64_t)B * H * W * cq)), dim3(kT), 0, st, dy, lddy, B, Ho, Wo,
                               cq, dx, lddx, H, W, sh, sw, align_corners);
    } else {
        hipLaunchKernelGGL((bilinear_bwd_kernel<false>), dim3(grid_for((int64_t)B * H * W * C)), dim3(kT), 0, st, dy, lddy, B, Ho,
                           Wo, C, dx, lddx, H, W, sh, sw, align_corners);
    }
    return check_launch("bilinear_bwd_kernel");
}

// ---- pooling / broadcast -------------------------------------------------------------------------------------
int pp_image_colsum(const float* x, int64_t ldx, int B, int64_t P, int C, float mul, float* out, int64_t ldo, pp_stream_t stream)
{
    if (!x || !out) return fail(PP_ERR_BAD_ARG, "image_colsum: null");
    hipLaunchKernelGGL(image_colsum_kernel, dim3((unsigned)cdiv(C, 64), (unsigned)B), dim3(kT), 0, as_stream(stream), x, ldx, P,
                       C, mul, out, ldo);
    return check_launch("image_colsum_kernel");
}

int pp_image_broadcast(const float* v, int64_t ldv, int B, int64_t P, int C, float mul, float* y, int64_t ldy, pp_stream_t stream)
{
    if (!v || !y) return fail(PP_ERR_BAD_ARG, "image_broadcast: null");
    if (int rc = need_c4(C, "image_broadcast")) return rc;
    hipLaunchKernelGGL(image_broadcast_kernel, dim3(grid_for((int64_t)B * P * (C / 4))), dim3(kT), 0, as_stream(stream), v, ldv,
                       B, P, C / 4, mul, y, ldy);
    return check_launch("image_broadcast_kernel");
}

// ---- dropout ---------------------------------------------------------------------------------------------------
int pp_dropout(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t M, int C, float p, uint64_t seed,
               const uint64_t* seed_dev, pp_stream_t stream)
{
    if (!x || !y) return fail(PP_ERR_BAD_ARG, "dropout: null");
    if (p < 0.0f || p >= 1.0f) return fail(PP_ERR_BAD_ARG, "dropout: p=%f outside [0,1)", (double)p);
    if (C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0)
        hipLaunchKernelGGL(dropout4_kernel, dim3(grid_for(M * (C / 4))), dim3(kT), 0, as_stream(stream), x, ldx, y, ldy, M, C / 4,
                           p, 1.0f / (1.0f - p), seed, seed_dev);
    else
        hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(M * C)), dim3(kT), 0, as_stream(stream), x, ldx, y, ldy, M, C, p,
                           1.0f / (1.0f - p), seed, seed_dev);
    return check_launch("dropout_kernel");
}

int pp_dropout2d(const float* x, int64_t ldx, float* y, int64_t ldy, int B, int64_t P, int C, float p, uint64_t seed,
                 const uint64_t* seed_dev, pp_stream_t stream)
{
    if (!x || !y) return fail(PP_ERR_BAD_ARG, "dropout2d: null");
    if (p < 0.0f || p >= 1.0f) return fail(PP_ERR_BAD_ARG, "dropout2d: p=%f outside [0,1)", (double)p);
    if (B <= 0 || P <= 0 || C <= 0) return fail(PP_ERR_BAD_ARG, "dropout2d: B=%d P=%lld C=%d", B, (long long)P, C);
    hipLaunchKernelGGL(dropout2d_kernel, dim3(grid_for((int64_t)B * P * C)), dim3(kT), 0, as_stream(stream), x, ldx, y, ldy, B, P, C,
                       p, 1.0f / (1.0f - p), seed, seed_dev);
    return check_launch("dropout2d_kernel");
}

// ---- loss ----------------------------------------------------------------------------------------------------------
size_t pp_sparse_ce_workspace_bytes(void) { return 1024 * 2 * 4 + 256; }

int pp_sparse_ce_fwd_bwd(const float* logits, int B, int C, int64_t HW, int64_t sB, int64_t sC, const int64_t* target,
                         int ignore_index, float* loss, float* count, const float* grad_out, float* dlogits, void* workspace,
                         size_t ws_bytes, pp_stream_t stream)
{
    if (!logits || !target || !loss || !count) return fail(PP_ERR_BAD_ARG, "sparse_ce: null");
    if (!workspace || ws_bytes < pp_sparse_ce_workspace_bytes()) return fail(PP_ERR_WORKSPACE, "sparse_ce: workspace");
    hipStream_t st = as_stream(stream);
    float* part = reinterpret_cast<float*>(workspace);
    int nblk = (int)cdiv((int64_t)B * HW, kT * 8);
    if (nblk > 1024) nblk = 1024;
    if (nblk < 1) nblk = 1;
    hipLaunchKernelGGL(ce_partial_kernel, dim3(nblk), dim3(kT), 0, st, logits, target, B, C, HW, sB, sC, ignore_index, part);
    if (int rc = check_launch("ce_partial_kernel")) return rc;
    hipLaunchKernelGGL(ce_finalize_wave_kernel, dim3(1), dim3(64), 0, st, part, nblk, loss, count);
    if (int rc = check_launch("ce_finalize_wave_kernel")) return rc;
    if (dlogits) {
        hipLaunchKernelGGL(ce_bwd_kernel, dim3(grid_for((int64_t)B * HW)), dim3(kT), 0, st, logits, target, B, C, HW, sB, sC,
                           ignore_index, count, grad_out, dlogits);
        if (int rc = check_launch("ce_bwd_kernel")) return rc;
    }
    return PP_OK;
}

size_t pp_sparse_ce_lowres_workspace_bytes(void) { return pp_sparse_ce_workspace_bytes(); }

int pp_sparse_ce_lowres_fwd_bwd(const float* low, int64_t ldx, int B, int C, int h, int w, int H, int W, int align_corners,
                                const int64_t* target, int ignore_index, float* loss, float* count, const float* grad_out,
                                float* dlow, int64_t lddx, void* workspace, size_t ws_bytes, pp_stream_t stream)
{
    if (!low || !target || !loss || !count) return fail(PP_ERR_BAD_ARG, "sparse_ce_lowres: null");
    if (B < 1 || C < 1 || h < 1 || w < 1 || H < 1 || W < 1 || ldx < C || (dlow && lddx < C))
        return fail(PP_ERR_BAD_ARG, "sparse_ce_lowres: bad shape B=%d C=%d %dx%d -> %dx%d ldx=%lld lddx=%lld", B, C, h, w, H, W,
                    (long long)ldx, (long long)lddx);
    if (C > 64) return fail(PP_ERR_UNSUPPORTED, "sparse_ce_lowres: C=%d > 64", C);
    if (!workspace || ws_bytes < pp_sparse_ce_lowres_workspace_bytes()) return fail(PP_ERR_WORKSPACE, "sparse_ce_lowres: workspace");
    hipStream_t st = as_stream(stream);
    float sh, sw;
    bil_scales(h, w, H, W, align_corners, 0.0f, 0.0f, sh, sw);
    const int al = align_corners ? 1 : 0;
    float* part = reinterpret_cast<float*>(workspace);
    int nblk = (int)cdiv((int64_t)B * H * W, kT * 8);
    if (nblk > 1024) nblk = 1024;
    if (nblk < 1) nblk = 1;
    const unsigned gb = (unsigned)cdiv((int64_t)B * h * w, kT);
#define PP_CE_LOW(CM, EX)                                                                                                      \
    do {                                                                                                                       \
        hipLaunchKernelGGL((ce_lowres_partial_kernel<CM, EX>), dim3(nblk), dim3(kT), 0, st, low, ldx, B, C, h, w, H, W, sh, sw, al, \
                           target, ignore_index, part);                                                                        \
        if (int rc = check_launch("ce_lowres_partial_kernel")) return rc;                                                      \
        hipLaunchKernelGGL(ce_finalize_wave_kernel, dim3(1), dim3(64), 0, st, part, nblk, loss, count);                        \
        if (int rc = check_launch("ce_finalize_wave_kernel")) return rc;                                                       \
        if (dlow) {                                                                                                            \
            hipLaunchKernelGGL((ce_lowres_bwd_kernel<CM, EX>), dim3(gb), dim3(kT), 0, st, low, ldx, B, C, h, w, H, W, sh, sw, al, \
                               target, ignore_index, count, grad_out, dlow, lddx);                                             \
            if (int rc = check_launch("ce_lowres_bwd_kernel")) return rc;                                                      \
        }                                                                                                                      \
    } while (0)
    switch (C) {
        case 11: PP_CE_LOW(11, true); break;
        case 19: PP_CE_LOW(19, true); break;
        case 21: PP_CE_LOW(21, true); break;
        default:
            if (C <= 32) PP_CE_LOW(32, false);
            else PP_CE_LOW(64, false);
    }
#undef PP_CE_LOW
    return PP_OK;
}

// ---- optimiser ------------------------------------------------------------------------------------------------------
int pp_adam_step_flat(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t n_split,
                      float lr_a, float lr_b, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                      float grad_scale, const float* hyper_dev, pp_stream_t stream)
{
    if (!params || !grads || !exp_avg || !exp_avg_sq) return fail(PP_ERR_BAD_ARG, "adam: null");
    if (n < 1 || step < 1) return fail(PP_ERR_BAD_ARG, "adam: bad n/step");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(kT), 0, as_stream(stream), params, grads, exp_avg, exp_avg_sq, n,
                       n_split, lr_a, lr_b, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale, hyper_dev);
    return check_launch("adam_kernel");
}

int pp_sgd_step_flat(float* params, const float* grads, float* momentum_buf, int64_t n, int64_t n_split, float lr_a, float lr_b,
                     float momentum, float weight_decay, int64_t step, float grad_scale, const float* hyper_dev,
                     pp_stream_t stream)
{
    if (!params || !grads || (momentum != 0.0f && !momentum_buf)) return fail(PP_ERR_BAD_ARG, "sgd: null");
    if (n < 1 || step < 1) return fail(PP_ERR_BAD_ARG, "sgd: bad n/step");
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n)), dim3(kT), 0, as_stream(stream), params, grads, momentum_buf, n, n_split,
                       lr_a, lr_b, momentum, weight_decay, step == 1 ? 1 : 0, grad_scale, hyper_dev);
    return check_launch("sgd_kernel");
}

int pp_add2d(const float* a, int64_t lda, const float* b, int64_t ldb, float* y, int64_t ldy, int64_t M, int C, pp_stream_t stream)
{
    if (!a || !b || !y) return fail(PP_ERR_BAD_ARG, "add2d: null");
    if (C % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldy % 4 == 0 &&
        ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(y)) & 15) == 0) {
        const int flat = (lda == C && ldb == C && ldy == C) ? 1 : 0;
        hipLaunchKernelGGL(add2d_v4_kernel, dim3(grid_for(M * (C / 4))), dim3(kT), 0, as_stream(stream), a, lda, b, ldb, y, ldy, M, C / 4, flat);
        return check_launch("add2d_v4_kernel");
    }
    hipLaunchKernelGGL(add2d_kernel, dim3(grid_for(M * C)), dim3(kT), 0, as_stream(stream), a, lda, b, ldb, y, ldy, M, C);
    return check_launch("add2d_kernel");
}

// ---- layout ------------------------------------------------------------------------------------------------------------
int pp_nchw_to_nhwc(const float* x, int B, int C, int64_t HW, float* y, int64_t ldy, pp_stream_t stream)
{
    if (!x || !y) return fail(PP_ERR_BAD_ARG, "nchw_to_nhwc: null");
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((int64_t)B * HW * C)), dim3(kT), 0, as_stream(stream), x, B, C, HW, y, ldy);
    return check_launch("nchw_to_nhwc_kernel");
}

int pp_nhwc_to_nchw(const float* x, int64_t ldx, int B, int C, int64_t HW, float* y, pp_stream_t stream)
{
    if (!x || !y) return fail(PP_ERR_BAD_ARG, "nhwc_to_nchw: null");
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((int64_t)B * HW * C)), dim3(kT), 0, as_stream(stream), x, ldx, B, C, HW, y);
    return check_launch("nhwc_to_nchw_kernel");
}

}  // extern "C"
