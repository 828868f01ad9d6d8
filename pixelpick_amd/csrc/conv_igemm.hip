// conv_igemm.hip — dense convolutions of the PixelPick networks as NHWC implicit GEMMs on the fp32
// MFMA pipe of gfx950 (v_mfma_f32_32x32x2_f32: exact f32 fma chains at 157 TF peak).
//
// Replaces the ATen/cuDNN calls behind every dense nn.Conv2d of the reference (SURVEY.md §8 N3, N7, N9,
// N10, N12, N13 and R2/R3): networks/mobilenet_v2.py:42,48,56 (pointwise), networks/aspp.py:9-10,49-58
// (1x1 + atrous 3x3), networks/deeplab.py:24 (low-level 1x1), networks/decoders.py:107,111,116
// (SegmentHead 3x3 + classifier), and their autograd backward (model.py:121).
//
//   forward      Y[m][n]   = sum_{t,c} X[pix(m,t)][c] * W[t][c][n] (+ bias[n])     m = (b,oh,ow)
//   backward-x   dX[m][c]  = sum_{t,n} dY[pix'(m,t)][n] * W[t][c][n]               (stride 1)
//   backward-w   dW[t][c][n] = sum_m X[pix(m,t)][c] * dY[m][n]                     (split over m)
//
// Layouts: activations NHWC with an explicit pixel stride `ld` (so channel slices of a wider tensor —
// the reference's torch.cat inputs/outputs, aspp.py:73, deeplab.py:50 — are read and written in place);
// weights HWIO = [kh][kw][Cin][Cout].  Taps that fall outside the image for every output pixel
// (atrous d=18 on a 16x32 map) are dropped on the host.
//
// Tiling: 256 threads = 4 waves; block tile BM x BN, K step 16; operands staged through LDS by
// registers (next K-step's global loads are in flight while the current one is multiplied).  A fragment
// reads are ds_read_b128 using a permuted K order (lane-half h consumes k = 8q+4h+j), B fragment reads
// are conflict-free ds_read_b32.
#include <atomic>
#include <type_traits>

#include "pp_common.h"
#include "bn_xchg.h"
#include "conv_types.h"

namespace pp {


static int g_conv_xcd_remap = 1;
static int g_conv_novec = 0;
static int g_conv_lds_pad = 0;     // extra dynamic LDS bytes for the 128x128 kernels: caps co-resident blocks per CU
static int g_conv_variant = 0;   // large-tile kernel: 0 = 128x128 tiles (default), 2 = 128x64 tiles (A/B)

typedef float f32x16s __attribute__((ext_vector_type(16)));      // sixteen consecutive floats at a wave-uniform, 64-byte aligned address: one s_load_dwordx16
constexpr int BK = 16;
// dword-aligned wide loads (global loads need dword alignment only): the packed three-channel image of the stem
struct __attribute__((packed, aligned(4))) StemF4 { float x, y, z, w; };
struct __attribute__((packed, aligned(4))) StemF3 { float x, y, z; };
struct __attribute__((packed, aligned(4))) StemF2 { float x, y; };

// ---- LDS tiles ---------------------------------------------------------------------------------------
// "MK" form: tile[rows][BK + 4]   (K contiguous, 80-B row pitch: conflict-free ds_read_b128)
// "KM" form: tile[BK][rows + 4]   (rows contiguous: conflict-free ds_read_b32)

// One K-step (16) of MFMAs for a wave tile of TM x TN 32x32 blocks.
//   A_MK: A tile in MK form, else KM form.   B_NK: B tile in "NK" (= MK-like) form, else KN form.
template <int TM, int TN, bool A_MK, bool B_NK, int PITCH_A, int PITCH_B, int BKT = BK>
__device__ __forceinline__ void mma_step(const float* As, const float* Bs, int a_row0, int b_col0, f32x16 (&acc)[TM][TN])
{
    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, h = lane >> 5;
    constexpr int NQ = BKT / 8;
    // All fragment reads of the K-step are issued up front (the MFMAs of group q only wait for group q's reads
    // through hipcc's counted lgkmcnt), so no LDS latency sits between MFMAs: measured, reads placed lazily in
    // front of each group of 4 MFMAs left the matrix pipe idle ~25 % of the MFMA phase at one wave per SIMD.
    float a[NQ][TM][4], b[NQ][TN][4];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int k0 = 8 * q + 4 * h;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int r = a_row0 + tm * 32 + l31;
            if constexpr (A_MK) {
                const float4 v = *reinterpret_cast<const float4*>(As + r * PITCH_A + k0);
                a[q][tm][0] = v.x; a[q][tm][1] = v.y; a[q][tm][2] = v.z; a[q][tm][3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) a[q][tm][j] = As[(k0 + j) * PITCH_A + r];
            }
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int c = b_col0 + tn * 32 + l31;
            if constexpr (B_NK) {
                const float4 v = *reinterpret_cast<const float4*>(Bs + c * PITCH_B + k0);
                b[q][tn][0] = v.x; b[q][tn][1] = v.y; b[q][tn][2] = v.z; b[q][tn][3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) b[q][tn][j] = Bs[(k0 + j) * PITCH_B + c];
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][tm][j], b[q][tn][j], acc[tm][tn], 0, 0, 0);
}

// ---- epilogue that also finishes a training BatchNorm (+ residual, activation) --------------------------------------------------
// conv -> BatchNorm as ONE launch for the layers whose whole grid is co-resident (<= half of the kernel's occupancy x CUs, host-
// checked): a 1/16-resolution pointwise convolution is 12-20 us and its BatchNorm launch another 11.5, of which 4.5 are dispatch and
// 2.3 the two passes over a tensor the convolution had in registers a moment earlier (profiles/r03_bn_phases.txt).  Here the block
//   1. sums its tile's columns (lane -> 32-lane halves -> the WM wave rows through LDS, fixed order),
//   2. publishes the 32-channel strips' partial sums as tagged words and combines the strip's R partial rows (bn_xchg.h: the
//      exchange of bn_fused_fwd_kernel - same memory, same epoch protocol, fp64 combine in a fixed order),
//   3. computes mean / invstd / scale / shift with bn_fused_fwd_kernel's expressions (the backward recomputes the activation mask
//      from them), block row 0 writes them and the running statistics,
//   4. stores the raw convolution (the BatchNorm backward's input) AND the normalised, activated output.
// scratch: >= (2 * WM * BN + 64) floats + (256 + 2 * BN) doubles of LDS nobody reads any more (the caller put a barrier in front).
// scratch layout shared by the two fused epilogues (LDS nobody reads any more): doubles shd[256], tot[NSTRIP][64]; floats
// colsum[2][WM][BNT], aff[2][BNT]; one unsigned
template <int BNT, int WM>
struct BnScratch {
    static constexpr int NSTRIP = BNT / 32;
    double* shd; double* tot; float* colsum; float* aff; unsigned* sh_tag;
    __device__ __forceinline__ explicit BnScratch(float* scratch)
    {
        shd = reinterpret_cast<double*>(scratch);
        tot = shd + 256;
        colsum = reinterpret_cast<float*>(tot + NSTRIP * 64);
        aff = colsum + 2 * WM * BNT;
        sh_tag = reinterpret_cast<unsigned*>(aff + 2 * BNT);
    }
    static constexpr int kFloats = (256 + NSTRIP * 64) * 2 + 2 * WM * BNT + 2 * BNT + 4;
};

// colsum[stat][wave row][column] is filled (no barrier yet) -> aff[0][c] = scale, aff[1][c] = shift of the tile's columns; block row 0
// writes mean / invstd / running statistics.  Ends with a barrier.
template <int BNT, int WM, bool BWDF = false>
__device__ __forceinline__ void bn_block_finish(const ConvParams& p, const BnScratch<BNT, WM>& S, int n0, int mt, unsigned tag0)
{
    constexpr int NSTRIP = BNT / 32;
    const BnTrain& bn = p.bn;
    const int tid = threadIdx.x;
    if (tid == 0) *S.sh_tag = tag0;
    __syncthreads();
    const unsigned tag = *S.sh_tag;
    // publish: thread (stat, column) adds the wave rows in order
    if (tid < 2 * BNT) {
        const int stat = tid / BNT, c = tid - stat * BNT;
        const int n = n0 + c;
        if (n < p.Cn) {
            float v = S.colsum[(stat * WM + 0) * BNT + c];
#pragma unroll
            for (int w = 1; w < WM; ++w) v += S.colsum[(stat * WM + w) * BNT + c];
            const int strip = n >> 5, cl = n & 31;
            xchg_put(bn.part + ((int64_t)strip * bn.R + mt) * 64 + stat * 32 + cl, v, tag);
        }
    }
    // combine: all strips of the tile at once - thread (sub, strip, output) adds partial rows sub, sub + NSUB, ... in that order,
    // sixteen words in flight (a strip after the other with four in flight was ~6 dependent round trips to fine-grained memory:
    // the fused launch ran 10 us longer than the plain convolution, i.e. as long as the BatchNorm launch it replaces)
    {
        constexpr int NSUB = 256 / (64 * NSTRIP);
        const int o = tid & 63, sl = (tid >> 6) % NSTRIP, sub = tid / (64 * NSTRIP);
        const int strip = (n0 >> 5) + sl;
        double sacc = 0.0;
        if (strip * 32 < p.Cn) {
            const xword* pp_ = bn.part + (int64_t)strip * bn.R * 64 + o;
            int c = sub;
            for (; c + 15 * NSUB < bn.R; c += 16 * NSUB) {
                xword wv[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) wv[j] = __hip_atomic_load(pp_ + (int64_t)(c + j * NSUB) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    sacc += (double)((unsigned)(wv[j] >> 32) == tag ? __uint_as_float((unsigned)wv[j]) : xchg_get(pp_ + (int64_t)(c + j * NSUB) * 64, tag));
            }
            for (; c + 3 * NSUB < bn.R; c += 4 * NSUB) {
                xword wv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) wv[j] = __hip_atomic_load(pp_ + (int64_t)(c + j * NSUB) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    sacc += (double)((unsigned)(wv[j] >> 32) == tag ? __uint_as_float((unsigned)wv[j]) : xchg_get(pp_ + (int64_t)(c + j * NSUB) * 64, tag));
            }
            for (; c < bn.R; c += NSUB) sacc += (double)xchg_get(pp_ + (int64_t)c * 64, tag);
        }
        S.shd[tid] = sacc;
        __syncthreads();
        if (tid < 64 * NSTRIP) {
            double a = S.shd[tid];
#pragma unroll
            for (int k = 1; k < NSUB; ++k) a += S.shd[k * 64 * NSTRIP + tid];
            S.tot[tid] = a;                                 // tot[sl * 64 + o]: tid = sl * 64 + o for tid < 64 * NSTRIP
        }
        __syncthreads();
    }
    launch_done(bn.sync);
    if constexpr (BWDF) {
        // backward: aff[0][c] = sum of the masked gradient, aff[1][c] = sum of masked gradient x normalised input (bn_fused_bwd_kernel)
        if (tid < BNT) {
            const int n = n0 + tid;
            if (n < p.Cn) {
                const int sl = tid >> 5, cl = tid & 31;
                const float db = (float)S.tot[sl * 64 + cl], dg = (float)S.tot[sl * 64 + 32 + cl];
                S.aff[tid] = db;
                S.aff[BNT + tid] = dg;
                if (mt == 0) { bn.dbeta[n] = db; bn.dgamma[n] = dg; }
            }
        }
        __syncthreads();
        return;
    }
    // per-column affine (bn_fused_fwd_kernel's expressions: the backward recomputes the activation mask from them)
    if (tid < BNT) {
        const int n = n0 + tid;
        if (n < p.Cn) {
            const int sl = tid >> 5, cl = tid & 31;
            const double count = (double)p.M;
            const double mu = S.tot[sl * 64 + cl] / count;
            double var = S.tot[sl * 64 + 32 + cl] / count - mu * mu;
            if (var < 0.0) var = 0.0;
            const float is = (float)(1.0 / sqrt(var + (double)bn.eps));
            const float sc = bn.gamma[n] * is;
            S.aff[tid] = sc;
            S.aff[BNT + tid] = bn.beta[n] - (float)mu * sc;
            if (mt == 0) {
                bn.mean[n] = (float)mu;
                bn.invstd[n] = is;
                if (bn.running_mean) {
                    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
                    bn.running_mean[n] = (1.0f - bn.momentum) * bn.running_mean[n] + bn.momentum * (float)mu;
                    bn.running_var[n] = (1.0f - bn.momentum) * bn.running_var[n] + bn.momentum * (float)unbiased;
                }
            }
        }
    }
    __syncthreads();
}

template <int TM, int TN, int WM, int WN>
__device__ __forceinline__ void conv_epilogue_bn(const ConvParams& p, f32x16 (&acc)[TM][TN], int64_t m0, int n0, int wm, int wn, int mt,
                                                 float* scratch, unsigned tag0)
{
    constexpr int BNT = WN * TN * 32;                      // columns of the block tile
    const BnTrain& bn = p.bn;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const BnScratch<BNT, WM> S(scratch);
    // column sums of what this lane holds (rows < M)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const float v = m < p.M ? acc[tm][tn][r] : 0.0f;
                s1 += v;
                s2 = fmaf(v, v, s2);
            }
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (hh == 0) {
            const int c = (wn * TN + tn) * 32 + l31;
            S.colsum[(0 * WM + wm) * BNT + c] = s1;
            S.colsum[(1 * WM + wm) * BNT + c] = s2;
        }
    }
    bn_block_finish<BNT, WM>(p, S, n0, mt, tag0);
    // stores: the raw convolution (the BatchNorm backward's input) and the normalised, activated output
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int c = (wn * TN + tn) * 32 + l31;
        const int n = n0 + c;
        if (n >= p.Cn) continue;
        const float sc = S.aff[c], sf = S.aff[BNT + c];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (m < p.M) {
                    const float v = acc[tm][tn][r];
                    p.y[m * p.ldy + n] = v;
                    float o = fmaf(v, sc, sf);
                    if (bn.res) o += bn.res[m * bn.ldr + n];
                    bn.y[m * bn.ldy + n] = epi_act(o, bn.act);
                }
            }
    }
}

// Backward-data convolution + the BatchNorm backward of the layer in FRONT of it (conv -> BatchNorm -> act -> THIS conv, the
// activated output having no other consumer): the tile in registers is the gradient of the BatchNorm's output; with the BatchNorm's
// input (one more tile read) the lane forms the masked gradient u and the two column sums of bn_fused_bwd_kernel, the blocks
// exchange them as in the forward form, and the lane writes the gradient of the BatchNorm's INPUT (bn_dx, the BatchNorm kernels'
// own expression).  The gradient of the BatchNorm's output is never written.
template <int TM, int TN, int WM, int WN>
__device__ __forceinline__ void conv_epilogue_bn_bwd(const ConvParams& p, f32x16 (&acc)[TM][TN], int64_t m0, int n0, int wm, int wn, int mt,
                                                     float* scratch, unsigned tag0)
{
    constexpr int BNT = WN * TN * 32;
    const BnTrain& bn = p.bn;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const BnScratch<BNT, WM> S(scratch);
    float xs[TM][TN][16];                                  // the BatchNorm's input at this lane's elements
    float mu_[TN], is_[TN], ga_[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + (wn * TN + tn) * 32 + l31;
        const bool ok = n < p.Cn;
        const float mu = ok ? bn.mean[n] : 0.0f, is = ok ? bn.invstd[n] : 0.0f, ga = ok ? bn.gamma[n] : 0.0f, be = ok ? bn.beta[n] : 0.0f;
        mu_[tn] = mu; is_[tn] = is; ga_[tn] = ga;
        const float zsc = ga * is, zsf = be - mu * zsc;    // the forward's scale / shift: z = fma(x, zsc, zsf) is what the activation saw
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const bool in = ok && m < p.M;
                const float x = in ? bn.bx[m * bn.ldbx + n] : 0.0f;
                xs[tm][tn][r] = x;
                float u = in ? acc[tm][tn][r] : 0.0f;
                if (in && bn.gin) u += bn.gin[m * bn.ldgin + n];
                if (bn.act != 0) u *= act_mask(fmaf(x, zsc, zsf), bn.act);
                acc[tm][tn][r] = u;
                s1 += u;
                s2 = fmaf(u, (x - mu) * is, s2);
            }
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (hh == 0) {
            const int c = (wn * TN + tn) * 32 + l31;
            S.colsum[(0 * WM + wm) * BNT + c] = s1;
            S.colsum[(1 * WM + wm) * BNT + c] = s2;
        }
    }
    bn_block_finish<BNT, WM, true>(p, S, n0, mt, tag0);
    const float inv_count = 1.0f / (float)p.M;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int c = (wn * TN + tn) * 32 + l31;
        const int n = n0 + c;
        if (n >= p.Cn) continue;
        const float db = S.aff[c], dg = S.aff[BNT + c];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (m < p.M) {
                    if (bn.dres) bn.dres[m * bn.lddr + n] = acc[tm][tn][r];
                    bn.y[m * bn.ldy + n] = bn_dx(acc[tm][tn][r], xs[tm][tn][r], mu_[tn], is_[tn], ga_[tn], db, dg, inv_count);
                }
            }
    }
}

// ---- forward / backward-data kernel ---------------------------------------------------------------------
// BWD == false: B operand W[t][c][n]  -> KN form (n contiguous in memory)
// BWD == true : B operand W[t][n'][k'] with k' = conv Cout contiguous -> NK form
// VEC: every channel count / pixel stride is a multiple of 4 and the bases are 16-B aligned.  Then all global
// loads are UNCONDITIONAL float4 loads from a clamped address and the zero-fill of out-of-range elements happens
// when the registers are written to LDS — so the loads of step k+1 stay in flight across the MFMAs of step k
// (conditional loads make hipcc drain vmcnt(0) at the branch joins, serialising memory latency and MFMA work).
template <int BM, int BN, int WM, int WN, bool BWD, bool VEC, int BKT = BK>
__global__ __launch_bounds__(kThreads) void conv_igemm_kernel(ConvParams p)
{
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int KQ = BKT / 4;                                   // float4 per tile row along K
    constexpr int RPP = kThreads / KQ;                            // A rows staged per pass
    constexpr int PITCH_MK = BKT + 4;                             // 16 -> 20, 64 -> 68 floats: conflict-free ds_read_b128
    constexpr int PITCH_A = PITCH_MK;
    constexpr int PITCH_B = BWD ? PITCH_MK : BN + 4;
    constexpr int A_F4 = BM * BKT / 4 / kThreads;                // float4 per thread for the A tile
    constexpr int B_TOTAL = BN * BKT / 4;
    constexpr int B_F4 = (B_TOTAL + kThreads - 1) / kThreads;
    static_assert(A_F4 >= 1, "tile too small for 256 threads");

    __shared__ __attribute__((aligned(16))) float As[BM * PITCH_A];
    __shared__ __attribute__((aligned(16))) float Bs[BWD ? BN * PITCH_MK : BKT * (BN + 4)];

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order (blocks are dispatched round-robin over the 8 XCDs, each with its own L2): XCD x
    // works through a contiguous range of M tiles with the N tiles of one M tile back to back, so the A rows
    // (and the 3x3 halo rows shared with the neighbouring M tile) are re-read from that XCD's L2.
    int mt, nt;
    {
        const int ntn = p.n_tiles, nblk = gridDim.x;
        const int bid = blockIdx.x;
        const int per_xcd = nblk / 8;
        if (p.xcd_remap && per_xcd * 8 == nblk) {
            const int lin = (bid & 7) * per_xcd + (bid >> 3);
            mt = lin / ntn;
            nt = lin - mt * ntn;
        } else {
            mt = bid / ntn;
            nt = bid - mt * ntn;
        }
    }
    const int64_t m0 = (int64_t)mt * BM;
    const int n0 = nt * BN;

    // A staging: thread -> (row, k-quad); rows fixed for the whole K loop
    // MK-form staging map: within a wave, 64/KQ consecutive LANES take consecutive ROWS of one k-quad (same address set per
    // wave instruction as quad-fastest, so the global loads coalesce identically), which makes the 16-byte LDS stores
    // walk rows at the 80-byte pitch like the fragment reads do instead of packing 4 quads of one row into 4 lanes
    constexpr int RPW = 64 / KQ;                                  // rows per wave and pass
    const int a_kq = (tid & 63) / RPW;
    const int a_rl = (tid >> 6) * RPW + (tid & (RPW - 1));       // row of this thread within a pass (RPP rows)
    int a_b[A_F4], a_ih[A_F4], a_iw[A_F4];
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
        const int64_t m = m0 + a_rl + i * RPP;
        if (m < p.M) {
            const unsigned mu = (unsigned)m;                  // M < 2^31 (checked on the host)
            const unsigned t = mu / (unsigned)p.Wo;
            const int ow = (int)(mu - t * (unsigned)p.Wo);
            const unsigned bb = t / (unsigned)p.Ho;
            const int oh = (int)(t - bb * (unsigned)p.Ho);
            a_b[i] = (int)bb;
            a_ih[i] = oh * p.stride;
            a_iw[i] = ow * p.stride;
        } else {
            a_b[i] = -1; a_ih[i] = 0; a_iw[i] = 0;
        }
    }

    const int nchunk = (p.Ck + BKT - 1) / BKT;
    const int nk = p.taps.n * nchunk;
    const bool w_vec = BWD ? (p.Cout % 4 == 0) : (p.Cout % 4 == 0);

    struct Regs {
        float4 a[A_F4], b[B_F4];
        unsigned ok;       // VEC: bit i = a[i] valid, bit 8+i = b[i] valid
        int c0;            // channel base of the K step these registers hold
    };
    Regs R0;

    auto load_tiles = [&](int ks, Regs& R) {
        float4 (&ra)[A_F4] = R.a;
        float4 (&rb)[B_F4] = R.b;
        unsigned& okmask = R.ok;
        // K order: channel chunk OUTER, tap inner - the taps of a 3x3 window re-read the same input rows (shifted by
        // one pixel), so with the tap innermost those lines are still in L1/L2; tap-outer walked the whole 304-channel
        // row set once per tap (5 MB per XCD > its 4 MB L2: FETCH_SIZE 8x the algorithmic input)
        int ti, c0;
        if (p.tap_inner) {
            const int ch = ks / p.taps.n;
            ti = ks - ch * p.taps.n;
            c0 = ch * BKT;
        } else {
            ti = ks / nchunk;
            c0 = (ks - ti * nchunk) * BKT;
        }
        const int dh = p.taps.dh[ti], dw = p.taps.dw[ti];
        const int wt = p.taps.widx[ti];
        R.c0 = c0;
        if constexpr (VEC) {
            okmask = 0;
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                int ih = a_ih[i] + dh, iw = a_iw[i] + dw;
                const int c = c0 + a_kq * 4;
                bool ok = a_b[i] >= 0 && c < p.Ck;
                if (BWD && p.bwd_stride > 1) {
                    ok = ok && ih >= 0 && iw >= 0 && (ih % p.bwd_stride) == 0 && (iw % p.bwd_stride) == 0;
                    ih /= p.bwd_stride;
                    iw /= p.bwd_stride;
                }
                ok = ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                const int64_t off = ok ? (((int64_t)a_b[i] * p.H + ih) * p.W + iw) * p.ldx + c : 0;
                ra[i] = *reinterpret_cast<const float4*>(p.x + off);
                okmask |= ok ? (1u << i) : 0u;
            }
#pragma unroll
            for (int i = 0; i < B_F4; ++i) {
                const int e = tid + i * kThreads;
                bool ok;
                int64_t off;
                if constexpr (!BWD) {
                    const int kr = e / (BN / 4), nq = e % (BN / 4);
                    const int c = c0 + kr, n = n0 + nq * 4;
                    ok = e < B_TOTAL && c < p.Cin && n < p.Cout;
                    off = ((int64_t)wt * p.Cin + c) * p.Cout + n;
                } else {
                    const int col = (e >> 6) * RPW + (e & (RPW - 1)), kq = (e & 63) / RPW;
                    const int cin = n0 + col, k = c0 + kq * 4;
                    ok = e < B_TOTAL && cin < p.Cin && k < p.Cout;
                    off = ((int64_t)wt * p.Cin + cin) * p.Cout + k;
                }
                rb[i] = *reinterpret_cast<const float4*>(p.w + (ok ? off : 0));
                okmask |= ok ? (1u << (8 + i)) : 0u;
            }
            return;
        }
        // A: gather BK channels of the tap's input pixel for each row
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            int ih = a_ih[i] + dh, iw = a_iw[i] + dw;
            const int c = c0 + a_kq * 4;
            bool sok = true;
            if (BWD && p.bwd_stride > 1) {
                sok = ih >= 0 && iw >= 0 && (ih % p.bwd_stride) == 0 && (iw % p.bwd_stride) == 0;
                ih /= p.bwd_stride;
                iw /= p.bwd_stride;
            }
            if (sok && a_b[i] >= 0 && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W && c < p.Ck) {
                const float* src = p.x + (((int64_t)a_b[i] * p.H + ih) * p.W + iw) * p.ldx + c;
                if (c + 3 < p.Ck) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    v.x = src[0];
                    if (c + 1 < p.Ck) v.y = src[1];
                    if (c + 2 < p.Ck) v.z = src[2];
                }
            }
            ra[i] = v;
        }
        // B
        if constexpr (!BWD) {
            // rows k = c0 + kr (Cin index), cols n0 + nq*4 ..   W[(wt*Cin + c)*Cout + n]
#pragma unroll
            for (int i = 0; i < B_F4; ++i) {
                const int e = tid + i * kThreads;          // float4 index in the BK x BN tile
                const int kr = e / (BN / 4), nq = e % (BN / 4);
                const int c = c0 + kr, n = n0 + nq * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < B_TOTAL && c < p.Cin && n < p.Cout) {
                    const float* src = p.w + ((int64_t)wt * p.Cin + c) * p.Cout + n;
                    if (w_vec) {
                        v = *reinterpret_cast<const float4*>(src);
                    } else {
                        v.x = src[0];
                        if (n + 1 < p.Cout) v.y = src[1];
                        if (n + 2 < p.Cout) v.z = src[2];
                        if (n + 3 < p.Cout) v.w = src[3];
                    }
                }
                rb[i] = v;
            }
        } else {
            // output column = conv Cin index (n0 + col), k = conv Cout index (c0 + kq*4), contiguous in memory
#pragma unroll
            for (int i = 0; i < B_F4; ++i) {
                const int e = tid + i * kThreads;
                const int col = (e >> 6) * RPW + (e & (RPW - 1)), kq = (e & 63) / RPW;
                const int cin = n0 + col, k = c0 + kq * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < B_TOTAL && cin < p.Cin && k < p.Cout) {
                    const float* src = p.w + ((int64_t)wt * p.Cin + cin) * p.Cout + k;
                    if (w_vec && k + 3 < p.Cout) {
                        v = *reinterpret_cast<const float4*>(src);
                    } else {
                        v.x = src[0];
                        if (k + 1 < p.Cout) v.y = src[1];
                        if (k + 2 < p.Cout) v.z = src[2];
                        if (k + 3 < p.Cout) v.w = src[3];
                    }
                }
                rb[i] = v;
            }
        }
    };

    auto store_tiles = [&](Regs& R) {
        float4 (&ra)[A_F4] = R.a;
        float4 (&rb)[B_F4] = R.b;
        const unsigned okmask = R.ok;
        const int st_c0 = R.c0;
        if constexpr (VEC) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!BWD && p.in_scale) {
                // producer BatchNorm + activation on load (st_c0: the channel base of the step these registers hold)
                const int c = st_c0 + a_kq * 4;
                if (c < p.Ck) {
                    const float4 sc = *reinterpret_cast<const float4*>(p.in_scale + c), sf = *reinterpret_cast<const float4*>(p.in_shift + c);
#pragma unroll
                    for (int i = 0; i < A_F4; ++i) {
                        ra[i].x = epi_act(fmaf(ra[i].x, sc.x, sf.x), p.in_act); ra[i].y = epi_act(fmaf(ra[i].y, sc.y, sf.y), p.in_act);
                        ra[i].z = epi_act(fmaf(ra[i].z, sc.z, sf.z), p.in_act); ra[i].w = epi_act(fmaf(ra[i].w, sc.w, sf.w), p.in_act);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < A_F4; ++i)
                if (!((okmask >> i) & 1u)) ra[i] = z;
#pragma unroll
            for (int i = 0; i < B_F4; ++i)
                if (!((okmask >> (8 + i)) & 1u)) rb[i] = z;
        }
#pragma unroll
        for (int i = 0; i < A_F4; ++i)
            *reinterpret_cast<float4*>(As + (a_rl + i * RPP) * PITCH_A + a_kq * 4) = ra[i];
        if constexpr (!BWD) {
#pragma unroll
            for (int i = 0; i < B_F4; ++i) {
                const int e = tid + i * kThreads;
                const int kr = e / (BN / 4), nq = e % (BN / 4);
                if (e < B_TOTAL) *reinterpret_cast<float4*>(Bs + kr * PITCH_B + nq * 4) = rb[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < B_F4; ++i) {
                const int e = tid + i * kThreads;
                if (e < B_TOTAL) *reinterpret_cast<float4*>(Bs + ((e >> 6) * RPW + (e & (RPW - 1))) * PITCH_B + ((e & 63) / RPW) * 4) = rb[i];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // Prefetch distance 1 (measured: a second register set with the loads of step k+2 in flight as well was 6 %
    // SLOWER — 90 vs 96 TF on the SegmentHead conv — so global latency is not what limits this kernel).
    const int ks_beg = p.splits > 1 ? (int)blockIdx.y * p.ks_per_split : 0;
    const int ks_end = p.splits > 1 ? (ks_beg + p.ks_per_split < nk ? ks_beg + p.ks_per_split : nk) : nk;
    load_tiles(ks_beg, R0);
    for (int ks = ks_beg; ks < ks_end; ++ks) {
        store_tiles(R0);
        __syncthreads();
        if (ks + 1 < ks_end) load_tiles(ks + 1, R0);
        if constexpr (BKT > BK && TM * TN >= 4) {
            // large wave tile with a deep K step: fragments of 16 K values at a time (all of them up front would cost
            // 32 more VGPRs per 16 K values and a wave per SIMD of occupancy)
#pragma unroll
            for (int s = 0; s < BKT / BK; ++s)
                mma_step<TM, TN, true, BWD, PITCH_A, PITCH_B, BK>(As + s * BK, BWD ? Bs + s * BK : Bs + s * BK * PITCH_B,
                                                                 wm * TM * 32, wn * TN * 32, acc);
        } else {
            mma_step<TM, TN, true, BWD, PITCH_A, PITCH_B, BKT>(As, Bs, wm * TM * 32, wn * TN * 32, acc);
        }
        __syncthreads();
    }

    if constexpr (BM == 128 && BN == 32 && WM == 4 && WN == 1 && VEC && BKT == BK) {
        if (p.bn.part) {                                           // conv -> training BatchNorm in this launch (conv_epilogue_bn / _bwd)
            __syncthreads();                                       // the tiles are scratch from here on
            static_assert(BM * PITCH_A >= BnScratch<32, 4>::kFloats, "the exchange scratch fits the A tile");
            if constexpr (BWD) conv_epilogue_bn_bwd<TM, TN, WM, WN>(p, acc, m0, n0, wm, wn, mt, As, tag_issue(p.bn.sync));
            else               conv_epilogue_bn<TM, TN, WM, WN>(p, acc, m0, n0, wm, wn, mt, As, tag_issue(p.bn.sync));
            return;
        }
    }
    conv_epilogue<TM, TN>(p, acc, m0, n0, wm, wn);
}

// ---- LDS-DMA variant of the 128x128 kernel (VEC operands only) ----------------------------------------------
// Same tiles, same MFMA order and the same results as conv_igemm_kernel<128,128,2,2,BWD,true>, but the operand tiles go
// global -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave instruction) into a THREE-stage ring: no staging
// registers, no ds_write pass, the loads of steps k+1 and k+2 in flight behind the MFMAs of step k, ONE barrier per
// K step (the register-staged kernel has two and a prefetch distance of one).  Conventions:
//   * LDS-DMA writes lane-linearly (wave-uniform base + lane*16), so the tiles are unpadded; bank conflicts are avoided
//     by a swizzle applied on the SOURCE side (which element a lane fetches) and again in the fragment reads:
//       MK / NK form [128 rows][4 quads]:  quad q of row r sits at slot r*4 + (q ^ ((r >> 2) & 3))
//       KN form      [16 k][128 n]      :  row k is rotated by 32*((k >> 2) & 1) floats
//   * out-of-range elements (padding taps, ragged channel tails, rows past M) are fetched from a 16-byte zero word;
//   * ordering: each wave waits for ITS pieces with a counted s_waitcnt vmcnt, then one s_barrier publishes the stage
//     (MI355X_MICROARCH.md item 7); the asm statements keep the DMA out of hipcc's own waitcnt bookkeeping.
__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void conv_glds16(const float* gsrc, uint32_t lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// two consecutive 1-KiB pieces (LDS destinations lds_dst and lds_dst + 1024) behind ONE M0 set-up: the instruction offset of
// the second load moves both its LDS destination and its global address by 1024, so its pointer is passed 1024 bytes low
__device__ __forceinline__ void conv_glds16x2(const float* g0, const float* g1, uint32_t lds_dst)
{
    unsigned keep;
    const char* g1m = reinterpret_cast<const char*>(g1) - 1024;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %2, off offset:1024\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g0), "v"(g1m), "s"(lds_dst) : "memory");
}

template <int TM, int TN> struct DmaFrags { float a[2][TM][4], b[2][TN][4]; };

// BNF: the instantiation whose epilogue also finishes (forward) / runs the backward of (backward-data) a training BatchNorm -
// conv_epilogue_bn / conv_epilogue_bn_bwd.  A separate instantiation with a looser register budget: inside the plain kernel the extra
// epilogue pushed the 128-register kernels into scratch (124-176 bytes per lane) for every launch, fused or not.
template <int BM, int BN, bool BWD, bool BNF = false>
__global__ __launch_bounds__(kThreads, (BNF ? 3 : (BM * BN >= 128 * 128 ? 3 : 4))) void conv_igemm_dma_kernel(ConvParams p)
{
    constexpr int TM = BM / 64, TN = BN / 64, WN = 2, NSTAGE = 3;
    constexpr int PA = BM / 64, PB = BN / 64;                     // 1-KiB DMA pieces per wave and K step
    constexpr int NQB = BN / 4;                                   // quads per k row of the KN-form B tile
    using Frags = DmaFrags<TM, TN>;
    constexpr int STAGE_FLOATS = BM * BK + BK * BN;               // 4096 floats = 16 KiB
    __shared__ __attribute__((aligned(1024))) float smem[NSTAGE * STAGE_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int mt, nt;
    {
        const int ntn = p.n_tiles, nblk = gridDim.x;
        const int bid = blockIdx.x;
        const int per_xcd = nblk / 8;
        if (p.xcd_remap && per_xcd * 8 == nblk) {
            const int lin = (bid & 7) * per_xcd + (bid >> 3);
            mt = lin / ntn;
            nt = lin - mt * ntn;
        } else {
            mt = bid / ntn;
            nt = bid - mt * ntn;
        }
    }
    const int64_t m0 = (int64_t)mt * BM;
    const int n0 = nt * BN;
    unsigned tag0 = 0;                       // conv -> BatchNorm in one launch: this launch's exchange tag, requested now, used in the epilogue
    if constexpr (BNF) tag0 = tag_issue(p.bn.sync);
    const float* zero = g_zero16;
    asm volatile("" : "+v"(zero));          // keep the pointer in registers (hipcc re-derives it from the PC in every K step otherwise)
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);

    // ---- per-thread addressing, fixed for the whole K loop (host guarantees 32-bit element offsets, <= 32 taps, unit
    //      backward stride): A piece i of this wave = 16 rows x 4 quads, lane -> row = piece*16 + lane/4, slot lane&3
    int a_e0[PA], a_c[PA];
    unsigned a_vm[PA];            // bit t: tap t reads inside the image for this row
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = (wave * PA + i) * 16 + (lane >> 2);
        a_c[i] = ((lane & 3) ^ ((row >> 2) & 3)) * 4;            // logical channel offset fetched into this slot
        const int64_t m = m0 + row;
        a_e0[i] = 0;
        a_vm[i] = 0u;
        if (m < p.M) {
            const unsigned mu = (unsigned)m;
            const unsigned t = mu / (unsigned)p.Wo;
            const int ow = (int)(mu - t * (unsigned)p.Wo);
            const unsigned bb = t / (unsigned)p.Ho;
            const int oh = (int)(t - bb * (unsigned)p.Ho);
            const int ih0 = oh * p.stride, iw0 = ow * p.stride;
            a_e0[i] = (((int)bb * p.H + ih0) * p.W + iw0) * (int)p.ldx + a_c[i];
            for (int t2 = 0; t2 < p.taps.n; ++t2) {
                const int ih = ih0 + p.taps.dh[t2], iw = iw0 + p.taps.dw[t2];
                a_vm[i] |= ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) ? (1u << t2) : 0u;
            }
        }
    }
    int b_e[PB], b_k[PB];          // B piece i: element offset inside one (tap, chunk) slab, and the coordinate checked per step
    bool b_ok[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int P = (wave * PB + i) * 64 + lane;
        if constexpr (!BWD) {
            const int k = P / NQB, nq = ((P % NQB) - 8 * ((k >> 2) & 1)) & (NQB - 1);
            b_k[i] = k;                                           // checked against Cin with c0
            b_ok[i] = n0 + nq * 4 < p.Cout;
            b_e[i] = k * p.Cout + n0 + nq * 4;
        } else {
            const int row = P >> 2, kq = (P & 3) ^ ((row >> 2) & 3);
            b_k[i] = kq * 4;                                      // checked against Cout with c0
            b_ok[i] = n0 + row < p.Cin;
            b_e[i] = (n0 + row) * p.Cout + kq * 4;
        }
    }

    const int nchunk = (p.Ck + BK - 1) / BK;
    const int nk = p.taps.n * nchunk;
    const int ks_beg = p.splits > 1 ? (int)blockIdx.y * p.ks_per_split : 0;
    const int ks_end = p.splits > 1 ? (ks_beg + p.ks_per_split < nk ? ks_beg + p.ks_per_split : nk) : nk;
    const int n = ks_end - ks_beg;

    // per-tap uniform offsets in LDS (a dynamic index into the kernel-argument tap table would be a scalar load per step,
    // and scalar loads force lgkmcnt(0) waits that also drain the fragment reads)
    __shared__ int s_tap[32][2];
    if (tid < p.taps.n) {
        s_tap[tid][0] = (p.taps.dh[tid] * p.W + p.taps.dw[tid]) * (int)p.ldx + (p.multi ? p.tap_coff[tid] : 0);
        s_tap[tid][1] = p.multi ? p.tap_woff[tid] : p.taps.widx[tid] * p.Cin * p.Cout;
    }
    __syncthreads();
    // (tap, chunk) of the next step to issue, advanced incrementally (steps are issued strictly in order)
    int is_ti, is_ch;
    if (p.tap_inner) { is_ch = ks_beg / p.taps.n; is_ti = ks_beg - is_ch * p.taps.n; }
    else             { is_ti = ks_beg / nchunk;   is_ch = ks_beg - is_ti * nchunk; }
    int te_a = s_tap[is_ti][0], te_b = s_tap[is_ti][1];

    // ---- the DMA of one K step, in four slices (one 1-KiB piece each) so that they can sit between MFMA groups
    int st_ti = 0, st_c0 = 0, st_aoff = 0, st_boff = 0;
    uint32_t st_la = 0;
    auto issue_begin = [&](int stage) {
        st_ti = is_ti;
        st_c0 = is_ch * BK;
        st_aoff = __builtin_amdgcn_readfirstlane(te_a) + st_c0;
        st_boff = __builtin_amdgcn_readfirstlane(te_b) + (BWD ? st_c0 : st_c0 * p.Cout);
        st_la = lds0 + (uint32_t)(stage * STAGE_FLOATS * 4);
    };
    auto issue_a = [&](int i) {
        const bool ok = ((a_vm[i] >> st_ti) & 1u) && (st_c0 + a_c[i] < p.Ck);
        const float* src = p.x + (size_t)(unsigned)(a_e0[i] + st_aoff);       // >= 0 whenever ok (zero-extend: no sign fix-up)
        conv_glds16(ok ? src : zero, st_la + (uint32_t)((wave * PA + i) * 1024));
    };
    auto issue_b = [&](int i) {
        const bool ok = b_ok[i] && (st_c0 + b_k[i] < (BWD ? p.Cout : p.Cin));
        const float* src = p.w + (size_t)(unsigned)(b_e[i] + st_boff);
        conv_glds16(ok ? src : zero, st_la + (uint32_t)(BM * BK * 4 + (wave * PB + i) * 1024));
    };
    const bool tap_in = p.tap_inner != 0;
    const int inner_lim = tap_in ? p.taps.n : nchunk;
    auto a_src = [&](int i) -> const float* {
        const bool ok = ((a_vm[i] >> st_ti) & 1u) && (st_c0 + a_c[i] < p.Ck);
        return ok ? p.x + (size_t)(unsigned)(a_e0[i] + st_aoff) : zero;
    };
    auto b_src = [&](int i) -> const float* {
        const bool ok = b_ok[i] && (st_c0 + b_k[i] < (BWD ? p.Cout : p.Cin));
        return ok ? p.w + (size_t)(unsigned)(b_e[i] + st_boff) : zero;
    };
    auto issue_a_all = [&]() {
        if constexpr (PA == 2) conv_glds16x2(a_src(0), a_src(1), st_la + (uint32_t)(wave * PA * 1024));
        else issue_a(0);
    };
    auto issue_b_all = [&]() {
        if constexpr (PB == 2) conv_glds16x2(b_src(0), b_src(1), st_la + (uint32_t)(BM * BK * 4 + wave * PB * 1024));
        else issue_b(0);
    };
    auto issue_end = [&]() {      // advance to the following step and fetch its tap entry now (used one K step later)
        // branch-free (scalar selects): the inner counter wraps into the outer one
        int inner = (tap_in ? is_ti : is_ch) + 1, outer = tap_in ? is_ch : is_ti;
        const bool wrap = inner == inner_lim;
        inner = wrap ? 0 : inner;
        outer += wrap ? 1 : 0;
        is_ti = tap_in ? inner : outer;
        is_ch = tap_in ? outer : inner;
        const int tn = is_ti < p.taps.n ? is_ti : 0;
        te_a = s_tap[tn][0];
        te_b = s_tap[tn][1];
    };
    auto issue = [&](int stage) {
        issue_begin(stage);
#pragma unroll
        for (int i = 0; i < PA; ++i) issue_a(i);
#pragma unroll
        for (int i = 0; i < PB; ++i) issue_b(i);
        issue_end();
    };

    const int l31 = lane & 31, h = lane >> 5;
    const int swz = (l31 >> 2) & 3;
    auto read_a = [&](int stage, Frags& F, int q, int tm) {
        const float4* As4 = reinterpret_cast<const float4*>(smem + stage * STAGE_FLOATS);
        const int r = (wm * TM + tm) * 32 + l31;
        const float4 v = As4[r * 4 + ((2 * q + h) ^ swz)];
        F.a[q][tm][0] = v.x; F.a[q][tm][1] = v.y; F.a[q][tm][2] = v.z; F.a[q][tm][3] = v.w;
    };
    auto read_b = [&](int stage, Frags& F, int q, int tn) {
        const float* Bs = smem + stage * STAGE_FLOATS + BM * BK;
        const int c = (wn * TN + tn) * 32 + l31;
        if constexpr (BWD) {
            const float4 v = reinterpret_cast<const float4*>(Bs)[c * 4 + ((2 * q + h) ^ swz)];
            F.b[q][tn][0] = v.x; F.b[q][tn][1] = v.y; F.b[q][tn][2] = v.z; F.b[q][tn][3] = v.w;
        } else {
            const int cc = (c + 32 * h) & (BN - 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) F.b[q][tn][j] = Bs[(8 * q + 4 * h + j) * BN + cc];
        }
    };
    auto read_frags = [&](int stage, Frags& F) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int t = 0; t < TM; ++t) read_a(stage, F, q, t);
#pragma unroll
            for (int t = 0; t < TN; ++t) read_b(stage, F, q, t);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    auto mma4 = [&](const Frags& F, int q, int j) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(F.a[q][tm][j], F.b[q][tn][j], acc[tm][tn], 0, 0, 0);
    };

    // One K step.  The fragments of step k are already in `cur` (read one step ago), so its MFMAs start right behind the
    // barrier; the 32 MFMAs go out in eight groups of four and every group is followed by a SLICE of the step's other
    // work - two fragment reads of step k+1, or one DMA piece of step k+3 - which issues in the shadow of the group's
    // last MFMA (64 matrix-pipe cycles) instead of forming a separate phase in which the matrix pipe idles.
    // STEADY (compile time): the step is at least three steps from the end of the K loop, so every `if` below is known
    // to be taken - the steady-state loop body has no branches (a few runtime-uniform branches cost this kernel ~10 %)
    auto kstep = [&](auto steady_tag, auto stage_tag, int k, Frags& cur, Frags& nxt) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        constexpr int SK = decltype(stage_tag)::value;            // k % NSTAGE when known at compile time, else -1
        const bool rd = STEADY || k + 1 < n, dm = STEADY || k + 3 < n;
        const int sn = SK >= 0 ? (SK + 1) % NSTAGE : (k + 1) % NSTAGE;
        if (rd) {
            if (STEADY || k + 2 < n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PA + PB) : "memory");   // my pieces of step k+1 landed; k+2 flies on
            else           asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                        // everyone's pieces; stage k%3 is free
            asm volatile("" ::: "memory");
        }
        mma4(cur, 0, 0); __builtin_amdgcn_sched_barrier(0);
        if (rd) {
#pragma unroll
            for (int t = 0; t < TM; ++t) read_a(sn, nxt, 0, t);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma4(cur, 0, 1); __builtin_amdgcn_sched_barrier(0);
        if (rd) {
#pragma unroll
            for (int t = 0; t < TN; ++t) read_b(sn, nxt, 0, t);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma4(cur, 0, 2); __builtin_amdgcn_sched_barrier(0);
        if (rd) {
#pragma unroll
            for (int t = 0; t < TM; ++t) read_a(sn, nxt, 1, t);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma4(cur, 0, 3); __builtin_amdgcn_sched_barrier(0);
        if (rd) {
#pragma unroll
            for (int t = 0; t < TN; ++t) read_b(sn, nxt, 1, t);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma4(cur, 1, 0); __builtin_amdgcn_sched_barrier(0);
        if (dm) { issue_begin(SK >= 0 ? SK : k % NSTAGE); issue_a_all(); }
        __builtin_amdgcn_sched_barrier(0);
        mma4(cur, 1, 1); __builtin_amdgcn_sched_barrier(0);
        mma4(cur, 1, 2); __builtin_amdgcn_sched_barrier(0);
        if (dm) issue_b_all();
        __builtin_amdgcn_sched_barrier(0);
        mma4(cur, 1, 3); __builtin_amdgcn_sched_barrier(0);
        if (dm) issue_end();
    };

    Frags F0, F1;
    if (n > 0) issue(0);
    if (n > 1) issue(1);
    if (n > 2) issue(2);
    if (n > 0) {
        if (n > 2)      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (PA + PB)) : "memory");
        else if (n > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PA + PB) : "memory");
        else            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        read_frags(0, F0);
    }
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>; using S2 = std::integral_constant<int, 2>;
    using SR = std::integral_constant<int, -1>;
    int k = 0;
    for (; k + 8 < n; k += 6) {                                   // steady state, six steps: ring slots are compile-time constants
        kstep(std::true_type{}, S0{}, k, F0, F1);
        kstep(std::true_type{}, S1{}, k + 1, F1, F0);
        kstep(std::true_type{}, S2{}, k + 2, F0, F1);
        kstep(std::true_type{}, S0{}, k + 3, F1, F0);
        kstep(std::true_type{}, S1{}, k + 4, F0, F1);
        kstep(std::true_type{}, S2{}, k + 5, F1, F0);
    }
    for (; k + 4 < n; k += 2) {                                   // steady state: both steps have k + 3 < n
        kstep(std::true_type{}, SR{}, k, F0, F1);
        kstep(std::true_type{}, SR{}, k + 1, F1, F0);
    }
    for (; k < n; k += 2) {                                       // the last (up to four) steps
        kstep(std::false_type{}, SR{}, k, F0, F1);
        if (k + 1 < n) kstep(std::false_type{}, SR{}, k + 1, F1, F0);
    }
    if constexpr (BNF) {
        __syncthreads();                                           // the ring is scratch from here on
        if constexpr (BWD) conv_epilogue_bn_bwd<TM, TN, (BM / 32 / TM), WN>(p, acc, m0, n0, wm, wn, mt, smem, tag0);
        else               conv_epilogue_bn<TM, TN, (BM / 32 / TM), WN>(p, acc, m0, n0, wm, wn, mt, smem, tag0);
        return;
    }
    conv_epilogue<TM, TN>(p, acc, m0, n0, wm, wn);
}

// ---- deep-K pointwise convolutions on few rows: in-block split-K over wave-private LDS-DMA pipelines -----------------------------
// 960 -> 160, 960 -> 320, 1280 -> 256 and the backward-data of 160 -> 960 at 2048 rows (mobilenet_v2.py:56, aspp.py:73-75): the
// output has 64-192 64x64 tiles for 256 CUs, so the tiled kernel slices K over grid.y and pays a [splits][M][N] round trip plus a
// second launch (23-29 us per layer, of which ~5 are the reduce and its kernel boundary).  Here a block owns ONE (32*TM) x 32
// output tile and its four waves split K four ways (round 2's conv1x1_direct_kernel did that with per-lane row loads straight into
// fragment registers and lost above K = 640; this kernel supersedes it for every K >= 256) - each wave streams its K slice through a
// PRIVATE ring of LDS stages with coalesced 1-KiB LDS-DMA pieces (global_load_lds_dwordx4), NST-1 K steps of loads in flight per
// wave, no block barrier inside the K loop (a wave only waits for its own pieces with a counted s_waitcnt vmcnt), and the DMA
// pieces of step k+NST-1 are issued between the MFMA pairs of step k.  The four partial tiles meet in LDS once, every wave adds a
// quarter of the tile in wave order (deterministic) and stores it: split-K without the workspace and without the second launch.
//   A operand (activations / dY): MK form [32*TM rows][16 k], quad q of row r at slot r*4 + (q ^ ((r >> 2) & 3))  (as conv_igemm_dma_kernel)
//   B operand forward  (W[k][n], n contiguous): KN form [16 k][32 n], logical row k stored at physical row k ^ ((k >> 2) & 1) so that
//                      the two half-waves of a fragment read (k, k + 4) use different halves of the 64 banks
//   B operand backward (W[n][k], k contiguous): NK form [32 n][16 k], swizzled like A
// Measured (960 -> 160 at 2048 rows): 15.3-15.8 us whatever the tile / ring depth (23.6 for split-K + reduce, 26.7 for the direct
// kernel); floor of the design = 6.4 us of MFMA work on the busiest CU + launch, first round trip and the reduce / store tail.
template <int TM, int TN, int NST, bool BWD, bool AFF = false>
__global__ __launch_bounds__(kThreads, 1) void conv1x1_ksplit_dma_kernel(ConvParams p)
{
    // AFF: p.in_scale / in_shift / in_act (the producer's training BatchNorm + activation) applied to the A fragments as they
    // are read; the wave keeps the scale / shift entries of ITS K slice in a private LDS table (<= 2 x 512 floats)
    constexpr int TBL = AFF ? 512 : 0;
    constexpr int A_FL = TM * 32 * BK, B_FL = BK * TN * 32, ST_FL = A_FL + B_FL;
    constexpr int PA = TM * 2, PB = TN * 2, NP = PA + PB;    // 1-KiB pieces per wave and K step
    constexpr int WAVE_FL = NST * ST_FL;
    constexpr int BN = TN * 32;
    static_assert(WAVE_FL >= TM * TN * 16 * 64, "the final reduction reuses a wave's ring");
    static_assert(NP <= 8 && (NST - 2) * NP < 64, "pieces are issued behind the eight MFMA groups; vmcnt is a 6-bit counter");
    __shared__ __attribute__((aligned(1024))) float smem[4 * WAVE_FL + 4 * 2 * TBL + 4];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = blockIdx.x / p.n_tiles, nt = blockIdx.x - mt * p.n_tiles;
    const int64_t m0 = (int64_t)mt * (32 * TM);
    const int n0 = nt * BN;
    unsigned tag0 = 0;                                       // conv -> BatchNorm in one launch: the exchange tag, requested now
    if constexpr (!AFF) { if (p.bn.part) tag0 = tag_issue(p.bn.sync); }
    const float* zero = g_zero16;
    asm volatile("" : "+v"(zero));
    float* wsm = smem + wave * WAVE_FL;
    const uint32_t lds_w = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)wsm);

    // ---- fixed per-lane source addressing --------------------------------------------------------------------------
    int a_off[PA], a_c[PA];                                  // element offset of this lane's row (< 0: the row reads zeros), channel offset
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = i * 16 + (lane >> 2);
        a_c[i] = ((lane & 3) ^ ((row >> 2) & 3)) * 4;
        a_off[i] = -1;
        const int64_t m = m0 + row;
        if (m < p.M) {
            const unsigned mu = (unsigned)m;
            const unsigned t = mu / (unsigned)p.Wo;
            const int ow = (int)(mu - t * (unsigned)p.Wo);
            const unsigned bb = t / (unsigned)p.Ho;
            const int oh = (int)(t - bb * (unsigned)p.Ho);
            const int ih = oh + p.taps.dh[0], iw = ow + p.taps.dw[0];
            if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
                a_off[i] = (((int)bb * p.H + ih) * p.W + iw) * (int)p.ldx + a_c[i];
        }
    }
    int b_off[PB], b_k[PB];
    bool b_ok[PB];
    const float* wbase = p.w + (int64_t)p.taps.widx[0] * p.Cin * p.Cout;
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        if constexpr (!BWD) {
            // KN form [16 k][BN n]: a piece is 64 lanes x 16 B = (256 / BN) k rows of BN floats
            constexpr int QN = BN / 4;                       // quads per k row
            const int P = i * 64 + lane;
            const int kp = P / QN, k = kp ^ ((kp >> 2) & 1), nq = P % QN;
            b_k[i] = k;                                      // checked against Cin with the step's channel base
            b_ok[i] = n0 + nq * 4 < p.Cout;
            b_off[i] = k * p.Cout + n0 + nq * 4;
        } else {
            const int row = i * 16 + (lane >> 2), kq = (lane & 3) ^ ((row >> 2) & 3);
            b_k[i] = kq * 4;                                 // checked against Cout
            b_ok[i] = n0 + row < p.Cin;
            b_off[i] = (n0 + row) * p.Cout + kq * 4;
        }
    }

    // K steps of this wave
    const int nsteps = (p.Ck + BK - 1) / BK;
    const int per = (nsteps + 3) / 4;
    const int s_beg = wave * per;
    const int n = (s_beg + per < nsteps ? s_beg + per : nsteps) - s_beg;      // may be <= 0

    auto a_src = [&](int i, int c0) -> const float* {
        const bool ok = a_off[i] >= 0 && c0 + a_c[i] < p.Ck;
        return ok ? p.x + (size_t)(unsigned)(a_off[i] + c0) : zero;
    };
    auto b_src = [&](int i, int c0) -> const float* {
        const bool ok = b_ok[i] && c0 + b_k[i] < (BWD ? p.Cout : p.Cin);
        return ok ? wbase + (size_t)(unsigned)(b_off[i] + (BWD ? c0 : c0 * p.Cout)) : zero;
    };
    int is_stage = 0, is_c0 = s_beg * BK;                    // ring slot / channel base of the next step to issue
    auto issue_piece = [&](int piece) {                      // piece 0..PA-1: A, PA..NP-1: B
        const uint32_t st = lds_w + (uint32_t)(is_stage * ST_FL * 4);
        if (piece < PA) conv_glds16(a_src(piece, is_c0), st + (uint32_t)(piece * 1024));
        else            conv_glds16(b_src(piece - PA, is_c0), st + (uint32_t)(A_FL * 4 + (piece - PA) * 1024));
    };
    auto issue_advance = [&]() {
        is_stage = is_stage + 1 == NST ? 0 : is_stage + 1;
        is_c0 += BK;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float* tbl = smem + 4 * WAVE_FL + wave * 2 * TBL;        // [0, TBL): scale, [TBL, 2 TBL): shift of channels s_beg*16 ...
    if constexpr (AFF) {
        for (int i = lane; i < per * BK; i += 64) {
            const int c = s_beg * BK + i;
            tbl[i] = c < p.Ck ? p.in_scale[c] : 0.0f;
            tbl[TBL + i] = c < p.Ck ? p.in_shift[c] : 0.0f;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // before the counted DMA waits start; wave-private: no barrier
    }
    int tbl_k = 0;                                           // table offset of the step whose fragments are read next

    float fa[2][TM][4], fb[2][TN][4];
    auto read_frags = [&](int stage) {
        const float* As = wsm + stage * ST_FL;
        const float* Bs = As + A_FL;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int r = tm * 32 + l31;
                const float4 v = reinterpret_cast<const float4*>(As)[r * 4 + ((2 * q + h) ^ ((r >> 2) & 3))];
                fa[q][tm][0] = v.x; fa[q][tm][1] = v.y; fa[q][tm][2] = v.z; fa[q][tm][3] = v.w;
            }
            if constexpr (AFF) {
                const float4 sc = *reinterpret_cast<const float4*>(tbl + tbl_k + 8 * q + 4 * h);
                const float4 sf = *reinterpret_cast<const float4*>(tbl + TBL + tbl_k + 8 * q + 4 * h);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    fa[q][tm][0] = epi_act(fmaf(fa[q][tm][0], sc.x, sf.x), p.in_act);
                    fa[q][tm][1] = epi_act(fmaf(fa[q][tm][1], sc.y, sf.y), p.in_act);
                    fa[q][tm][2] = epi_act(fmaf(fa[q][tm][2], sc.z, sf.z), p.in_act);
                    fa[q][tm][3] = epi_act(fmaf(fa[q][tm][3], sc.w, sf.w), p.in_act);
                }
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int c = tn * 32 + l31;
                if constexpr (BWD) {
                    const float4 v = reinterpret_cast<const float4*>(Bs)[c * 4 + ((2 * q + h) ^ ((c >> 2) & 3))];
                    fb[q][tn][0] = v.x; fb[q][tn][1] = v.y; fb[q][tn][2] = v.z; fb[q][tn][3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[q][tn][j] = Bs[((8 * q + 4 * h + j) ^ h) * BN + c];
                }
            }
        }
    };
    auto mma = [&](int q, int j) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][tm][j], fb[q][tn][j], acc[tm][tn], 0, 0, 0);
    };

    // ---- the K loop: NST-1 steps of DMA in flight, the pieces of step k+NST-1 issued between the MFMA groups of step k ----------
    if (n > 0) {
#pragma unroll
        for (int s = 0; s < NST - 1; ++s)
            if (s < n) {
#pragma unroll
                for (int i = 0; i < NP; ++i) issue_piece(i);
                issue_advance();
            }
        int k = 0, rd = 0;
        for (; k + NST - 1 < n; ++k) {                      // steady state: steps k .. k+NST-2 are in flight
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 2) * NP) : "memory");
            read_frags(rd);
            rd = rd + 1 == NST ? 0 : rd + 1;
            tbl_k += BK;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 8; ++g) {                   // 8 MFMA groups (one per (q, j)); the NP pieces go out behind the first ones
                mma(g >> 2, g & 3);
                __builtin_amdgcn_sched_barrier(0);
                if (g < NP) issue_piece(g);
                __builtin_amdgcn_sched_barrier(0);
            }
            issue_advance();
        }
        for (; k < n; ++k) {                                // the last NST-1 steps: nothing left to issue
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            read_frags(rd);
            rd = rd + 1 == NST ? 0 : rd + 1;
            tbl_k += BK;
#pragma unroll
            for (int g = 0; g < 8; ++g) mma(g >> 2, g & 3);
        }
    }

    // ---- the four partial tiles meet in LDS; wave w adds accumulator registers 4w..4w+3 (rows 8w + 4h + 0..3 of every 32-row
    //      MFMA block) in wave order and stores them.  (A wave writes into ITS ring, which only it has been reading: no barrier
    //      in front of the stores.)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) wsm[((tm * TN + tn) * 16 + r) * 64 + lane] = acc[tm][tn][r];
    __syncthreads();
    if constexpr (!BWD && !AFF) {
        if (p.bn.part) {
            // conv -> training BatchNorm in this launch (see conv_epilogue_bn): this lane's 4 * TM rows of every column tile, summed over
            // the four K slices in wave order, stay in registers across the exchange
            constexpr int BNT = TN * 32;
            float vv[TM][TN][4];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int e = ((tm * TN + tn) * 16 + wave * 4 + rr) * 64 + lane;
                        float v = smem[0 * WAVE_FL + e];
                        v += smem[1 * WAVE_FL + e];
                        v += smem[2 * WAVE_FL + e];
                        v += smem[3 * WAVE_FL + e];
                        vv[tm][tn][rr] = v;
                    }
            __syncthreads();                                 // every partial tile has been read: the rings are scratch now
            const BnScratch<BNT, 4> S(smem);
            static_assert(4 * WAVE_FL >= BnScratch<BNT, 4>::kFloats, "the exchange scratch fits the rings");
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int64_t m = m0 + tm * 32 + rr + 8 * wave + 4 * h;
                        const float v = m < p.M ? vv[tm][tn][rr] : 0.0f;
                        s1 += v;
                        s2 = fmaf(v, v, s2);
                    }
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (h == 0) {
                    S.colsum[(0 * 4 + wave) * BNT + tn * 32 + l31] = s1;
                    S.colsum[(1 * 4 + wave) * BNT + tn * 32 + l31] = s2;
                }
            }
            bn_block_finish<BNT, 4>(p, S, n0, mt, tag0);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int c = tn * 32 + l31, n_col = n0 + c;
                if (n_col >= p.Cn) continue;
                const float sc = S.aff[c], sf = S.aff[BNT + c];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int64_t m = m0 + tm * 32 + rr + 8 * wave + 4 * h;
                        if (m < p.M) {
                            const float v = vv[tm][tn][rr];
                            p.y[m * p.ldy + n_col] = v;
                            float o = fmaf(v, sc, sf);
                            if (p.bn.res) o += p.bn.res[m * p.bn.ldr + n_col];
                            p.bn.y[m * p.bn.ldy + n_col] = epi_act(o, p.bn.act);
                        }
                    }
            }
            return;
        }
    }
    if constexpr (BWD && !AFF) {
        if (p.bn.part) {
            // backward-data + the BatchNorm backward of the layer in front, in this launch (see conv_epilogue_bn_bwd): the lane's rows of
            // the gradient tile, summed over the four K slices in wave order, and the BatchNorm's input at the same elements stay in
            // registers across the exchange
            constexpr int BNT = TN * 32;
            const BnTrain& bn = p.bn;
            float vv[TM][TN][4], xs[TM][TN][4];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int e = ((tm * TN + tn) * 16 + wave * 4 + rr) * 64 + lane;
                        float v = smem[0 * WAVE_FL + e];
                        v += smem[1 * WAVE_FL + e];
                        v += smem[2 * WAVE_FL + e];
                        v += smem[3 * WAVE_FL + e];
                        vv[tm][tn][rr] = v;
                    }
            __syncthreads();                                 // every partial tile has been read: the rings are scratch now
            const BnScratch<BNT, 4> S(smem);
            static_assert(4 * WAVE_FL >= BnScratch<BNT, 4>::kFloats, "the exchange scratch fits the rings");
            float mu_[TN], is_[TN], ga_[TN];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int n_col = n0 + tn * 32 + l31;
                const bool ok = n_col < p.Cn;
                const float mu = ok ? bn.mean[n_col] : 0.0f, is = ok ? bn.invstd[n_col] : 0.0f, ga = ok ? bn.gamma[n_col] : 0.0f,
                            be = ok ? bn.beta[n_col] : 0.0f;
                mu_[tn] = mu; is_[tn] = is; ga_[tn] = ga;
                const float zsc = ga * is, zsf = be - mu * zsc;
                float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int64_t m = m0 + tm * 32 + rr + 8 * wave + 4 * h;
                        const bool in = ok && m < p.M;
                        const float x = in ? bn.bx[m * bn.ldbx + n_col] : 0.0f;
                        xs[tm][tn][rr] = x;
                        float u = in ? vv[tm][tn][rr] : 0.0f;
                        if (in && bn.gin) u += bn.gin[m * bn.ldgin + n_col];
                        if (bn.act != 0) u *= act_mask(fmaf(x, zsc, zsf), bn.act);
                        vv[tm][tn][rr] = u;
                        s1 += u;
                        s2 = fmaf(u, (x - mu) * is, s2);
                    }
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (h == 0) {
                    S.colsum[(0 * 4 + wave) * BNT + tn * 32 + l31] = s1;
                    S.colsum[(1 * 4 + wave) * BNT + tn * 32 + l31] = s2;
                }
            }
            bn_block_finish<BNT, 4, true>(p, S, n0, mt, tag0);
            const float inv_count = 1.0f / (float)p.M;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int c = tn * 32 + l31, n_col = n0 + c;
                if (n_col >= p.Cn) continue;
                const float db = S.aff[c], dg = S.aff[BNT + c];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int64_t m = m0 + tm * 32 + rr + 8 * wave + 4 * h;
                        if (m < p.M) {
                            if (bn.dres) bn.dres[m * bn.lddr + n_col] = vv[tm][tn][rr];
                            bn.y[m * bn.ldy + n_col] = bn_dx(vv[tm][tn][rr], xs[tm][tn][rr], mu_[tn], is_[tn], ga_[tn], db, dg, inv_count);
                        }
                    }
            }
            return;
        }
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n_col = n0 + tn * 32 + l31;
        if (n_col >= p.Cn) continue;
        const float bv = p.bias ? p.bias[n_col] : 0.0f;
        const bool affine = p.epi.gamma != nullptr;
        float sc = 1.0f, sf = 0.0f;
        if (affine) {
            const float is = 1.0f / sqrtf(p.epi.var[n_col] + p.epi.eps);
            sc = p.epi.gamma[n_col] * is;
            sf = p.epi.beta[n_col] - p.epi.mean[n_col] * sc;
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = wave * 4 + rr;
                const int e = ((tm * TN + tn) * 16 + r) * 64 + lane;
                float v = smem[0 * WAVE_FL + e];
                v += smem[1 * WAVE_FL + e];
                v += smem[2 * WAVE_FL + e];
                v += smem[3 * WAVE_FL + e];
                const int64_t m = m0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < p.M) {
                    float o = v + bv;
                    if (affine) o = fmaf(o, sc, sf);
                    if (p.epi.res) o += p.epi.res[m * p.epi.ldr + n_col];
                    if (p.accumulate) o += p.y[m * p.ldy + n_col];
                    p.y[m * p.ldy + n_col] = epi_act(o, p.epi.act);
                }
            }
    }
}

// Backward-data of the pointwise expand convolutions at 1/2 and 1/4 resolution (mobilenet_v2.py:48: 16 -> 96 on 4 x 128 x 256 pixels,
// 24 -> 144 on 4 x 64 x 128): dx[m][0..CN) = dy[m][0..Ck) . W^T with CN = 16 / 24 / 32 - a read of dy (50 MB for the first) with 6 KiB of
// weights.  As 128 x 32 MFMA tiles the launch took 57 us (1 TB/s): sixteen-channel K steps read 64-byte pieces of every row, one step
// in flight.  Here a block takes 64 rows WHOLE (their Ck floats are contiguous: the tile is one coalesced stream, every load of the
// block in flight at once), parks them in LDS, and wave w computes output channels [w CN/4, (w+1) CN/4) of row `lane` on the VALU
// with the weights as wave-uniform (scalar) operands; the 64 x CN result leaves through LDS as full rows.
// BWD == false: the forward of the narrow-OUTPUT pointwise layers (project convolutions 32 -> 16, 96 -> 24, 144 -> 24 / 32 on the same
// maps): the same kernel with the weights read as W[k][n].
template <int CN, int MAXL, bool BWD>                       // MAXL float4 loads per thread: 64 rows x Ck / 4 / 256 threads = Ck / 16, rounded up
__global__ __launch_bounds__(256) void conv1x1_rows_kernel(ConvParams p, unsigned inv_kq)
{
    constexpr int CPW = CN / 4;                             // output channels per wave
    extern __shared__ __attribute__((aligned(16))) float rows_smem[];
    __shared__ int srcoff[64];                              // element offset of the rows' source pixels (< 0: outside the gradient map)
    const int pitch = p.Ck + 4;                             // floats per LDS row (16-byte aligned, rows on different banks)
    float* tile = rows_smem;                                // [64][pitch]
    float* outt = rows_smem + 64 * pitch;                   // [64][CN]
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int64_t m0 = (int64_t)blockIdx.x * 64;
    const int kq = p.Ck >> 2;                               // float4 per row
    // the wave's weights - CPW output channels x Ck - live in CPW x NRJ registers, lane l of register (j, r) holding k = 64 r + l;
    // the FMA loop broadcasts them with v_readlane (as scalar LOADS in that loop each K step waited for the scalar cache: with two
    // blocks per CU the 96 -> 24 / 144 -> 24 layers took 18 / 23 us).  backward: W[n][k] (k contiguous), forward: W[k][n]
    constexpr int NRJ = (MAXL * 16 + 63) / 64;
    float wreg[CPW][NRJ];
    {
        const float* wbase = p.w + (int64_t)p.taps.widx[0] * p.Cin * p.Cout;
#pragma unroll
        for (int j = 0; j < CPW; ++j)
#pragma unroll
            for (int r = 0; r < NRJ; ++r) {
                const int k = 64 * r + lane;
                const int kc = k < p.Ck ? k : 0;
                const float wv = BWD ? wbase[(int64_t)(wave * CPW + j) * p.Cout + kc] : wbase[(int64_t)kc * p.Cout + wave * CPW + j];
                wreg[j][r] = k < p.Ck ? wv : 0.0f;
            }
    }
    if (t < 64) {
        const int64_t m = m0 + t;
        int off = -1;
        if (m < p.M) {
            const unsigned mu = (unsigned)m;
            const unsigned tq = mu / (unsigned)p.Wo;
            const int ow = (int)(mu - tq * (unsigned)p.Wo);
            const unsigned bb = tq / (unsigned)p.Ho;
            const int oh = (int)(tq - bb * (unsigned)p.Ho);
            const int ih = oh + p.taps.dh[0], iw = ow + p.taps.dw[0];
            if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) off = (((int)bb * p.H + ih) * p.W + iw) * (int)p.ldx;
        }
        srcoff[t] = off;
    }
    __syncthreads();
    // every load of the block in flight at once: unconditional loads from clamped addresses, zero-fill where the tile is written
    const int total = 64 * kq;
    float4 v[MAXL];
    unsigned okm = 0;
#pragma unroll
    for (int i = 0; i < MAXL; ++i) {
        const int e = t + i * 256;
        const int ec = e < total ? e : 0;
        const int r = (int)(((unsigned)ec * inv_kq) >> 20), q = ec - r * kq;      // ec / kq, exact for ec < 3072 (host-checked)
        const int off = srcoff[r];
        const bool ok = e < total && off >= 0;
        v[i] = *reinterpret_cast<const float4*>(p.x + (ok ? (size_t)(unsigned)off + (size_t)(q * 4) : 0));
        okm |= ok ? (1u << i) : 0u;
    }
#pragma unroll
    for (int i = 0; i < MAXL; ++i) {
        const int e = t + i * 256;
        if (e < total) {
            const int r = (int)(((unsigned)e * inv_kq) >> 20), q = e - r * kq;
            *reinterpret_cast<float4*>(tile + r * pitch + q * 4) = ((okm >> i) & 1u) ? v[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    float acc[CPW];
#pragma unroll
    for (int j = 0; j < CPW; ++j) acc[j] = 0.0f;
    const float* trow = tile + lane * pitch;
#pragma unroll
    for (int r = 0; r < NRJ; ++r) {
        const int kend = p.Ck - 64 * r < 64 ? p.Ck - 64 * r : 64;
        for (int kk = 0; kk < kend; kk += 4) {
            const float4 x4 = *reinterpret_cast<const float4*>(trow + 64 * r + kk);
#pragma unroll
            for (int j = 0; j < CPW; ++j) {
                const int wv = __float_as_int(wreg[j][r]);
                acc[j] = fmaf(x4.x, __int_as_float(__builtin_amdgcn_readlane(wv, kk + 0)), acc[j]);
                acc[j] = fmaf(x4.y, __int_as_float(__builtin_amdgcn_readlane(wv, kk + 1)), acc[j]);
                acc[j] = fmaf(x4.z, __int_as_float(__builtin_amdgcn_readlane(wv, kk + 2)), acc[j]);
                acc[j] = fmaf(x4.w, __int_as_float(__builtin_amdgcn_readlane(wv, kk + 3)), acc[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < CPW; ++j) outt[lane * CN + wave * CPW + j] = acc[j];
    __syncthreads();
    for (int e = t; e < 64 * (CN / 4); e += 256) {
        const int r = e / (CN / 4), q = e - r * (CN / 4);
        const int64_t m = m0 + r;
        if (m >= p.M) continue;
        float4 o = *reinterpret_cast<const float4*>(outt + r * CN + q * 4);
        float* dst = p.y + m * p.ldy + q * 4;
        if (p.accumulate) {
            const float4 a = *reinterpret_cast<const float4*>(dst);
            o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        }
        *reinterpret_cast<float4*>(dst) = o;
    }
}

// Forward of the narrow-INPUT pointwise layers (the expand convolutions 16 -> 96, 24 -> 144, 32 -> 192 on the 1/2- and 1/4-resolution
// maps, fixed padding folded in): y[m][0..CN) = x[m][0..CK) . W, a WRITE of y (51 MB for the first) with CK x CN weights.  As 128 x 128
// MFMA tiles with one K step the launch took 48 us.  Here: 64 rows per block, wave w computes the CN / 4 output channels
// [w CN/4, (w+1) CN/4) of row `lane` on the VALU (weights as scalar operands), the 64 x CN tile leaves through LDS as full rows.
// BWD == true: the backward-data of the narrow-OUTPUT layers on the same maps (project 96 -> 24 / 144 -> 24: dy has 24 channels, dx 96 /
// 144), the same kernel with the weights read as W[n][k] (k contiguous).
template <int CK, int CN, bool BWD = false>
__global__ __launch_bounds__(256) void conv1x1_fwd_widen_kernel(ConvParams p)
{
    constexpr int NPW = CN / 4, XP = CK + 4, OP = CN + 4;   // outputs per wave, LDS pitches (rows on different banks)
    static_assert(NPW % 4 == 0 && CK % 4 == 0, "float4 pieces");
    __shared__ __attribute__((aligned(16))) float xt[64 * XP];
    __shared__ __attribute__((aligned(16))) float ot[64 * OP];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int64_t m0 = (int64_t)blockIdx.x * 64;
    constexpr int KQ = CK / 4;
    // backward: the wave's NPW x CK weights (one contiguous run of W[n][k]) lane-distributed over NR registers, broadcast with v_readlane
    // in the FMA loop: no memory access there (as scalar loads per output channel the 144-wide form stalled on the scalar cache: 34 us)
    constexpr int NR = BWD ? (CK * NPW + 63) / 64 : 1;
    float wreg[NR];
    if constexpr (BWD) {
        const float* wbase = p.w + (int64_t)p.taps.widx[0] * p.Cin * p.Cout + (int64_t)(wave * NPW) * p.Cout;   // W[n0 + j][k]: one run of NPW * CK floats
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int f = r * 64 + lane;
            wreg[r] = wbase[f < CK * NPW ? f : 0];
        }
    }
    for (int e = t; e < 64 * KQ; e += 256) {
        const int r = e / KQ, q = e - r * KQ;
        const int64_t m = m0 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < p.M) {
            const unsigned mu = (unsigned)m;
            const unsigned tq = mu / (unsigned)p.Wo;
            const int ow = (int)(mu - tq * (unsigned)p.Wo);
            const unsigned bb = tq / (unsigned)p.Ho;
            const int oh = (int)(tq - bb * (unsigned)p.Ho);
            const int ih = oh + p.taps.dh[0], iw = ow + p.taps.dw[0];
            if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
                v = *reinterpret_cast<const float4*>(p.x + (((int64_t)bb * p.H + ih) * p.W + iw) * p.ldx + q * 4);
        }
        *reinterpret_cast<float4*>(xt + r * XP + q * 4) = v;
    }
    __syncthreads();
    float acc[NPW];
#pragma unroll
    for (int j = 0; j < NPW; ++j) acc[j] = 0.0f;
    float xr[CK];
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(xt + lane * XP + q * 4);
        xr[q * 4 + 0] = v.x; xr[q * 4 + 1] = v.y; xr[q * 4 + 2] = v.z; xr[q * 4 + 3] = v.w;
    }
    if constexpr (BWD) {
#pragma unroll
        for (int j = 0; j < NPW; ++j)
#pragma unroll
            for (int k = 0; k < CK; ++k) {
                const int f = j * CK + k;                                    // compile-time after unrolling
                acc[j] = fmaf(xr[k], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wreg[f / 64]), f % 64)), acc[j]);
            }
    } else {
        // forward: W[k][n], n contiguous - a wave's NPW weights of a k are one run: scalar operands (measured against the v_readlane form:
        // 16 -> 96 14.1 vs 20.5 us, 24 -> 144 14.6 vs 15.5 us)
        const float* __restrict__ wr = p.w + (int64_t)p.taps.widx[0] * p.Cin * p.Cout + wave * NPW;
#pragma unroll
        for (int k = 0; k < CK; ++k) {
            const float* wk = wr + (int64_t)k * p.Cout;
#pragma unroll
            for (int j = 0; j < NPW; ++j) acc[j] = fmaf(xr[k], wk[j], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < NPW; j += 4)
        *reinterpret_cast<float4*>(ot + lane * OP + wave * NPW + j) = make_float4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]);
    __syncthreads();
    constexpr int NQ = CN / 4;
    for (int e = t; e < 64 * NQ; e += 256) {
        const int r = e / NQ, q = e - r * NQ;
        const int64_t m = m0 + r;
        if (m < p.M) {
            float4 o = *reinterpret_cast<const float4*>(ot + r * OP + q * 4);
            float* dst = p.y + m * p.ldy + q * 4;
            if (BWD && p.accumulate) {
                const float4 a = *reinterpret_cast<const float4*>(dst);
                o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
            }
            *reinterpret_cast<float4*>(dst) = o;
        }
    }
}

// Forward of the MobileNetV2 stem (mobilenet_v2.py:7-12: Conv2d(3, 32, 3, stride 2, padding 1, bias=False) on an even-sized image with
// packed pixels): 6 MB in, 17 MB out.  As 128 x 32 MFMA tiles with scalar (non-vector: Cin = 3) operand loads the launch took 27 us.
// Here: 64 output pixels per block; wave w computes output channels [8w, 8w + 8) of pixel `lane`; the 27 taps of a pixel are three runs
// of nine contiguous floats (the weight gradient's wgrad_stem3x3s2_kernel reads them the same way), the 27 x 8 weights of a wave are
// scalar operands; the 64 x 32 tile leaves through LDS as full 128-byte rows.
__global__ __launch_bounds__(256) void conv_stem3x3s2_fwd_kernel(ConvParams p)
{
    __shared__ __attribute__((aligned(16))) float ot[64 * 36];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int64_t m0 = (int64_t)blockIdx.x * 64, m = m0 + lane;
    float xv[27];
    {
        const bool rv = m < p.M;
        const unsigned mu = (unsigned)(rv ? m : 0);
        const unsigned tq = mu / (unsigned)p.Wo;
        const int ow = (int)(mu - tq * (unsigned)p.Wo);
        const unsigned bb = tq / (unsigned)p.Ho;
        const int oh = (int)(tq - bb * (unsigned)p.Ho);
        const bool left = ow > 0;
#pragma unroll
        for (int th = 0; th < 3; ++th) {
            const int ih = oh * 2 - 1 + th;
            const bool ok = rv && (unsigned)ih < (unsigned)p.H;
            const float* px = p.x + (ok ? (((int64_t)bb * p.H + ih) * p.W + ow * 2) * 3 : 3);
            const StemF3 a = *reinterpret_cast<const StemF3*>(px - ((ok && left) ? 3 : 0));
            const StemF4 b = *reinterpret_cast<const StemF4*>(px);
            const StemF2 c = *reinterpret_cast<const StemF2*>(px + 4);
            const bool oka = ok && left;
            xv[th * 9 + 0] = oka ? a.x : 0.0f; xv[th * 9 + 1] = oka ? a.y : 0.0f; xv[th * 9 + 2] = oka ? a.z : 0.0f;
            xv[th * 9 + 3] = ok ? b.x : 0.0f; xv[th * 9 + 4] = ok ? b.y : 0.0f; xv[th * 9 + 5] = ok ? b.z : 0.0f;
            xv[th * 9 + 6] = ok ? b.w : 0.0f; xv[th * 9 + 7] = ok ? c.x : 0.0f; xv[th * 9 + 8] = ok ? c.y : 0.0f;
        }
    }
    const float* __restrict__ wr = p.w + wave * 8;          // W[j][n], j = (th * 3 + tw) * 3 + c, n contiguous (HWIO)
    float acc[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[n] = 0.0f;
#pragma unroll
    for (int j = 0; j < 27; ++j) {
        const float* wj = wr + j * 32;
#pragma unroll
        for (int n = 0; n < 8; ++n) acc[n] = fmaf(xv[j], wj[n], acc[n]);
    }
    *reinterpret_cast<float4*>(ot + lane * 36 + wave * 8) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4*>(ot + lane * 36 + wave * 8 + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    __syncthreads();
    for (int e = t; e < 64 * 8; e += 256) {
        const int r = e >> 3, q = e & 7;
        if (m0 + r < p.M) *reinterpret_cast<float4*>(p.y + (m0 + r) * p.ldy + q * 4) = *reinterpret_cast<const float4*>(ot + r * 36 + q * 4);
    }
}

// Tile / ring choice of conv1x1_ksplit_dma_kernel, from the measured table (profiles/r03_conv1x1_ksplit.txt; all six shapes x six
// (TM, TN, NST) candidates at 2048 rows): the ring depth does not matter (3 vs 6 stages: +-0.5 us - the K loop is bound by the
// MFMA chain of ONE wave per SIMD, not by bytes in flight), the tile does through the grid: 64x32 tiles while the grid stays within
// ~one block per CU (960 -> 160: 160 blocks, 1280 -> 256: 256), 32x64 for the backward-data form (its B tile is row-major like A),
// 32x32 tiles and two blocks per CU beyond that (960 -> 320: 640 blocks, 22.6 vs 24.5 us).
struct KsplitCfg { int tm, tn, nst; };
static KsplitCfg ksplit_choose(int64_t M, int Cn, bool bwd, int force)
{
    static const KsplitCfg cand[] = {{1, 1, 5}, {2, 1, 3}, {1, 2, 3}};
    if (force >= 1 && force <= 3) return cand[force - 1];
    const KsplitCfg c = bwd ? cand[2] : cand[1];
    if (cdiv(M, 32 * c.tm) * cdiv(Cn, 32 * c.tn) > 288) return cand[0];
    return c;
}

// split-K second stage: y[m][n] = epilogue(bias[n] + sum_z part[z][m][n]), z in fixed order (deterministic)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* part, int splits, int64_t M, int Cn,
                                                            const float* bias, float* y, int64_t ldy, Epilogue epi, int accumulate)
{
    const int64_t MN = M * Cn;
    if ((Cn & 3) == 0 && (ldy & 3) == 0 && epi.gamma == nullptr && epi.res == nullptr && epi.act == 0) {
        // training form (bias only): 16-byte accesses
        const int cq = Cn >> 2;
        const int64_t total = M * cq;
        for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
            const int64_t m = e / cq;
            const int q = (int)(e - m * cq);
            float4 s = bias ? *reinterpret_cast<const float4*>(bias + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float* src = part + m * Cn + q * 4;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            int z = 0;
            for (; z + 4 <= splits; z += 4) {          // four slices in flight, added in slice order (same sums as one at a time)
                const float4 v0 = *reinterpret_cast<const float4*>(src + (int64_t)z * MN);
                const float4 v1 = *reinterpret_cast<const float4*>(src + (int64_t)(z + 1) * MN);
                const float4 v2 = *reinterpret_cast<const float4*>(src + (int64_t)(z + 2) * MN);
                const float4 v3 = *reinterpret_cast<const float4*>(src + (int64_t)(z + 3) * MN);
                a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
                a.x += v1.x; a.y += v1.y; a.z += v1.z; a.w += v1.w;
                a.x += v2.x; a.y += v2.y; a.z += v2.z; a.w += v2.w;
                a.x += v3.x; a.y += v3.y; a.z += v3.z; a.w += v3.w;
            }
            for (; z < splits; ++z) {
                const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)z * MN);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            if (accumulate) {
                const float4 o = *reinterpret_cast<const float4*>(y + m * ldy + q * 4);
                s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
            }
            *reinterpret_cast<float4*>(y + m * ldy + q * 4) = s;
        }
        return;
    }
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < MN; e += (int64_t)gridDim.x * 256) {
        const int64_t m = e / Cn;
        const int n = (int)(e - m * Cn);
        float s = 0.0f;
        for (int z = 0; z < splits; ++z) s += part[(int64_t)z * MN + e];
        if (bias) s += bias[n];
        if (epi.gamma) {
            const float is = 1.0f / sqrtf(epi.var[n] + epi.eps);
            const float sc = epi.gamma[n] * is;
            s = fmaf(s, sc, epi.beta[n] - epi.mean[n] * sc);
        }
        if (epi.res) s += epi.res[m * epi.ldr + n];
        if (accumulate) s += y[m * ldy + n];
        y[m * ldy + n] = epi_act(s, epi.act);
    }
}

// split-K second stage in front of a training BatchNorm: the same sums in the same order as splitk_reduce_kernel's vector
// path, plus the column sums / sums of squares of the rows this block owns -> stats[blockIdx.x][2][Cn].
// Threads: q = tid % cq (channel quad, fixed per thread so that its sums stay in registers), rl = tid / cq (row lane).
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(const float* part, int splits, int64_t M, int Cn, const float* bias,
                                                                  float* y, int64_t ldy, int accumulate, float* stats, int64_t rows_per_block)
{
    __shared__ float4 sh[2][256];
    const int cq = Cn >> 2;
    const int nrl = 256 / cq;
    const int tid = threadIdx.x;
    const int q = tid % cq, rl = tid / cq;
    const bool active = rl < nrl;
    const int64_t MN = M * Cn;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    float4 t1 = make_float4(0.f, 0.f, 0.f, 0.f), t2 = t1;
    if (active) {
        const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int64_t m = r0 + rl; m < r1; m += nrl) {
            const float* src = part + m * Cn + q * 4;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            int z = 0;
            for (; z + 4 <= splits; z += 4) {
                const float4 v0 = *reinterpret_cast<const float4*>(src + (int64_t)z * MN);
                const float4 v1 = *reinterpret_cast<const float4*>(src + (int64_t)(z + 1) * MN);
                const float4 v2 = *reinterpret_cast<const float4*>(src + (int64_t)(z + 2) * MN);
                const float4 v3 = *reinterpret_cast<const float4*>(src + (int64_t)(z + 3) * MN);
                a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
                a.x += v1.x; a.y += v1.y; a.z += v1.z; a.w += v1.w;
                a.x += v2.x; a.y += v2.y; a.z += v2.z; a.w += v2.w;
                a.x += v3.x; a.y += v3.y; a.z += v3.z; a.w += v3.w;
            }
            for (; z < splits; ++z) {
                const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)z * MN);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
            float4 s = bv;
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            if (accumulate) {
                const float4 o = *reinterpret_cast<const float4*>(y + m * ldy + q * 4);
                s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
            }
            *reinterpret_cast<float4*>(y + m * ldy + q * 4) = s;
            t1.x += s.x; t1.y += s.y; t1.z += s.z; t1.w += s.w;
            t2.x = fmaf(s.x, s.x, t2.x); t2.y = fmaf(s.y, s.y, t2.y); t2.z = fmaf(s.z, s.z, t2.z); t2.w = fmaf(s.w, s.w, t2.w);
        }
    }
    sh[0][tid] = t1;
    sh[1][tid] = t2;
    __syncthreads();
    if (active && rl == 0) {
        for (int l = 1; l < nrl; ++l) {                    // fixed order over the row lanes
            const float4 a = sh[0][l * cq + q], b = sh[1][l * cq + q];
            t1.x += a.x; t1.y += a.y; t1.z += a.z; t1.w += a.w;
            t2.x += b.x; t2.y += b.y; t2.z += b.z; t2.w += b.w;
        }
        *reinterpret_cast<float4*>(stats + ((int64_t)blockIdx.x * 2 + 0) * Cn + q * 4) = t1;
        *reinterpret_cast<float4*>(stats + ((int64_t)blockIdx.x * 2 + 1) * Cn + q * 4) = t2;
    }
}

// ---- weight gradient ---------------------------------------------------------------------------------------
// GEMM rows = Cin (c), cols = Cout (n), reduction = output pixels m of one split; one tap per blockIdx.z/..
struct WgradParams {
    const float* x;    // forward input activations
    const float* dy;   // output gradient
    float* part;       // [splits][ntaps_live][Cin][Cout] partial sums
    int64_t ldx, lddy;
    int B, H, W, Ho, Wo, Cin, Cout, stride;
    int64_t M;
    int64_t m_per_split;
    int pointwise;      // 1x1, stride 1, pad 0: input pixel == output pixel, no index decode
    int xcd_remap;
    float* bias_part;   // optional [splits][Cout]: column sums of dy (the bias gradient), taken by the blocks of tile column 0
    // split-K reduction folded into this launch (wgrad_fold_tail): arrival counters of the launch's tiles (zero before and after), the
    // final gradient and the slice count; counters == NULL: the partial sums are all this launch leaves (a reduce launch follows)
    int* counters; float* dw; int splits;
    ConvTaps taps;
};

// Arrival counters of the weight-gradient launches that reduce their own split-K slices: a block adds one to its tile's counter when its
// partial tile is out; the block that finds splits - 1 there is the last of the tile, re-arms the counter and sums the tile's slices in
// SLICE ORDER (the order wgrad_reduce4_kernel used: bit-identical, whichever block arrives last).  The words are zero at module load and
// zero again after every launch; launches take disjoint ranges of the ring in turn (host: wgrad_counters_take), so launches that overlap on
// two queues never share a word.
constexpr int kWgradCounters = 1 << 16;
__device__ int g_wgrad_counters[kWgradCounters];

// A partial sum of a launch that reduces its own slices leaves the block as an agent-scope (sc1, write-through) store and is read back by
// the tile's last block with agent-scope loads: the slices cross XCDs, whose L2s are not coherent for ordinary accesses, and the
// alternative - ordinary stores + a release fence per block - is a buffer_wbl2 of the whole L2 per block (measured: the DeepLab step 5.26
// -> 6.90 ms, FPNSeg 19.4 -> 26.3 ms).  With every data access agent-scope, s_waitcnt vmcnt(0) in front of the arrival is the release.
__device__ __forceinline__ void wgrad_put(const WgradParams& p, float* dst, float v)
{
#ifdef PP_DEBUG_KNOBS
    if (p.counters) { __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
#endif
    *dst = v;
}

template <int BM, int BN>
__device__ __forceinline__ void wgrad_fold_tail(const WgradParams& p, int ti, int ctile, int ctiles, int ntile, int c0, int n0)
{
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // my partial tile has been written through before my arrival is counted
    __syncthreads();
    if (threadIdx.x == 0) {
        int* ctr = p.counters + ((int64_t)ti * ctiles + ctile) * gridDim.y + ntile;
        const int seen = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = seen == p.splits - 1;
        if (last) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    const int64_t cn = (int64_t)p.Cin * p.Cout, stride = (int64_t)p.taps.n * cn;
    const float* src0 = p.part + (int64_t)ti * cn;
    float* dst0 = p.dw + (int64_t)p.taps.widx[ti] * cn;
    constexpr int NQ = BN / 4;
    auto ld4 = [](const float* q) -> float4 {                // four agent-scope loads (past this XCD's L2)
        float4 v;
        v.x = __hip_atomic_load(q + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.y = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.z = __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.w = __hip_atomic_load(q + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return v;
    };
    for (int e = threadIdx.x; e < BM * NQ; e += kThreads) {
        const int r = e / NQ, q = e - r * NQ;
        const int c = c0 + r, n = n0 + q * 4;
        if (c >= p.Cin || n >= p.Cout) continue;
        const float* src = src0 + (int64_t)c * p.Cout + n;
        float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
        int k = 0;
        for (; k + 4 <= p.splits; k += 4) {                  // four slices in flight, added in slice order
            const float4 a = ld4(src + (int64_t)k * stride), b = ld4(src + (int64_t)(k + 1) * stride);
            const float4 cc = ld4(src + (int64_t)(k + 2) * stride), d = ld4(src + (int64_t)(k + 3) * stride);
            sacc.x += a.x; sacc.y += a.y; sacc.z += a.z; sacc.w += a.w;
            sacc.x += b.x; sacc.y += b.y; sacc.z += b.z; sacc.w += b.w;
            sacc.x += cc.x; sacc.y += cc.y; sacc.z += cc.z; sacc.w += cc.w;
            sacc.x += d.x; sacc.y += d.y; sacc.z += d.z; sacc.w += d.w;
        }
        for (; k < p.splits; ++k) {
            const float4 a = ld4(src + (int64_t)k * stride);
            sacc.x += a.x; sacc.y += a.y; sacc.z += a.z; sacc.w += a.w;
        }
        *reinterpret_cast<float4*>(dst0 + (int64_t)c * p.Cout + n) = sacc;
    }
}

template <int BM, int BN, int WM, int WN, bool VEC>
__global__ __launch_bounds__(kThreads) void conv_wgrad_kernel(WgradParams p)
{
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int PITCH_A = BM + 4, PITCH_B = BN + 4;
    constexpr int A_F4 = BM * BK / 4 / kThreads, B_F4 = BN * BK / 4 / kThreads;
    static_assert(A_F4 >= 1 && B_F4 >= 1, "tile too small");

    __shared__ __attribute__((aligned(16))) float As[BK * PITCH_A];
    __shared__ __attribute__((aligned(16))) float Bs[BK * PITCH_B];

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ctiles = (p.Cin + BM - 1) / BM;
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (one L2 each).  All (tap, c-tile, n-tile) blocks
    // of one pixel split read the same x / dy rows, so XCD j is given a contiguous range of SPLITS with all their tiles
    // (measured FETCH_SIZE of the SegmentHead weight gradient: 11x its algorithmic input with the plain order).
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_remap) {
        const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        if ((total & 7u) == 0) {
            const unsigned lin = bx + gx * (by + gy * bz);
            const unsigned nl = (lin & 7u) * (total >> 3) + (lin >> 3);
            bx = nl % gx;
            const unsigned t2 = nl / gx;
            by = t2 % gy;
            bz = t2 / gy;
        }
    }
    const int c0 = (int)(bx % (unsigned)ctiles) * BM;
    const int ti = (int)(bx / (unsigned)ctiles);            // live tap index
    const int n0 = (int)by * BN;
    const int split = (int)bz;
    const int64_t m_beg = (int64_t)split * p.m_per_split;
    const int64_t m_end = m_beg + p.m_per_split < p.M ? m_beg + p.m_per_split : p.M;
    const int dh = p.taps.dh[ti], dw = p.taps.dw[ti];
    const bool dy_vec = (p.lddy % 4 == 0) && (p.Cout % 4 == 0);
    const bool x_vec = (p.ldx % 4 == 0) && (p.Cin % 4 == 0);

    float4 ra[A_F4], rb[B_F4];
    unsigned okmask = 0;
    const bool want_bias = p.bias_part != nullptr && bx == 0;
    float4 bsum[B_F4];
#pragma unroll
    for (int i = 0; i < B_F4; ++i) bsum[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    auto load_tiles = [&](int64_t mb) {
        if constexpr (VEC) {   // unconditional float4 loads from clamped addresses; zero-fill happens at the LDS write
            okmask = 0;
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                const int e = tid + i * kThreads;
                const int kr = e / (BM / 4), cq = e % (BM / 4);
                const int64_t m = mb + kr;
                const int c = c0 + cq * 4;
                bool ok = m < m_end && c < p.Cin;
                int64_t pix = m;
                if (!p.pointwise) {
                    const unsigned mu = (unsigned)(ok ? m : 0);
                    const unsigned t = mu / (unsigned)p.Wo;
                    const int ow = (int)(mu - t * (unsigned)p.Wo);
                    const unsigned b = t / (unsigned)p.Ho;
                    const int oh = (int)(t - b * (unsigned)p.Ho);
                    const int ih = oh * p.stride + dh, iw = ow * p.stride + dw;
                    ok = ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                    pix = ((int64_t)b * p.H + ih) * p.W + iw;
                }
                ra[i] = *reinterpret_cast<const float4*>(p.x + (ok ? pix * p.ldx + c : 0));
                okmask |= ok ? (1u << i) : 0u;
            }
#pragma unroll
            for (int i = 0; i < B_F4; ++i) {
                const int e = tid + i * kThreads;
                const int kr = e / (BN / 4), nq = e % (BN / 4);
                const int64_t m = mb + kr;
                const int n = n0 + nq * 4;
                const bool ok = m < m_end && n < p.Cout;
                rb[i] = *reinterpret_cast<const float4*>(p.dy + (ok ? m * p.lddy + n : 0));
                okmask |= ok ? (1u << (8 + i)) : 0u;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int e = tid + i * kThreads;
            const int kr = e / (BM / 4), cq = e % (BM / 4);
            const int64_t m = mb + kr;
            const int c = c0 + cq * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < m_end && c < p.Cin) {
                int64_t pix = m;
                bool inb = true;
                if (!p.pointwise) {
                    const unsigned mu = (unsigned)m;          // M < 2^31 (checked on the host)
                    const unsigned t = mu / (unsigned)p.Wo;
                    const int ow = (int)(mu - t * (unsigned)p.Wo);
                    const unsigned b = t / (unsigned)p.Ho;
                    const int oh = (int)(t - b * (unsigned)p.Ho);
                    const int ih = oh * p.stride + dh, iw = ow * p.stride + dw;
                    inb = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                    pix = ((int64_t)b * p.H + ih) * p.W + iw;
                }
                if (inb) {
                    const float* src = p.x + pix * p.ldx + c;
                    if (x_vec) {
                        v = *reinterpret_cast<const float4*>(src);
                    } else {
                        v.x = src[0];
                        if (c + 1 < p.Cin) v.y = src[1];
                        if (c + 2 < p.Cin) v.z = src[2];
                        if (c + 3 < p.Cin) v.w = src[3];
                    }
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int e = tid + i * kThreads;
            const int kr = e / (BN / 4), nq = e % (BN / 4);
            const int64_t m = mb + kr;
            const int n = n0 + nq * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < m_end && n < p.Cout) {
                const float* src = p.dy + m * p.lddy + n;
                if (dy_vec) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    v.x = src[0];
                    if (n + 1 < p.Cout) v.y = src[1];
                    if (n + 2 < p.Cout) v.z = src[2];
                    if (n + 3 < p.Cout) v.w = src[3];
                }
            }
            rb[i] = v;
        }
    };
    auto store_tiles = [&]() {
        if constexpr (VEC) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < A_F4; ++i)
                if (!((okmask >> i) & 1u)) ra[i] = z;
#pragma unroll
            for (int i = 0; i < B_F4; ++i)
                if (!((okmask >> (8 + i)) & 1u)) rb[i] = z;
        }
        if (want_bias) {              // dy tile rows of this K-step (out-of-range rows / columns are zero by now)
#pragma unroll
            for (int i = 0; i < B_F4; ++i) {
                bsum[i].x += rb[i].x; bsum[i].y += rb[i].y; bsum[i].z += rb[i].z; bsum[i].w += rb[i].w;
            }
        }
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int e = tid + i * kThreads;
            *reinterpret_cast<float4*>(As + (e / (BM / 4)) * PITCH_A + (e % (BM / 4)) * 4) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int e = tid + i * kThreads;
            *reinterpret_cast<float4*>(Bs + (e / (BN / 4)) * PITCH_B + (e % (BN / 4)) * 4) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    if (m_beg < m_end) {
        load_tiles(m_beg);
        for (int64_t mb = m_beg; mb < m_end; mb += BK) {
            store_tiles();
            __syncthreads();
            if (mb + BK < m_end) load_tiles(mb + BK);
            mma_step<TM, TN, false, false, PITCH_A, PITCH_B>(As, Bs, wm * TM * 32, wn * TN * 32, acc);
            __syncthreads();
        }
    }

    if (want_bias) {
        // thread e = tid + i*256 staged row e / (BN/4), column quad e % (BN/4): fold the rows of each column quad (fixed order)
        constexpr int NQB = BN / 4;
        float4* red = reinterpret_cast<float4*>(As);            // the A tile is dead; kThreads float4 = 4 KiB fit in it
        float4 tsum = bsum[0];
#pragma unroll
        for (int i = 1; i < B_F4; ++i) { tsum.x += bsum[i].x; tsum.y += bsum[i].y; tsum.z += bsum[i].z; tsum.w += bsum[i].w; }
        red[tid] = tsum;
        __syncthreads();
        if (tid < NQB) {
            float4 t4 = red[tid];
            for (int g = 1; g < kThreads / NQB; ++g) {
                const float4 o = red[g * NQB + tid];
                t4.x += o.x; t4.y += o.y; t4.z += o.z; t4.w += o.w;
            }
            const int n = n0 + tid * 4;
            float* bp = p.bias_part + (int64_t)split * p.Cout;
            if (n + 0 < p.Cout) bp[n + 0] = t4.x;
            if (n + 1 < p.Cout) bp[n + 1] = t4.y;
            if (n + 2 < p.Cout) bp[n + 2] = t4.z;
            if (n + 3 < p.Cout) bp[n + 3] = t4.w;
        }
        __syncthreads();
    }
    const int lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    float* out = p.part + ((int64_t)split * p.taps.n + ti) * p.Cin * p.Cout;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + (wn * TN + tn) * 32 + l31;
        if (n >= p.Cout) continue;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = c0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (c < p.Cin) wgrad_put(p, out + (int64_t)c * p.Cout + n, acc[tm][tn][r]);
            }
    }
#ifdef PP_DEBUG_KNOBS
    if (p.counters) wgrad_fold_tail<BM, BN>(p, ti, (int)(bx % (unsigned)ctiles), ctiles, (int)by, c0, n0);
#endif
}

// (b, oh, ow) of an output pixel, advanced without divisions
struct RowIter {
    int ow, oh, bb;
    __device__ __forceinline__ void init(int64_t m, int Wo, int Ho)
    {
        const unsigned mu = (unsigned)m;
        const unsigned q = mu / (unsigned)Wo;
        ow = (int)(mu - q * (unsigned)Wo);
        bb = (int)(q / (unsigned)Ho);
        oh = (int)(q - (unsigned)bb * (unsigned)Ho);
    }
    __device__ __forceinline__ void next(int Wo, int Ho)
    {
        if (++ow == Wo) { ow = 0; if (++oh == Ho) { oh = 0; ++bb; } }
    }
    __device__ __forceinline__ void advance(int n, int Wo, int Ho)
    {
        ow += n;
        while (ow >= Wo) { ow -= Wo; if (++oh == Ho) { oh = 0; ++bb; } }
    }
};

// ---- LDS-DMA variant of the weight-gradient kernel (vector operands, no fused bias gradient) --------------------
// Same tiles / MFMA order / results as conv_wgrad_kernel<BM,BN,2,2,true>.  Both operand tiles are "k rows of contiguous
// channels" (x[pixel][c0 .. c0+BM), dy[pixel][n0 .. n0+BN)), i.e. exactly what the lane-linear LDS-DMA writes: a 1-KiB
// piece is 2 (128 wide) or 4 (64 wide) pixel rows.  Row k is rotated by 32*((k >> 2) & 1) floats on the source side so
// that the two half-waves of a fragment read (k and k+4) use different banks.  Pipeline and slicing as conv_igemm_dma_kernel:
// three stages, fragments of step k+1 read and the DMA of step k+3 issued between the MFMA groups of step k.
template <int BM, int BN>
__global__ __launch_bounds__(kThreads, (BM * BN >= 128 * 128 ? 3 : 4)) void conv_wgrad_dma_kernel(WgradParams p)
{
    constexpr int TM = BM / 64, TN = BN / 64, WN = 2, NSTAGE = 3;
    constexpr int PA = BM / 64, PB = BN / 64;
    constexpr int QA = BM / 4, QB = BN / 4;                       // quads per k row
    constexpr int STAGE_FLOATS = BK * BM + BK * BN;
    using Frags = DmaFrags<TM, TN>;
    __shared__ __attribute__((aligned(1024))) float smem[NSTAGE * STAGE_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ctiles = (p.Cin + BM - 1) / BM;
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_remap) {
        const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        if ((total & 7u) == 0) {
            const unsigned lin = bx + gx * (by + gy * bz);
            const unsigned nl = (lin & 7u) * (total >> 3) + (lin >> 3);
            bx = nl % gx;
            const unsigned t2 = nl / gx;
            by = t2 % gy;
            bz = t2 / gy;
        }
    }
    const int c0 = (int)(bx % (unsigned)ctiles) * BM;
    const int ti = (int)(bx / (unsigned)ctiles);
    const int n0 = (int)by * BN;
    const int split = (int)bz;
    const int64_t m_beg = (int64_t)split * p.m_per_split;
    const int64_t m_end = m_beg + p.m_per_split < p.M ? m_beg + p.m_per_split : p.M;
    const int dh = p.taps.dh[ti], dw = p.taps.dw[ti];
    const float* zero = g_zero16;
    asm volatile("" : "+v"(zero));          // keep the pointer in registers (hipcc re-derives it from the PC in every K step otherwise)
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
    const int n = m_beg < m_end ? (int)((m_end - m_beg + BK - 1) / BK) : 0;   // K steps of this split

    // per-lane row state: piece i of this wave covers k rows; this lane sits in row ka[i] / kb[i] of the 16-pixel step
    int ka[PA], ca[PA], kb[PB], nb[PB];
    bool ca_ok[PA], nb_ok[PB];
    RowIter ita[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int P = (wave * PA + i) * 64 + lane;
        ka[i] = P / QA;
        const int cq = ((P % QA) - 8 * ((ka[i] >> 2) & 1)) & (QA - 1);
        ca[i] = c0 + cq * 4;
        ca_ok[i] = ca[i] < p.Cin;
        if (!p.pointwise) ita[i].init(m_beg + ka[i] < p.M ? m_beg + ka[i] : 0, p.Wo, p.Ho);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int P = (wave * PB + i) * 64 + lane;
        kb[i] = P / QB;
        const int nq = ((P % QB) - 8 * ((kb[i] >> 2) & 1)) & (QB - 1);
        nb[i] = n0 + nq * 4;
        nb_ok[i] = nb[i] < p.Cout;
    }
    int64_t is_mb = m_beg;              // first pixel of the next step to issue
    uint32_t st_la = 0;
    auto a_src = [&](int i) -> const float* {
        const int64_t m = is_mb + ka[i];
        bool ok = m < m_end && ca_ok[i];
        int64_t pix = m;
        if (!p.pointwise) {
            const int ih = ita[i].oh * p.stride + dh, iw = ita[i].ow * p.stride + dw;
            ok = ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            pix = ((int64_t)ita[i].bb * p.H + ih) * p.W + iw;
            ita[i].advance(BK, p.Wo, p.Ho);
        }
        return ok ? p.x + (pix * p.ldx + ca[i]) : zero;
    };
    auto b_src = [&](int i) -> const float* {
        const int64_t m = is_mb + kb[i];
        const bool ok = m < m_end && nb_ok[i];
        return ok ? p.dy + (m * p.lddy + nb[i]) : zero;
    };
    auto issue_a = [&](int i) { conv_glds16(a_src(i), st_la + (uint32_t)((wave * PA + i) * 1024)); };
    auto issue_b = [&](int i) { conv_glds16(b_src(i), st_la + (uint32_t)(BK * BM * 4 + (wave * PB + i) * 1024)); };
    auto issue_a_all = [&]() {
        if constexpr (PA == 2) { const float* s0 = a_src(0); const float* s1 = a_src(1); conv_glds16x2(s0, s1, st_la + (uint32_t)(wave * PA * 1024)); }
        else issue_a(0);
    };
    auto issue_b_all = [&]() {
        if constexpr (PB == 2) { const float* s0 = b_src(0); const float* s1 = b_src(1); conv_glds16x2(s0, s1, st_la + (uint32_t)(BK * BM * 4 + wave * PB * 1024)); }
        else issue_b(0);
    };
    auto issue_begin = [&](int stage) { st_la = lds0 + (uint32_t)(stage * STAGE_FLOATS * 4); };
    auto issue_end = [&]() { is_mb += BK; };
    auto issue = [&](int stage) {
        issue_begin(stage);
#pragma unroll
        for (int i = 0; i < PA; ++i) issue_a(i);
#pragma unroll
        for (int i = 0; i < PB; ++i) issue_b(i);
        issue_end();
    };

    const int l31 = lane & 31, h = lane >> 5;
    auto read_a = [&](int stage, Frags& F, int q, int tm) {
        const float* As = smem + stage * STAGE_FLOATS;
        const int r = ((wm * TM + tm) * 32 + l31 + 32 * h) & (BM - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) F.a[q][tm][j] = As[(8 * q + 4 * h + j) * BM + r];
    };
    auto read_b = [&](int stage, Frags& F, int q, int tn) {
        const float* Bs = smem + stage * STAGE_FLOATS + BK * BM;
        const int c = ((wn * TN + tn) * 32 + l31 + 32 * h) & (BN - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) F.b[q][tn][j] = Bs[(8 * q + 4 * h + j) * BN + c];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    auto mma4 = [&](const Frags& F, int q, int j) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(F.a[q][tm][j], F.b[q][tn][j], acc[tm][tn], 0, 0, 0);
    };
    // STEADY (compile time): the step is at least three steps from the end of the K loop, so every `if` below is known
    // to be taken - the steady-state loop body has no branches (a few runtime-uniform branches cost this kernel ~10 %)
    auto kstep = [&](auto steady_tag, auto stage_tag, int k, Frags& cur, Frags& nxt) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        constexpr int SK = decltype(stage_tag)::value;            // k % NSTAGE when known at compile time, else -1
        const bool rd = STEADY || k + 1 < n, dm = STEADY || k + 3 < n;
        const int sn = SK >= 0 ? (SK + 1) % NSTAGE : (k + 1) % NSTAGE;
        if (rd) {
            if (STEADY || k + 2 < n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PA + PB) : "memory");
            else           asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        mma4(cur, 0, 0); __builtin_amdgcn_sched_barrier(0);
        if (rd) {
#pragma unroll
            for (int t = 0; t < TM; ++t) read_a(sn, nxt, 0, t);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma4(cur, 0, 1); __builtin_amdgcn_sched_barrier(0);
        if (rd) {
#pragma unroll
            for (int t = 0; t < TN; ++t) read_b(sn, nxt, 0, t);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma4(cur, 0, 2); __builtin_amdgcn_sched_barrier(0);
        if (rd) {
#pragma unroll
            for (int t = 0; t < TM; ++t) read_a(sn, nxt, 1, t);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma4(cur, 0, 3); __builtin_amdgcn_sched_barrier(0);
        if (rd) {
#pragma unroll
            for (int t = 0; t < TN; ++t) read_b(sn, nxt, 1, t);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma4(cur, 1, 0); __builtin_amdgcn_sched_barrier(0);
        if (dm) { issue_begin(SK >= 0 ? SK : k % NSTAGE); issue_a_all(); }
        __builtin_amdgcn_sched_barrier(0);
        mma4(cur, 1, 1); __builtin_amdgcn_sched_barrier(0);
        mma4(cur, 1, 2); __builtin_amdgcn_sched_barrier(0);
        if (dm) issue_b_all();
        __builtin_amdgcn_sched_barrier(0);
        mma4(cur, 1, 3); __builtin_amdgcn_sched_barrier(0);
        if (dm) issue_end();
    };

    Frags F0, F1;
    if (n > 0) issue(0);
    if (n > 1) issue(1);
    if (n > 2) issue(2);
    if (n > 0) {
        if (n > 2)      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (PA + PB)) : "memory");
        else if (n > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PA + PB) : "memory");
        else            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int t = 0; t < TM; ++t) read_a(0, F0, q, t);
#pragma unroll
            for (int t = 0; t < TN; ++t) read_b(0, F0, q, t);
        }
    }
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>; using S2 = std::integral_constant<int, 2>;
    using SR = std::integral_constant<int, -1>;
    int k = 0;
    for (; k + 8 < n; k += 6) {                                   // steady state, six steps: ring slots are compile-time constants
        kstep(std::true_type{}, S0{}, k, F0, F1);
        kstep(std::true_type{}, S1{}, k + 1, F1, F0);
        kstep(std::true_type{}, S2{}, k + 2, F0, F1);
        kstep(std::true_type{}, S0{}, k + 3, F1, F0);
        kstep(std::true_type{}, S1{}, k + 4, F0, F1);
        kstep(std::true_type{}, S2{}, k + 5, F1, F0);
    }
    for (; k + 4 < n; k += 2) {                                   // steady state: both steps have k + 3 < n
        kstep(std::true_type{}, SR{}, k, F0, F1);
        kstep(std::true_type{}, SR{}, k + 1, F1, F0);
    }
    for (; k < n; k += 2) {                                       // the last (up to four) steps
        kstep(std::false_type{}, SR{}, k, F0, F1);
        if (k + 1 < n) kstep(std::false_type{}, SR{}, k + 1, F1, F0);
    }

    const int hh = h;
    float* out = p.part + ((int64_t)split * p.taps.n + ti) * p.Cin * p.Cout;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int nn = n0 + (wn * TN + tn) * 32 + l31;
        if (nn >= p.Cout) continue;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = c0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (c < p.Cin) wgrad_put(p, out + (int64_t)c * p.Cout + nn, acc[tm][tn][r]);
            }
    }
#ifdef PP_DEBUG_KNOBS
    if (p.counters) wgrad_fold_tail<BM, BN>(p, ti, (int)(bx % (unsigned)ctiles), ctiles, (int)by, c0, n0);
#endif
}

// dW[widx[ti]][c][n] = sum_split part[split][ti][c][n]  (fixed order: deterministic); dead taps stay 0.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* part, int splits, int ntaps, int64_t cn,
                                                          ConvTaps taps, float* dw)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)ntaps * cn) return;
    const int ti = (int)(i / cn);
    const int64_t e = i - (int64_t)ti * cn;
    float s = 0.0f;
    for (int k = 0; k < splits; ++k) s += part[((int64_t)k * ntaps + ti) * cn + e];
    dw[(int64_t)taps.widx[ti] * cn + e] = s;
}

// float4 form (Cin*Cout % 4 == 0, 16-byte aligned buffers): four outputs per thread, the partials of four splits in flight
// at once, summed in split order (fixed order: deterministic).  The scalar form walks `splits` dependent 4-byte loads per
// thread: 66 us for the SegmentHead gradient (31 MB of partials) where the bytes need ~10.
template <typename WIdx>
__device__ __forceinline__ void wgrad_reduce4_body(int64_t blk, const float* part, int splits, int ntaps, int64_t cn, WIdx widx, float* dw)
{
    const int64_t cn4 = cn >> 2;
    const int64_t i = blk * 256 + threadIdx.x;
    if (i >= (int64_t)ntaps * cn4) return;
    const int ti = (int)(i / cn4);
    const int64_t e = (i - (int64_t)ti * cn4) * 4;
    const int64_t stride = (int64_t)ntaps * cn;
    const float* src = part + (int64_t)ti * cn + e;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    for (; k + 4 <= splits; k += 4) {
        const float4 a = *reinterpret_cast<const float4*>(src + (int64_t)k * stride);
        const float4 b = *reinterpret_cast<const float4*>(src + (int64_t)(k + 1) * stride);
        const float4 c = *reinterpret_cast<const float4*>(src + (int64_t)(k + 2) * stride);
        const float4 d = *reinterpret_cast<const float4*>(src + (int64_t)(k + 3) * stride);
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
        s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
        s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
    }
    for (; k < splits; ++k) {
        const float4 a = *reinterpret_cast<const float4*>(src + (int64_t)k * stride);
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    *reinterpret_cast<float4*>(dw + (int64_t)widx(ti) * cn + e) = s;
}
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const float* part, int splits, int ntaps, int64_t cn,
                                                           ConvTaps taps, float* dw)
{
    wgrad_reduce4_body((int64_t)blockIdx.x, part, splits, ntaps, cn, [&](int ti) { return taps.widx[ti]; }, dw);
}

// ---- weight gradient of the narrow layers ----------------------------------------------------------------------
// The first layers of both encoders and the classifier have 3..32 channels on one side (stem 3->32 / 3->64, MNv2
// 32->16, 16->96, 24->144, 96->24, 144->24/32, 192->32, classifier 256->19 / 128->19) and 33 K..524 K pixels.  As 64-
// or 128-wide MFMA tiles they run at 1-8 TF (3 of 64 rows live) and they are the LAST weight gradients of the backward
// pass, i.e. on the critical path of the join.  They are bandwidth problems: one read of x and dy.
//   lanes-over-Cout form: thread = output channel n, registers = the NTAPS*CIN input taps (x values are wave-uniform
//   loads), rows split over row lanes and blocks; partials [split][tap][c][n] feed the common fixed-order reduce.
// (row, column, image) of output pixel m, advanced incrementally: no integer division in the row loops

template <int CIN, int NTAPS, int U>
__global__ __launch_bounds__(256) void wgrad_narrow_in_kernel(WgradParams p, int NL, int RL, int64_t rows_per_split)
{
    const int t = threadIdx.x;
    const int n = t % NL, rl = t / NL;
    if (rl >= RL) return;
    const int64_t split = (int64_t)blockIdx.x * RL + rl;
    const int64_t m0 = split * rows_per_split;
    const int64_t m1 = m0 + rows_per_split < p.M ? m0 + rows_per_split : p.M;
    float acc[NTAPS * CIN];
#pragma unroll
    for (int j = 0; j < NTAPS * CIN; ++j) acc[j] = 0.0f;
    const bool live = n < p.Cout;
    const float* __restrict__ xg = p.x;
    const float* __restrict__ dyg = p.dy;
    // grid.y > 1: this block handles taps [tb, tb + NTAPS) of a larger kernel (the 7x7 stem: seven rows of seven taps,
    // 21 accumulators per thread instead of 147)
    const int tb = blockIdx.y * NTAPS;
    RowIter it;
    it.init(m0 < p.M ? m0 : 0, p.Wo, p.Ho);
    // U rows per trip: all their loads are issued before the first FMA (the rows are independent; one row at a time
    // exposes a full memory latency per row)
    for (int64_t m = m0; m < m1; m += U) {
        float g[U];
        float xv[U][NTAPS * CIN];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool rv = m + u < m1;
            g[u] = (live && rv) ? dyg[(m + u) * p.lddy + n] : 0.0f;
#pragma unroll
            for (int ti = 0; ti < NTAPS; ++ti) {
                const int ih = it.oh * p.stride + p.taps.dh[tb + ti], iw = it.ow * p.stride + p.taps.dw[tb + ti];
                const bool ok = rv && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                const float* xr = xg + (ok ? (((int64_t)it.bb * p.H + ih) * p.W + iw) * p.ldx : 0);
#pragma unroll
                for (int c = 0; c < CIN; ++c) {
                    const float v = xr[c];
                    xv[u][ti * CIN + c] = ok ? v : 0.0f;
                }
            }
            it.next(p.Wo, p.Ho);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < NTAPS * CIN; ++j) acc[j] = fmaf(xv[u][j], g[u], acc[j]);
    }
    if (live) {
        float* out = p.part + (split * (int64_t)p.taps.n + tb) * CIN * p.Cout + n;
#pragma unroll
        for (int j = 0; j < NTAPS * CIN; ++j) out[(int64_t)j * p.Cout] = acc[j];
    }
}

// The MobileNetV2 stem (mobilenet_v2.py:7-12: Conv2d(3, 32, 3, stride 2, padding 1)) is the LAST weight gradient of the backward pass:
// the optimiser waits for it and its reduce.  Its three input channels are contiguous, so the 3 x 3 taps of an output pixel are three
// runs of NINE contiguous floats (input row ih, pixels iw0-1 .. iw0+1): three wide loads per tap row (global loads need dword alignment
// only) instead of nine scalar ones - the generic kernel is bound by its 28 load instructions per output pixel - and the eight row lanes
// of a block meet in LDS before the partial sums leave it (one slice of partials per BLOCK: 1/8 of the reduce's input).
// W even, Wo = W / 2, pad 1: the right neighbour always exists, only iw0 - 1 (ow = 0) and ih (oh = 0) can fall outside.
template <int U>
__global__ __launch_bounds__(256) void wgrad_stem3x3s2_kernel(WgradParams p, int64_t rows_per_split)
{
    __shared__ float sh[8][27][33];
    const int t = threadIdx.x;
    const int n = t & 31, rl = t >> 5;
    const int64_t split = (int64_t)blockIdx.x * 8 + rl;
    const int64_t m0 = split * rows_per_split;
    const int64_t m1 = m0 + rows_per_split < p.M ? m0 + rows_per_split : p.M;
    float acc[27];
#pragma unroll
    for (int j = 0; j < 27; ++j) acc[j] = 0.0f;
    const bool live = n < p.Cout;
    const float* __restrict__ xg = p.x;
    const float* __restrict__ dyg = p.dy;
    RowIter it;
    it.init(m0 < p.M ? m0 : 0, p.Wo, p.Ho);
    for (int64_t m = m0; m < m1; m += U) {
        float g[U];
        float xv[U][27];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool rv = m + u < m1;
            g[u] = (live && rv) ? dyg[(m + u) * p.lddy + n] : 0.0f;
            const bool left = it.ow > 0;
#pragma unroll
            for (int th = 0; th < 3; ++th) {
                const int ih = it.oh * 2 - 1 + th;
                const bool ok = rv && (unsigned)ih < (unsigned)p.H;
                const float* px = xg + (ok ? (((int64_t)it.bb * p.H + ih) * p.W + it.ow * 2) * 3 : 3);   // pixel iw0 = 2 ow of row ih
                const StemF3 a = *reinterpret_cast<const StemF3*>(px - ((ok && left) ? 3 : 0));
                const StemF4 b = *reinterpret_cast<const StemF4*>(px);
                const StemF2 c = *reinterpret_cast<const StemF2*>(px + 4);
                const bool oka = ok && left;
                xv[u][th * 9 + 0] = oka ? a.x : 0.0f; xv[u][th * 9 + 1] = oka ? a.y : 0.0f; xv[u][th * 9 + 2] = oka ? a.z : 0.0f;
                xv[u][th * 9 + 3] = ok ? b.x : 0.0f; xv[u][th * 9 + 4] = ok ? b.y : 0.0f; xv[u][th * 9 + 5] = ok ? b.z : 0.0f;
                xv[u][th * 9 + 6] = ok ? b.w : 0.0f; xv[u][th * 9 + 7] = ok ? c.x : 0.0f; xv[u][th * 9 + 8] = ok ? c.y : 0.0f;
            }
            it.next(p.Wo, p.Ho);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 27; ++j) acc[j] = fmaf(xv[u][j], g[u], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 27; ++j) sh[rl][j][n] = acc[j];
    __syncthreads();
    float* out = p.part + (int64_t)blockIdx.x * 27 * p.Cout;           // [block][tap][c][n]
    for (int e = t; e < 27 * 32; e += 256) {
        const int j = e >> 5, nn = e & 31;
        float v = sh[0][j][nn];
#pragma unroll
        for (int r = 1; r < 8; ++r) v += sh[r][j][nn];                  // row lanes in order
        if (nn < p.Cout) out[(int64_t)j * p.Cout + nn] = v;
    }
}

// The ResNet stem (resnet_models.py:115-117: Conv2d(3, 64, 7, stride 2, padding 3)) is, like MobileNetV2's, the LAST weight gradient of
// the backward pass - the optimiser waits for it - and the generic kernel took 305 us on it at 4 x 256 x 512 (wgrad_narrow_in_kernel<3,7,4>
// over seven grid.y slices: every lane of a wave issued the same 21 vector loads per pixel).  Here the roles are turned round: a LANE is one
// of the 7 taps x 3 channels of a tap row (the 21 input values a pixel contributes to a tap row are contiguous in the packed image: input
// row 2 oh - 3 + th, pixels 2 ow - 3 .. 2 ow + 3 - one coalesced load), three tap rows share a wave (63 lanes), three waves a block (rows
// 0-2, 3-5, 6), and a thread keeps the 64 output channels of its (tap, channel) in registers.  dy[pixel][0..64) is the same for every lane:
// four aligned 16-dword SCALAR loads per pixel, entering the 64 FMAs as scalar operands.  One slice of partial sums per block
// ([49][3][Cout], the layout the reduce expects).  (First form, a lane per output channel with the 21 inputs as scalar loads: 203 us - the
// unaligned window became 21 single-dword scalar loads per pixel and wave.)
// (Measured variants of the loop: the input values of four pixels prefetched 114 us, the block's input patch staged in LDS 119 us, the
// FMAs as explicit v_fmac_f32 with scalar sources instead of the v_pk_fma_f32 pairs hipcc forms 202 us - this form 107 us: the loop is
// paced by the scalar cache delivering 256 bytes of dy per pixel and wave, not by the input loads.)
__global__ __launch_bounds__(192) void wgrad_stem7x7s2_kernel(WgradParams p, int64_t rows_per_block)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int th = wv * 3 + lane / 21, j = lane % 21;                       // tap row, (tap, channel) inside it
    const bool lane_on = lane < 63 && th < 7;
    const int64_t m0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t m1 = m0 + rows_per_block < p.M ? m0 + rows_per_block : p.M;
    float acc[64];
#pragma unroll
    for (int n = 0; n < 64; ++n) acc[n] = 0.0f;
    const float* __restrict__ xg = p.x;
    const float* __restrict__ dyg = p.dy;
    const int W3 = p.W * 3;
    RowIter it;
    it.init(m0 < p.M ? m0 : 0, p.Wo, p.Ho);
    for (int64_t m = m0; m < m1; ++m) {
        const int ih = it.oh * 2 - 3 + th, e = (it.ow * 2 - 3) * 3 + j;
        const bool ok = lane_on && (unsigned)ih < (unsigned)p.H && (unsigned)e < (unsigned)W3;
        const float xv = ok ? xg[((int64_t)it.bb * p.H + ih) * W3 + e] : 0.0f;
        const f32x16s* __restrict__ g = reinterpret_cast<const f32x16s*>(dyg + __builtin_amdgcn_readfirstlane((int)m) * p.lddy);   // wave-uniform: scalar loads
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x16s gq = g[q];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q * 16 + r] = fmaf(xv, gq[r], acc[q * 16 + r]);
        }
        it.next(p.Wo, p.Ho);
    }
    if (lane_on) {
        float* out = p.part + ((int64_t)blockIdx.x * 147 + th * 21 + j) * p.Cout;      // [block][tap = th*7 + tw][c][n], j = tw*3 + c
#pragma unroll
        for (int n = 0; n < 64; n += 4) *reinterpret_cast<float4*>(out + n) = make_float4(acc[n], acc[n + 1], acc[n + 2], acc[n + 3]);
    }
}

//   lanes-over-Cin form (1x1 convolutions with a narrow OUTPUT): thread = input channel c, registers = the COUT
//   outputs (dy values are wave-uniform loads).
template <int COUT, int U>
__global__ __launch_bounds__(256) void wgrad_narrow_out_kernel(WgradParams p, int NL, int RL, int64_t rows_per_split)
{
    const int t = threadIdx.x;
    const int c = t % NL, rl = t / NL;
    if (rl >= RL) return;
    const int64_t split = (int64_t)blockIdx.x * RL + rl;
    const int64_t m0 = split * rows_per_split;
    const int64_t m1 = m0 + rows_per_split < p.M ? m0 + rows_per_split : p.M;
    float acc[COUT], bacc[COUT];
#pragma unroll
    for (int j = 0; j < COUT; ++j) { acc[j] = 0.0f; bacc[j] = 0.0f; }
    const bool live = c < p.Cin;
    const bool want_bias = p.bias_part != nullptr && c == 0;      // the dy row is a wave-uniform load: lane 0 sums it
    const int dh = p.taps.dh[0], dw = p.taps.dw[0];
    const float* __restrict__ xg = p.x;
    const float* __restrict__ dyg = p.dy;
    RowIter it;
    it.init(m0 < p.M ? m0 : 0, p.Wo, p.Ho);
    for (int64_t m = m0; m < m1; m += U) {
        float xv[U];
        float g[U][COUT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool rv = m + u < m1;
            const int ih = it.oh * p.stride + dh, iw = it.ow * p.stride + dw;
            const bool ok = rv && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            const float v = xg[(ok ? (((int64_t)it.bb * p.H + ih) * p.W + iw) * p.ldx : 0) + (live ? c : 0)];
            xv[u] = (ok && live) ? v : 0.0f;
            const float* gr = dyg + (rv ? m + u : m0) * p.lddy;
#pragma unroll
            for (int j = 0; j < COUT; ++j) g[u][j] = gr[j];
            if (want_bias && rv) {
#pragma unroll
                for (int j = 0; j < COUT; ++j) bacc[j] += g[u][j];
            }
            it.next(p.Wo, p.Ho);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < COUT; ++j) acc[j] = fmaf(xv[u], g[u][j], acc[j]);
    }
    if (live) {
        float* out = p.part + split * (int64_t)p.Cin * COUT + (int64_t)c * COUT;
#pragma unroll
        for (int j = 0; j < COUT; ++j) out[j] = acc[j];
    }
    if (want_bias) {
        float* bp = p.bias_part + split * (int64_t)COUT;
#pragma unroll
        for (int j = 0; j < COUT; ++j) bp[j] = bacc[j];
    }
}

// Many-split reduce: 32 lanes per output add splits l, l+32, .. then a fixed LDS tree (deterministic).
template <typename WIdx>
__device__ __forceinline__ void wgrad_reduce_wide_body(int64_t blk, const float* part, int splits, int ntaps, int64_t cn, WIdx widx, float* dw,
                                                       float* sh /*[256]*/)
{
    const int t = threadIdx.x, lane = t >> 3;
    const int64_t i = blk * 8 + (t & 7);
    const int64_t total = (int64_t)ntaps * cn;
    float s = 0.0f;
    if (i < total) {
        const float* p0 = part + i;
        int k = lane;
        for (; k + 96 < splits; k += 128)
            s += (p0[(int64_t)k * total] + p0[(int64_t)(k + 32) * total]) + (p0[(int64_t)(k + 64) * total] + p0[(int64_t)(k + 96) * total]);
        for (; k < splits; k += 32) s += p0[(int64_t)k * total];
    }
    sh[t] = s;
    __syncthreads();
    for (int off = 128; off >= 8; off >>= 1) {
        if (t < off) sh[t] += sh[t + off];
        __syncthreads();
    }
    if (t < 8 && i < total) {
        const int ti = (int)(i / cn);
        dw[(int64_t)widx(ti) * cn + (i - (int64_t)ti * cn)] = sh[t];
    }
}
__global__ __launch_bounds__(256) void wgrad_reduce_wide_kernel(const float* part, int splits, int ntaps, int64_t cn,
                                                               ConvTaps taps, float* dw)
{
    __shared__ float sh[256];
    wgrad_reduce_wide_body((int64_t)blockIdx.x, part, splits, ntaps, cn, [&](int ti) { return taps.widx[ti]; }, dw, sh);
}

// ---- every partial-sum reduce of a backward pass in ONE launch ---------------------------------------------------------------
// The weight-gradient kernels leave [splits][...] partial sums; each layer's reduce is a 5-20 us launch of a few hundred
// blocks (~60 per DeepLab step, 0.41 ms of the weight-gradient queue).  pp_*_bwd_weight_partials stop after the partial kernel
// and describe the reduce as a pp_reduce_job; pp_wgrad_reduce_batch runs up to 64 jobs per launch - block -> (job, block of the
// job) through a prefix table in the kernel arguments, then the SAME code the single-layer kernels run (bit-identical results).
constexpr int kBatchJobs = 64;
struct ReduceBatch {
    const float* part[kBatchJobs];
    float* dst[kBatchJobs];
    int cn[kBatchJobs];            // elements per tap (kinds 1, 2) / outputs (kind 3)
    int splits[kBatchJobs];
    int start[kBatchJobs + 1];     // first block of every job
    uint8_t ntaps[kBatchJobs], kind[kBatchJobs];
    uint8_t widx[kBatchJobs][12];
    int n;
};
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(ReduceBatch b)
{
    __shared__ double shd[256];
    int j = 0;
    while (j + 1 < b.n && (int)blockIdx.x >= b.start[j + 1]) ++j;      // uniform: scalar loads from the kernel arguments
    const int64_t blk = (int)blockIdx.x - b.start[j];
    const float* part = b.part[j];
    float* dst = b.dst[j];
    const int cn = b.cn[j], splits = b.splits[j], ntaps = b.ntaps[j];
    const uint8_t* wi = b.widx[j];
    if (b.kind[j] == 1) {
        wgrad_reduce4_body(blk, part, splits, ntaps, (int64_t)cn, [&](int ti) { return (int)wi[ti]; }, dst);
    } else if (b.kind[j] == 2) {
        wgrad_reduce_wide_body(blk, part, splits, ntaps, (int64_t)cn, [&](int ti) { return (int)wi[ti]; }, dst, reinterpret_cast<float*>(shd));
    } else {                                                            // 3: sum_partials_kernel (nn_ops.hip), mul = 1
        const int64_t i = blk * 8 + (threadIdx.x & 7);
        const double s = lanes32_sum(part, splits, (int64_t)cn, i, i < cn, shd);
        if (i < cn && threadIdx.x < 8) dst[i] = (float)(s * 1.0);
    }
}

// dbias[n] = sum_m dy[m][n]: per-row-block partials (grid.y row blocks), then a fixed-order fp64 combine.
__global__ __launch_bounds__(256) void bias_grad_partial_kernel(const float* dy, int64_t M, int C, int64_t ld,
                                                               int64_t rows_per_block, float* part /*[gridDim.y][C]*/)
{
    __shared__ float sh[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ry = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    float s = 0.0f;
    if (c < C) {
        // eight rows in flight per thread (one load per trip made the FPN decoder's 67 MB gradients a chain of load latencies: 45 us per
        // launch, 1.5 TB/s); fixed combination order
        float q[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int64_t m = r0 + ry;
        for (; m + 28 < r1; m += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = dy[(m + 4 * u) * ld + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] += v[u];
        }
        for (; m < r1; m += 4) q[0] += dy[m * ld + c];
        s = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
    }
    sh[ry][threadIdx.x & 63] = s;
    __syncthreads();
    if (ry == 0 && c < C)
        part[(int64_t)blockIdx.y * C + c] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void bias_grad_final_kernel(const float* part, int nblk, int C, float* dbias)
{
    __shared__ double sh[256];
    const int t = threadIdx.x, lane = t >> 3;
    const int c = blockIdx.x * 8 + (t & 7);
    double s = 0.0;
    if (c < C)
        for (int b = lane; b < nblk; b += 32) s += (double)part[(int64_t)b * C + c];
    sh[t] = s;
    __syncthreads();
    for (int off = 128; off >= 8; off >>= 1) {
        if (t < off) sh[t] += sh[t + off];
        __syncthreads();
    }
    if (t < 8 && c < C) dbias[c] = (float)sh[t];
}

// ---- fp32 convolutions on the bf16 matrix pipe: three-way operand split ("bf16x3") -------------------------------------------
// The fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate.  An fp32 value is EXACTLY the sum of three bf16 values
// (hi = bf16(a), mid = bf16(a - hi), lo = bf16(a - hi - mid): 8 + 8 + 8 mantissa bits, both subtractions are exact), a product of
// two bf16 values is exact in fp32, and the matrix pipe accumulates in fp32; so
//     a*b = hi*hi + (hi*mid + mid*hi) + (mid*mid + hi*lo + lo*hi) + O(2^-24 |a b|)
// - six bf16 MFMAs (v_mfma_f32_32x32x16_bf16, 32 cycles for 16 K values) replace eight fp32 MFMAs of 64 cycles: 2.67x the matrix
// rate at the accuracy of an fp32 GEMM (the dropped terms mid*lo, lo*mid, lo*lo are below the rounding of the fp32 accumulation;
// tests/test_conv_x3_gpu.py measures both paths against float64).  The operands are split ONCE per tensor by a bandwidth kernel
// (x3_split_kernel: [rows][C] fp32 -> 3 planes [rows + 1][Kp] bf16, Kp = C rounded up to 16, plus a row of zeros that padding
// taps / ragged rows read), the weights once per call into [tap][n][Kp] (k contiguous: the layout both MFMA operands want), and
// the implicit-GEMM kernel is a pure LDS-DMA pipeline: no conversion in its loop.
//   tile 128 x BN (BN = 128 | 64), 4 waves (2 x 2), K step = 16 bf16 per plane (32-byte rows), three stages of
//   3 x (128 + BN) x 32 B; a wave DMAs 32 A rows and 32 B rows of each plane per step (1-KiB pieces, lane -> (row, 16-byte half),
//   half swizzled by bit 3 of the row so that the fragment reads are conflict-free); fragments are ds_read_b128.
__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f)
{
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7F800000u) == 0x7F800000u) return (uint16_t)(u >> 16);       // inf / nan: truncate
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

__device__ __forceinline__ void x3_split1(float v, uint16_t& hi, uint16_t& mid, uint16_t& lo)
{
    hi = f32_to_bf16_rne(v);
    const float r1 = v - bf16_to_f32(hi);
    mid = f32_to_bf16_rne(r1);
    const float r2 = r1 - bf16_to_f32(mid);
    lo = f32_to_bf16_rne(r2);
}

// x [rows][C] (pixel stride ldx) -> planes [3][Kp/16 chunks][rows + 1][16] bf16: CHUNK-major, so that the 32 consecutive rows x 16
// channels a wave DMAs per piece are ONE contiguous KiB (eight full cache lines; row-major planes made every piece touch 32 lines
// for 32 bytes each and the kernel ran at the texture-address rate: 178 us of data movement for 92 us of MFMA work).
// One thread = (chunk, row, 8-channel half): consecutive threads write consecutive 16-byte pieces.
__global__ __launch_bounds__(256) void x3_split_kernel(const float* x, int64_t ldx, int64_t rows, int C, uint16_t* out, int Kp, int64_t plane)
{
    const int nchunk = Kp / 16;
    const int64_t per_chunk = (rows + 1) * 2;
    const int64_t total = per_chunk * nchunk;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int ch = (int)(e / per_chunk);
        const int64_t rem = e - (int64_t)ch * per_chunk;
        const int64_t r = rem >> 1;
        const int c0 = ch * 16 + (int)(rem & 1) * 8;
        uint16_t h[8], m[8], l[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { h[j] = m[j] = l[j] = 0; }
        if (r < rows) {
            if (c0 + 8 <= C && (ldx & 3) == 0) {
                const float4 v0 = *reinterpret_cast<const float4*>(x + r * ldx + c0), v1 = *reinterpret_cast<const float4*>(x + r * ldx + c0 + 4);
                const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) x3_split1(v[j], h[j], m[j], l[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (c0 + j < C) x3_split1(x[r * ldx + c0 + j], h[j], m[j], l[j]);
            }
        }
        uint4 ph, pm, pl;
        ph.x = h[0] | ((uint32_t)h[1] << 16); ph.y = h[2] | ((uint32_t)h[3] << 16); ph.z = h[4] | ((uint32_t)h[5] << 16); ph.w = h[6] | ((uint32_t)h[7] << 16);
        pm.x = m[0] | ((uint32_t)m[1] << 16); pm.y = m[2] | ((uint32_t)m[3] << 16); pm.z = m[4] | ((uint32_t)m[5] << 16); pm.w = m[6] | ((uint32_t)m[7] << 16);
        pl.x = l[0] | ((uint32_t)l[1] << 16); pl.y = l[2] | ((uint32_t)l[3] << 16); pl.z = l[4] | ((uint32_t)l[5] << 16); pl.w = l[6] | ((uint32_t)l[7] << 16);
        const int64_t o = e * 8;                                    // (chunk, row, half) -> 8 bf16 each, in that order
        *reinterpret_cast<uint4*>(out + o) = ph;
        *reinterpret_cast<uint4*>(out + plane + o) = pm;
        *reinterpret_cast<uint4*>(out + 2 * plane + o) = pl;
    }
}

// weights HWIO [taps][Cin][Cout] -> B planes [3][Kp/16][taps * n_rows + 1][16].  transpose == true (forward): row = tap*Cout + n,
// k = c; false (backward-data): row = tap*Cin + c, k = n.  One thread = (chunk, row, half).
__global__ __launch_bounds__(256) void x3_split_w_kernel(const float* w, int taps, int Cin, int Cout, int transpose, uint16_t* out, int Kp,
                                                        int64_t plane)
{
    const int n_rows = transpose ? Cout : Cin, K = transpose ? Cin : Cout;
    const int nchunk = Kp / 16;
    const int64_t rows = (int64_t)taps * n_rows;
    const int64_t per_chunk = (rows + 1) * 2;
    const int64_t total = per_chunk * nchunk;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        // (chunk, half, row) with the row fastest: consecutive threads read consecutive n of one k (coalesced when transposing)
        const int64_t ch2 = e / (rows + 1);
        const int64_t r = e - ch2 * (rows + 1);
        const int ch = (int)(ch2 >> 1);
        const int k0 = ch * 16 + (int)(ch2 & 1) * 8;
        uint16_t h[8], m[8], l[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            h[j] = m[j] = l[j] = 0;
            if (r < rows && k0 + j < K) {
                const int t = (int)(r / n_rows), n = (int)(r - (int64_t)t * n_rows);
                const float v = transpose ? w[((int64_t)t * Cin + (k0 + j)) * Cout + n] : w[((int64_t)t * Cin + n) * Cout + k0 + j];
                x3_split1(v, h[j], m[j], l[j]);
            }
        }
        const int64_t o = ((int64_t)ch * per_chunk + r * 2 + (ch2 & 1)) * 8;
        uint4 ph, pm, pl;
        ph.x = h[0] | ((uint32_t)h[1] << 16); ph.y = h[2] | ((uint32_t)h[3] << 16); ph.z = h[4] | ((uint32_t)h[5] << 16); ph.w = h[6] | ((uint32_t)h[7] << 16);
        pm.x = m[0] | ((uint32_t)m[1] << 16); pm.y = m[2] | ((uint32_t)m[3] << 16); pm.z = m[4] | ((uint32_t)m[5] << 16); pm.w = m[6] | ((uint32_t)m[7] << 16);
        pl.x = l[0] | ((uint32_t)l[1] << 16); pl.y = l[2] | ((uint32_t)l[3] << 16); pl.z = l[4] | ((uint32_t)l[5] << 16); pl.w = l[6] | ((uint32_t)l[7] << 16);
        *reinterpret_cast<uint4*>(out + o) = ph;
        *reinterpret_cast<uint4*>(out + plane + o) = pm;
        *reinterpret_cast<uint4*>(out + 2 * plane + o) = pl;
    }
}

__device__ __forceinline__ void x3_glds16(uint32_t voff, const void* sbase_in, uint32_t lds_dst_in)
{
    // (wave-uniform by construction; readfirstlane tells the compiler so)
    const uint32_t lds_dst = __builtin_amdgcn_readfirstlane(lds_dst_in);
    const uint64_t sb = reinterpret_cast<uint64_t>(sbase_in);
    const uint64_t sbu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32)) << 32) |
                         (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sb);
    const void* sbase = reinterpret_cast<const void*>(sbu);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// the three planes of one operand piece behind ONE M0 save / restore: same lane offset, three plane bases, LDS destinations
// `stride` bytes apart
__device__ __forceinline__ void x3_glds16x3(uint32_t voff, const void* s0, const void* s1, const void* s2, uint32_t lds_dst_in, uint32_t stride_in)
{
    const uint32_t d0 = __builtin_amdgcn_readfirstlane(lds_dst_in);
    const uint32_t st = __builtin_amdgcn_readfirstlane(stride_in);
    auto uni = [](const void* q) -> const void* {
        const uint64_t v = reinterpret_cast<uint64_t>(q);
        return reinterpret_cast<const void*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) |
                                             (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v));
    };
    const void* b0 = uni(s0); const void* b1 = uni(s1); const void* b2 = uni(s2);
    unsigned keep, t;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
                 "s_add_u32 %1, %6, %7\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4\n\t"
                 "s_add_u32 %1, %1, %7\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep), "=&s"(t) : "v"(voff), "s"(b0), "s"(b1), "s"(b2), "s"(d0), "s"(st) : "memory", "scc");
}

// VAR: experiment switch of the 256 x 128 form (pp_debug_set_x3_variant; 0 = the product kernel; profiles/r05_conv_x3_power.txt).
// 11: the round-3 form (one barrier per K step, DMA four steps ahead); the others are changes to THAT form - 1 / 2: ring of 3 / 2 stages;
// 3: s_setprio 1 for the second-dispatched half of the block; 4: the upper four waves (A rows only) issue their DMA two MFMA groups
// later than their SIMD partners; 5 / 6 / 7: timing ablations with WRONG results - no MFMAs / MFMAs only (no DMA after the prologue, no
// fragment reads, no barrier) / no barrier and no wait; 8: one barrier per two K steps (DMA three steps ahead); 10: the step's DMA
// addresses derived behind its first MFMA group; 9: 8 + 10 = what VAR 0 is.
template <int BM, int BN, int VAR = 0>
__global__ __launch_bounds__(BM * 2, (BM == 128 ? 2 : 2)) void conv_x3_kernel(ConvParams p, X3Operands o)
{
    // waves: (BM / 64) x 2, each a 64 x (BN / 2) tile; 256 x 128 = eight waves = ONE block per CU (108 KiB of LDS), 128 x BN = four
    // waves, two blocks per CU.  Pipeline as conv_igemm_dma_kernel: the fragments of step k sit in registers (read during step
    // k-1), so the DMA of step k+3 can overwrite the ring slot of step k right behind the barrier; the 24 (12) MFMAs of a step go
    // out in six groups (one per product term) with the fragment reads of step k+1 and one DMA piece behind each group.
    constexpr int TM = 2, TN = BN / 64, WN = 2, NWAVE = BM / 32;
    constexpr int NSTAGE = VAR == 1 ? 3 : VAR == 2 ? 2 : (BM == 256 && BN == 128) ? 4 : (BM == 256 ? 5 : 3);      // one block per CU: as much ring as 160 KiB hold
    // the 256 x 128 product form (VAR 0; 11 = the round-3 form, one barrier per step, for A/B): measured 1 - 1.5 % faster, bit-identical
    constexpr bool PROD = VAR == 0 && BM == 256 && BN == 128;
    constexpr bool PAIR = PROD || VAR == 8 || VAR == 9;           // one barrier per TWO K steps; the DMA runs NSTAGE - 1 steps ahead
    constexpr bool LATE_BEGIN = PROD || VAR == 9 || VAR == 10;    // the step's DMA addresses (a table read from LDS) are derived behind the step's first MFMA group, not between the barrier and it
    constexpr int PRE = PAIR ? NSTAGE - 1 : NSTAGE;
    constexpr int A_BYTES = BM * 32, B_BYTES = BN * 32;                    // one plane of one stage
    constexpr int STAGE_BYTES = 3 * (A_BYTES + B_BYTES);
    constexpr int NBW = BN / 32;                                          // waves that DMA B rows (32 rows each)
    static_assert(NBW <= NWAVE, "B rows are loaded by the first BN/32 waves");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NSTAGE * STAGE_BYTES];
    struct Frags { bf16x8_t a[3][TM], b[3][TN]; };

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int mt, nt;
    {
        const int ntn = p.n_tiles, nblk = gridDim.x, bid = blockIdx.x, per_xcd = nblk / 8;
        if (p.xcd_remap && per_xcd * 8 == nblk) {
            const int lin = (bid & 7) * per_xcd + (bid >> 3);
            mt = lin / ntn; nt = lin - mt * ntn;
        } else {
            mt = bid / ntn; nt = bid - mt * ntn;
        }
    }
    const int64_t m0 = (int64_t)mt * BM;
    const int n0 = o.col_base + nt * BN;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);

    // ---- per-lane DMA addressing: lane -> (row = lane >> 1 of this wave's 32 rows, 16-byte half swizzled by bit 3 of the row)
    const int lr = lane >> 1;
    const int lhalf = (lane & 1) ^ ((lr >> 3) & 1);
    uint32_t a_e0 = 0;                                   // byte offset of (pixel row, half) inside a plane, tap (0,0), chunk 0
    unsigned a_vm = 0u;
    {
        const int64_t m = m0 + wave * 32 + lr;
        if (m < p.M) {
            const unsigned mu = (unsigned)m;
            const unsigned t = mu / (unsigned)p.Wo;
            const int ow = (int)(mu - t * (unsigned)p.Wo);
            const unsigned bb = t / (unsigned)p.Ho;
            const int oh = (int)(t - bb * (unsigned)p.Ho);
            const int ih0 = oh * p.stride, iw0 = ow * p.stride;
            a_e0 = (uint32_t)((((int)bb * p.H + ih0) * p.W + iw0) * 16 + lhalf * 8) * 2u;
            for (int t2 = 0; t2 < p.taps.n; ++t2) {
                const int ih = ih0 + p.taps.dh[t2], iw = iw0 + p.taps.dw[t2];
                a_vm |= ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) ? (1u << t2) : 0u;
            }
        }
    }
    const bool loads_b = wave < NBW;
    const int b_row = n0 + wave * 32 + lr;
    const bool b_ok = loads_b && b_row < p.Cn;
    const uint32_t b_e0 = (uint32_t)(b_row * 16 + lhalf * 8) * 2u;
    const uint32_t a_zero = o.a_zero + (uint32_t)(lhalf * 16), b_zero = o.b_zero + (uint32_t)(lhalf * 16);
    const uint16_t* a0 = o.a; const uint16_t* a1 = o.a + o.a_plane; const uint16_t* a2 = o.a + 2 * o.a_plane;
    const uint16_t* b0 = o.b; const uint16_t* b1 = o.b + o.b_plane; const uint16_t* b2 = o.b + 2 * o.b_plane;

    __shared__ int s_tap[32][2];
    if (tid < p.taps.n) {
        s_tap[tid][0] = (p.taps.dh[tid] * p.W + p.taps.dw[tid]) * 32;               // byte shift of the A row (32-byte rows)
        s_tap[tid][1] = p.taps.widx[tid] * o.n_rows * 32;                            // byte offset of the tap's B rows
    }
    __syncthreads();

    const int nchunk = o.Kp / 16;
    const int nk = p.taps.n * nchunk;
    const int ks_beg = p.splits > 1 ? (int)blockIdx.y * p.ks_per_split : 0;
    const int ks_end = p.splits > 1 ? (ks_beg + p.ks_per_split < nk ? ks_beg + p.ks_per_split : nk) : nk;
    const int n = ks_end - ks_beg;
    int is_ch = ks_beg / p.taps.n, is_ti = ks_beg - is_ch * p.taps.n;                // chunk outer, tap inner (L2 reuse of the halo rows)

    // the DMA of one step in (up to) six pieces; begin() computes the two lane offsets, piece(i) issues one load
    uint32_t st_va = 0, st_vb = 0, st_lds = 0;
    auto issue_begin = [&](int stage) {
        const int ta = s_tap[is_ti][0], tb = s_tap[is_ti][1];
        st_va = ((a_vm >> is_ti) & 1u) ? a_e0 + (uint32_t)ta + (uint32_t)is_ch * o.a_chunk : a_zero;
        st_vb = b_ok ? b_e0 + (uint32_t)tb + (uint32_t)is_ch * o.b_chunk : b_zero;
        st_lds = lds0 + (uint32_t)(stage * STAGE_BYTES) + (uint32_t)(wave * 1024);
        if (++is_ti == p.taps.n) { is_ti = 0; ++is_ch; }
    };
    auto issue_piece = [&](int i) {                      // two batches of three planes: A behind MFMA group 0, B behind group 3
        if constexpr (VAR == 4) {
            if (i == (loads_b ? 0 : 2)) x3_glds16x3(st_va, a0, a1, a2, st_lds, A_BYTES);
            else if (i == 3 && loads_b) x3_glds16x3(st_vb, b0, b1, b2, st_lds + 3 * A_BYTES, B_BYTES);
            return;
        }
        if (i == 0) x3_glds16x3(st_va, a0, a1, a2, st_lds, A_BYTES);
        else if (i == 3 && loads_b) x3_glds16x3(st_vb, b0, b1, b2, st_lds + 3 * A_BYTES, B_BYTES);
    };
    auto issue_all = [&](int stage) {
        issue_begin(stage);
#pragma unroll
        for (int i = 0; i < 6; ++i) issue_piece(i);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // fragment slot of (row r, logical half h): 16-byte slot r*2 + (h ^ ((r >> 3) & 1))
    int a_slot[TM], b_slot[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) { const int r = (wm * TM + t) * 32 + l31; a_slot[t] = r * 2 + (h ^ ((r >> 3) & 1)); }
#pragma unroll
    for (int t = 0; t < TN; ++t) { const int r = (wn * TN + t) * 32 + l31; b_slot[t] = r * 2 + (h ^ ((r >> 3) & 1)); }
    auto read_a = [&](int stage, Frags& F, int pl) {
        const bf16x8_t* S = reinterpret_cast<const bf16x8_t*>(smem + stage * STAGE_BYTES);
#pragma unroll
        for (int t = 0; t < TM; ++t) F.a[pl][t] = S[pl * (A_BYTES / 16) + a_slot[t]];
    };
    auto read_b = [&](int stage, Frags& F, int pl) {
        const bf16x8_t* S = reinterpret_cast<const bf16x8_t*>(smem + stage * STAGE_BYTES);
#pragma unroll
        for (int t = 0; t < TN; ++t) F.b[pl][t] = S[3 * (A_BYTES / 16) + pl * (B_BYTES / 16) + b_slot[t]];
    };
    // six product terms, smallest first: (A plane, B plane)
    auto mma_term = [&](const Frags& F, int term) {
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[TA[term]][tm], F.b[TB[term]][tn], acc[tm][tn], 0, 0, 0);
    };
    auto wait_pieces = [&](int steps_in_flight) {           // my own pieces: all but the newest `steps_in_flight` steps have landed
        if (loads_b) {
            if (steps_in_flight >= 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else if (steps_in_flight == 3) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
            else if (steps_in_flight == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (steps_in_flight == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (steps_in_flight >= 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (steps_in_flight == 3) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else if (steps_in_flight == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (steps_in_flight == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    // STEADY (compile time): at least NSTAGE steps remain, so every `if` of the step is taken - the steady-state loop has no branches
    // besides the wave-uniform "this wave loads B rows" one (runtime-uniform branches cost conv_igemm_dma_kernel ~10 %)
    auto kstep = [&](auto steady_tag, auto odd_tag, int k, int stage, Frags& cur, Frags& nxt) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        constexpr bool ODD = decltype(odd_tag)::value;
        const bool rd = STEADY || k + 1 < n, dm = (STEADY || k + PRE < n) && VAR != 6;
        const int sn = stage + 1 == NSTAGE ? 0 : stage + 1;
        const int sd = PRE == NSTAGE ? stage : (stage == 0 ? NSTAGE - 1 : stage - 1);      // ring slot of step k + PRE
        if (rd && VAR != 7 && VAR != 6 && !(PAIR && ODD)) {
            // step k+1 has landed; steps k+2 .. k+NSTAGE-1 (as far as they exist) may still fly
            if (PAIR) {
                // (PAIR: this barrier also covers step k+1, which reads step k+2's fragments: everything issued so far has landed)
                wait_pieces(0);
            } else if (STEADY) {
                wait_pieces(NSTAGE - 2);
            } else {
                const int fly = n - (k + 2) < NSTAGE - 2 ? n - (k + 2) : NSTAGE - 2;
                wait_pieces(fly > 0 ? fly : 0);
            }
            __builtin_amdgcn_s_barrier();                    // everyone's pieces of step k+1; ring slot k%NSTAGE is free
            asm volatile("" ::: "memory");
        }
        if (dm && !LATE_BEGIN) issue_begin(sd);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            if constexpr (VAR != 5) mma_term(cur, g);
            else if (g == 0) {                                   // keep the fragment reads alive without the MFMAs that consume them
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                    for (int t = 0; t < TM; ++t) asm volatile("" :: "v"(cur.a[pl][t]));
#pragma unroll
                    for (int t = 0; t < TN; ++t) asm volatile("" :: "v"(cur.b[pl][t]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (rd && VAR != 6) { if (g < 3) read_a(sn, nxt, g); else read_b(sn, nxt, g - 3); }
            if (LATE_BEGIN && g == 0 && dm) issue_begin(sd);
            if (dm) issue_piece(g);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if constexpr (VAR == 3) {
        if (wave >= NWAVE / 2) __builtin_amdgcn_s_setprio(1);
    }
    Frags F0, F1;
#pragma unroll
    for (int s0 = 0; s0 < PRE; ++s0)
        if (s0 < n) issue_all(s0);
    if (n > 0) {
        wait_pieces((n < PRE ? n : PRE) - 1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) { read_a(0, F0, pl); read_b(0, F0, pl); }
        if constexpr (VAR == 6) { F1 = F0; asm volatile("" : "+v"(F1.a[0][0]), "+v"(F1.b[0][0])); }
    }
    // (unrolling the steady state over lcm(2, NSTAGE) steps so that ring slots become immediates was measured SLOWER: 203 -> 240 us
    // on the 256 -> 256 layer - the loop body no longer fits the instruction cache)
    int k = 0;
    for (; k + NSTAGE + 1 < n; k += 2) {
        kstep(std::true_type{}, std::false_type{}, k, k % NSTAGE, F0, F1);
        kstep(std::true_type{}, std::true_type{}, k + 1, (k + 1) % NSTAGE, F1, F0);
    }
    for (; k + 1 < n; k += 2) {
        kstep(std::false_type{}, std::false_type{}, k, k % NSTAGE, F0, F1);
        kstep(std::false_type{}, std::true_type{}, k + 1, (k + 1) % NSTAGE, F1, F0);
    }
    if (k < n) kstep(std::false_type{}, std::false_type{}, k, k % NSTAGE, F0, F1);
    conv_epilogue<TM, TN>(p, acc, m0, n0, wm, wn);
}

// pp_yardstick_mfma_stream: conv_x3_kernel's MFMA sequence from registers and nothing else (tools/probe/mfma_peak.hip holds the same kernel
// as a stand-alone program)
__device__ __forceinline__ unsigned probe_hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int DATA>
__global__ __launch_bounds__(512) void mfma_stream_kernel(float* out, int iters, unsigned seed)
{
    bf16x8_t a[3][2], b[3][2];
    for (int pl = 0; pl < 3; ++pl)
        for (int t = 0; t < 2; ++t)
            for (int i = 0; i < 8; ++i) {
                float va = 0.f, vb = 0.f;
                if (DATA == 1) { va = 1.0f + threadIdx.x * 0.001f + i; vb = 1.0f - i; }
                if (DATA == 2) {
                    const unsigned h = probe_hash32(seed + ((blockIdx.x * 512 + threadIdx.x) * 3 + pl) * 32 + t * 8 + i), g = probe_hash32(h + 0x9e3779b9u);
                    va = __uint_as_float((h & 0x807fffffu) | ((125u + (h >> 23) % 4u) << 23));
                    vb = __uint_as_float((g & 0x807fffffu) | ((125u + (g >> 23) % 4u) << 23));
                }
                a[pl][t][i] = (__bf16)va; b[pl][t][i] = (__bf16)vb;
            }
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 6; ++g)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[g]][i], b[TB[g]][j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) out[0] = s;
}

// ---- weight gradient on the bf16 matrix pipe (bf16x3 split) --------------------------------------------------------------------
// dW[t][c][n] = sum_m X[pix(m) + s_t][c] * dY[m][n]: the reduction index is the PIXEL, while both operands are stored with the
// channel contiguous - and the bf16 MFMA wants eight consecutive k per lane.  The tiles therefore go into LDS as they are
// ([16 pixels][16 channels] sub-blocks straight from the chunk-major planes x3_split_kernel writes: a DMA piece = two such blocks,
// each 512 contiguous bytes) and the fragments are read with ds_read_b64_tr_b16, the LDS transpose read of gfx950: within 16 lanes,
// lane i supplies the address of 4 contiguous channels of pixel row i/4 and lane c receives channel c of the four rows
// (tools/probe/tr16.hip).  Two such reads = the 8 k values of one MFMA operand.  Piece w of a plane sits at w*1024 + ((w+1)/2)*128
// bytes, so that the two 16-lane groups a read cycle serves (channel blocks 2B and 2B+1) use different bank halves.
// Tile BM (64 | 128 input channels) x 128 output channels for ONE tap and one pixel slice; partial sums + wgrad_reduce4_kernel as
// the fp32 kernels.  Pipeline as conv_x3_kernel.
struct X3WOperands {
    const uint16_t* x;        // X planes  [3][Cin_p/16][rows_x + 1][16]
    const uint16_t* dy;       // dY planes [3][Cout_p/16][M + 1][16]
    int64_t x_plane, dy_plane;
    uint32_t x_chunk, dy_chunk;   // bytes per channel chunk
    uint32_t x_zero, dy_zero;     // byte offset of the zero row (chunk 0)
};

typedef short v4s16_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s16_t lds_v4s16_t;

template <int BM>
__global__ __launch_bounds__(kThreads, 2) void conv_wgrad_x3_kernel(WgradParams p, X3WOperands o)
{
    constexpr int BN = 128, TM = BM / 64, TN = 2, WN = 2, NSTAGE = 3;
    constexpr int NPA = BM / 32, NPB = BN / 32;                         // pieces per plane: waves [0, NPA) load A, all four load B
    constexpr int REG_A = NPA * 1024 + (NPA / 2) * 128, REG_B = NPB * 1024 + (NPB / 2) * 128;   // one plane of one stage: piece w at w*1024 + ((w+1)/2)*128
    constexpr int STAGE_BYTES = 3 * (REG_A + REG_B);
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NSTAGE * STAGE_BYTES];
    struct Frags { bf16x8_t a[3][TM], b[3][TN]; };

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ctiles = (p.Cin + BM - 1) / BM;
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_remap) {
        const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        if ((total & 7u) == 0) {
            const unsigned lin = bx + gx * (by + gy * bz);
            const unsigned nl = (lin & 7u) * (total >> 3) + (lin >> 3);
            bx = nl % gx;
            const unsigned t2 = nl / gx;
            by = t2 % gy;
            bz = t2 / gy;
        }
    }
    const int c0 = (int)(bx % (unsigned)ctiles) * BM;
    const int ti = (int)(bx / (unsigned)ctiles);
    const int n0 = (int)by * BN;
    const int split = (int)bz;
    const int64_t m_beg = (int64_t)split * p.m_per_split;
    const int64_t m_end = m_beg + p.m_per_split < p.M ? m_beg + p.m_per_split : p.M;
    const int dh = p.taps.dh[ti], dw = p.taps.dw[ti];
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
    const int n = m_beg < m_end ? (int)((m_end - m_beg + 15) / 16) : 0;   // K steps (16 pixels) of this slice

    // ---- DMA addressing: lane -> (block select lane >> 5, pixel row (lane >> 1) & 15, 16-byte half lane & 1)
    const int hs = lane >> 5, pr = (lane >> 1) & 15, half = lane & 1;
    const bool loads_a = wave < NPA;
    const int a_chunk = (c0 >> 4) + wave + (NPA)*hs;          // piece w = channel blocks (w, w + NPA) of the tile
    const int b_chunk = (n0 >> 4) + wave + (NPB)*hs;
    const bool a_cok = loads_a && a_chunk * 16 < p.Cin, b_cok = b_chunk * 16 < p.Cout;
    const uint32_t a_base = (uint32_t)a_chunk * o.x_chunk + (uint32_t)(half * 16);
    const uint32_t b_base = (uint32_t)b_chunk * o.dy_chunk + (uint32_t)(half * 16);
    const uint32_t a_zero = o.x_zero + (uint32_t)(half * 16), b_zero = o.dy_zero + (uint32_t)(half * 16);
    RowIter it;
    it.init(m_beg + pr < p.M ? m_beg + pr : 0, p.Wo, p.Ho);
    int64_t is_m = m_beg + pr;                                 // pixel of this lane in the next step to issue
    const uint16_t* x0 = o.x; const uint16_t* x1 = o.x + o.x_plane; const uint16_t* x2 = o.x + 2 * o.x_plane;
    const uint16_t* d0 = o.dy; const uint16_t* d1 = o.dy + o.dy_plane; const uint16_t* d2 = o.dy + 2 * o.dy_plane;
    const uint32_t a_piece = (uint32_t)(wave * 1024 + ((wave + 1) >> 1) * 128);
    const uint32_t b_piece = (uint32_t)(3 * REG_A + wave * 1024 + ((wave + 1) >> 1) * 128);

    uint32_t st_va = 0, st_vb = 0, st_lds = 0;
    auto issue_begin = [&](int stage) {
        const bool okm = is_m < m_end;
        const int ih = it.oh * p.stride + dh, iw = it.ow * p.stride + dw;
        const bool oka = okm && a_cok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        const uint32_t pix = (uint32_t)((it.bb * p.H + ih) * p.W + iw);
        st_va = oka ? a_base + pix * 32u : a_zero;
        st_vb = (okm && b_cok) ? b_base + (uint32_t)is_m * 32u : b_zero;
        st_lds = lds0 + (uint32_t)(stage * STAGE_BYTES);
        it.advance(16, p.Wo, p.Ho);
        is_m += 16;
    };
    auto issue_piece = [&](int i) {                      // two batches of three planes
        if (i == 0) { if (loads_a) x3_glds16x3(st_va, x0, x1, x2, st_lds + a_piece, REG_A); }
        else if (i == 3) x3_glds16x3(st_vb, d0, d1, d2, st_lds + b_piece, REG_B);
    };
    auto issue_all = [&](int stage) {
        issue_begin(stage);
#pragma unroll
        for (int i = 0; i < 6; ++i) issue_piece(i);
    };

    // ---- transpose-read addressing: MFMA row block B (32 channels) = channel blocks 2B, 2B+1; lane -> (block (lane >> 4) & 1,
    //      pixel row 8*(lane >> 5) + ((lane & 15) >> 2), 8-byte piece lane & 3); second read: + 4 rows = + 128 bytes
    auto tr_off = [&](int blk32, int npieces) -> uint32_t {
        const int ci = 2 * blk32 + ((lane >> 4) & 1);                  // channel block inside the tile
        const int piece = ci % npieces, second = ci / npieces;         // piece w holds blocks (w, w + npieces)
        return (uint32_t)(piece * 1024 + ((piece + 1) >> 1) * 128 + second * 512 + (8 * h + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8);
    };
    uint32_t a_tr[TM], b_tr[TN];                              // byte offsets inside a stage
#pragma unroll
    for (int t = 0; t < TM; ++t) a_tr[t] = tr_off(wm * TM + t, NPA);
#pragma unroll
    for (int t = 0; t < TN; ++t) b_tr[t] = (uint32_t)(3 * REG_A) + tr_off(wn * TN + t, NPB);
    auto read_op = [&](uint32_t off) -> bf16x8_t {
        const v4s16_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s16_t*)(smem + off));
        const v4s16_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s16_t*)(smem + off + 128));
        typedef short v8s16_t __attribute__((ext_vector_type(8)));
        v8s16_t u = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        return *reinterpret_cast<bf16x8_t*>(&u);
    };
    auto read_a = [&](int stage, Frags& F, int pl) {
#pragma unroll
        for (int t = 0; t < TM; ++t) F.a[pl][t] = read_op(a_tr[t] + (uint32_t)(stage * STAGE_BYTES + pl * REG_A));
    };
    auto read_b = [&](int stage, Frags& F, int pl) {
#pragma unroll
        for (int t = 0; t < TN; ++t) F.b[pl][t] = read_op(b_tr[t] + (uint32_t)(stage * STAGE_BYTES + pl * REG_B));
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    auto mma_term = [&](const Frags& F, int term) {
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[TA[term]][tm], F.b[TB[term]][tn], acc[tm][tn], 0, 0, 0);
    };
    auto wait_pieces = [&](int steps_in_flight) {
        if (loads_a) {
            if (steps_in_flight >= 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (steps_in_flight == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (steps_in_flight >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (steps_in_flight == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    // a wave whose 64 channel rows lie entirely past Cin (the ragged last tile: 304 channels = 2.4 tiles of 128) keeps loading its
    // pieces and meeting the barriers but issues no fragment reads and no MFMA: the tile costs the MFMA pipe what its live rows cost
    auto kstep = [&](auto steady_tag, auto live_tag, int k, int stage, Frags& cur, Frags& nxt) {
        constexpr bool STEADY = decltype(steady_tag)::value;     // >= NSTAGE steps remain: no branches in the step
        constexpr bool LIVE = decltype(live_tag)::value;
        const bool rd = STEADY || k + 1 < n, dm = STEADY || k + NSTAGE < n;
        const int sn = stage + 1 == NSTAGE ? 0 : stage + 1;
        if (rd) {
            wait_pieces(STEADY || k + 2 < n ? 1 : 0);        // step k+1 has landed (k+2 may still fly)
            __builtin_amdgcn_s_barrier();                    // everyone's pieces of step k+1; ring slot k%3 is free
            asm volatile("" ::: "memory");
        }
        if (dm) issue_begin(stage);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            if constexpr (LIVE) mma_term(cur, g);
            __builtin_amdgcn_sched_barrier(0);
            if (LIVE && rd) { if (g < 3) read_a(sn, nxt, g); else read_b(sn, nxt, g - 3); }
            if (dm) issue_piece(g);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    Frags F0, F1;
#pragma unroll
    for (int s0 = 0; s0 < NSTAGE; ++s0)
        if (s0 < n) issue_all(s0);
    auto run = [&](auto live_tag) {
        constexpr bool LIVE = decltype(live_tag)::value;
        if (n > 0) {
            wait_pieces((n < NSTAGE ? n : NSTAGE) - 1);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if constexpr (LIVE) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) { read_a(0, F0, pl); read_b(0, F0, pl); }
            }
        }
        int k = 0;
        for (; k + NSTAGE + 1 < n; k += 2) {
            kstep(std::true_type{}, live_tag, k, k % NSTAGE, F0, F1);
            kstep(std::true_type{}, live_tag, k + 1, (k + 1) % NSTAGE, F1, F0);
        }
        for (; k + 1 < n; k += 2) {
            kstep(std::false_type{}, live_tag, k, k % NSTAGE, F0, F1);
            kstep(std::false_type{}, live_tag, k + 1, (k + 1) % NSTAGE, F1, F0);
        }
        if (k < n) kstep(std::false_type{}, live_tag, k, k % NSTAGE, F0, F1);
    };
    const bool wave_live = __builtin_amdgcn_readfirstlane(c0 + wm * (TM * 32) < p.Cin ? 1 : 0) != 0;
    if (wave_live) run(std::true_type{});
    else run(std::false_type{});                             // nothing of this wave's rows exists: it only feeds the ring

    float* out = p.part + ((int64_t)split * p.taps.n + ti) * p.Cin * p.Cout;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int nn = n0 + (wn * TN + tn) * 32 + l31;
        if (nn >= p.Cout || !wave_live) continue;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = c0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (c < p.Cin) wgrad_put(p, out + (int64_t)c * p.Cout + nn, acc[tm][tn][r]);
            }
    }
#ifdef PP_DEBUG_KNOBS
    if (p.counters) wgrad_fold_tail<BM, 128>(p, ti, (int)(bx % (unsigned)ctiles), ctiles, (int)by, c0, n0);
#endif
}

// ---- host ---------------------------------------------------------------------------------------------------
// live taps for forward-style indexing: input row = oh*stride + (th*dil - pad)
// Forward (flip == false): rows are output pixels, source row = oh*stride + dh, H/W = input size.
// Backward-data (flip == true): rows are INPUT pixels (Ho/Wo = input size), source dY row = (ih + dh)/bstride when
// divisible, H/W = dY size; `stride` is 1 in that call and `bstride` the convolution's stride.
static void build_taps(ConvTaps& t, int kh, int kw, int stride, int pad, int dil, int H, int W, int Ho, int Wo, bool flip,
                       int bstride = 1)
{
    t.n = 0;
    for (int th = 0; th < kh; ++th)
        for (int tw = 0; tw < kw; ++tw) {
            const int dh = flip ? pad - th * dil : th * dil - pad;
            const int dw = flip ? pad - tw * dil : tw * dil - pad;
            bool okh = false, okw = false;
            for (int oh = 0; oh < Ho && !okh; ++oh) {
                const int v = oh * stride + dh;
                okh = v >= 0 && v % bstride == 0 && v / bstride < H;
            }
            for (int ow = 0; ow < Wo && !okw; ++ow) {
                const int v = ow * stride + dw;
                okw = v >= 0 && v % bstride == 0 && v / bstride < W;
            }
            if (!(okh && okw)) continue;
            t.dh[t.n] = dh;
            t.dw[t.n] = dw;
            t.widx[t.n] = th * kw + tw;
            ++t.n;
        }
}

// Tile configuration + split-K factor for an (M x Cn x K) implicit GEMM.  Layers whose output has fewer than ~192
// tiles (every 1/16-resolution conv of MobileNetV2/ASPP at the BASELINE batch) leave most of the 256 CUs idle and
// run their whole K loop at one block per CU, exposing the full global-load latency every K-step; slicing the
// (tap, channel-chunk) loop over grid.y fills the chip, at the price of one [splits][M][Cn] round trip.
struct ConvPlan {
    int cfg;            // 0: 128x32, 1: 128x128 (or 128x64 under the A/B knob), 2: 64x64
    int64_t tiles;
    int n_tiles;
    int splits, ks_per_split;
    bool bn64;          // cfg 1 with 128x64 tiles
};

static int g_conv_splitk = 1;
static int g_splitk_tiles = 192, g_splitk_target = 512, g_splitk_min_nk = 12, g_splitk_min_iters = 4;

static int g_conv_deepk = 1;
static int g_conv_ablate_reduce = 0;   // TIMING ONLY (wrong results): bit 0 / 1 skip the split-K reduce launch of fwd / bwd-data
static int g_conv_dma = 1;         // 128-row tiles, LDS-DMA three-stage kernel: 0 off, 1 (default) forward + backward-data, 2 forward only,
                                   // 3 forward + the backward-data of the 128x64-tiled layers only.  History: before the steady-state loop
                                   // lost its branches, backward-data through this kernel cost FPN 0.27 ms/step (48 KiB of LDS per block
                                   // beside the weight-gradient stream); after it: DeepLab 6.92 -> 6.89 ms/step, FPN 25.96 -> 26.02.
static int g_conv_dma64 = 1;       // 64x64 tiles through the LDS-DMA kernel: 0 off, 1 forward + backward-data (default: DeepLab 7.29 ->
                                   // 7.21 ms/step, FPN 28.06 -> 27.06), 2 forward only (7.25 / 27.42).  Replaces the 64-deep K step.
static int g_conv_big_bk32 = 0;    // 128x128 tiles with a 32-deep K step (A/B)
static int g_conv_n64 = 1;
static int g_conv_tap_inner = 1;
static int g_conv_bwd_rows = 1;       // pp_debug_set_conv_rows bit 0 switches conv1x1_rows_kernel / conv1x1_fwd_widen_kernel off (A/B)
static int64_t g_rows_fwd_min = 16384;    // pp_debug_set_conv_rows bit 1: forward rows kernel only from 65536 rows (A/B)
static int g_direct_rows_max = 4096;   // few-row pointwise layers (conv1x1_ksplit_dma_kernel): at most this many GEMM rows
static int g_conv_ksplit = 1, g_ksplit_k_min = 256;   // in-block split-K LDS-DMA kernel of the deep-K few-row 1x1 layers: 0 off, 1 rule, 2..4 force tile candidate 1..3 (A/B)
static int g_bwd_phases = 1;       // strided backward-data by pixel classes (below); pp_debug_set_conv_variant bit 24: masked-tap form (A/B)
static int g_shortk64 = 1;
static int g_big_tile_min = 384, g_wgrad_rows_min = 128;   // in-process sweep: rows_min 64/128: 7.30, 256: 7.32, 512: 7.66 ms

static ConvPlan plan_conv(int64_t M, int Cn, int Ck, int ntaps, bool vec = true)
{
    ConvPlan pl{};
    const int64_t mt128 = cdiv(M, 128), mt64 = cdiv(M, 64);
    if (Cn <= 32) {
        pl.cfg = 0; pl.n_tiles = 1; pl.tiles = mt128;
    } else if (Cn > 64 && mt128 * cdiv(Cn, 128) >= g_big_tile_min && !(g_shortk64 && ntaps == 1 && Ck <= 256)) {
        // (pointwise layers with a short reduction - ResNet50's 64 -> 256 at 32768 rows, 256 -> 1024 at 8192 - are all prologue and
        // epilogue at 128 x 128: a tile has four or sixteen K steps to amortise its 64 KiB of output over.  64 x 64 tiles: 31.1 -> 25.0 us
        // and 56.6 -> 51.0 us, tools/graded_1x1_sweep.py; thresholds bit 28 restores the large tiles)
        pl.cfg = 1;
        // ragged output width (backward-data of the 304-channel SegmentHead input): 128-wide tiles compute
        // cdiv(Cn,128)*128 columns (21 % waste at 304); 64-wide tiles are ~10 % slower per flop but waste 5 %
        const int rem = Cn % 128;
        pl.bn64 = g_conv_variant == 2 || (g_conv_n64 && rem != 0 && rem <= 64 && (cdiv(Cn, 128) * 128 - Cn) * 100 > 15 * Cn);
        pl.n_tiles = (int)cdiv(Cn, pl.bn64 ? 64 : 128);
        pl.tiles = mt128 * pl.n_tiles;
        if (g_conv_big_bk32 && !pl.bn64 && vec && Ck >= 64) pl.cfg = 4;
    } else {
        pl.cfg = 2; pl.n_tiles = (int)cdiv(Cn, 64); pl.tiles = mt64 * pl.n_tiles;
        // Deep-K variant: at one or two 64x64 blocks per CU the 16-deep K step has 512 MFMA cycles per wave to hide
        // ~2000 cycles of global-load latency behind; a 64-deep step has 2048 (and 4x the bytes in flight).
        // Measured (tools/ab_step.py, bench.py --network FPN): it pays on the ResNet50 shapes (8192 rows, K >= 256:
        // 28.15 -> 27.92 ms/step) and costs on MobileNetV2's 2048-row layers (7.37 -> 7.43 ms/step: 152 VGPRs, fewer
        // co-resident blocks for the split-K slices), so it is keyed on both.
        if (g_conv_deepk && !g_conv_dma64 && vec && Ck >= 256 && M >= 4096) pl.cfg = 3;
    }
    const int bk = pl.cfg == 3 ? 64 : (pl.cfg == 4 ? 32 : BK);
    const int nk = ntaps * (int)cdiv(Ck, bk);
    pl.splits = 1;
    pl.ks_per_split = nk;
    if (g_conv_splitk && pl.tiles < g_splitk_tiles && nk >= (pl.cfg == 3 ? 4 : g_splitk_min_nk)) {
        int64_t s = cdiv(g_splitk_target, pl.tiles);
        const int min_iters = pl.cfg == 3 ? 2 : g_splitk_min_iters;
        if (s > nk / min_iters) s = nk / min_iters;
        if (s > 32) s = 32;
        if (s > 1) {
            pl.ks_per_split = (int)cdiv(nk, s);
            pl.splits = (int)cdiv(nk, pl.ks_per_split);
        }
    }
    return pl;
}

// Shape-level test of the in-block split-K pointwise kernel (the launcher adds the pointer / stride alignment checks; when those fail
// although the workspace query answered 0, the tiled kernel runs as a single pass).
static bool ksplit_shape_ok(int64_t M, int Cn, int Ck, int ntaps, int stride)
{
    return g_conv_ksplit && ntaps == 1 && stride == 1 && M <= g_direct_rows_max && Ck >= g_ksplit_k_min && Cn >= 32 && Cn <= 512 &&
           Ck % 4 == 0 && Cn % 4 == 0;
}

// Partial-statistics rows a forward convolution with this plan writes (ConvParams::stats): one per wave row of the tile
// grid, or one per block of the split-K reduce.  0: this shape cannot deliver statistics (the caller runs the plain BatchNorm).
static int64_t splitk_stats_rows_per_block(int64_t M) { return M >= 8192 ? 64 : 16; }

static int64_t conv_stats_rows(const ConvPlan& pl, int64_t M, int Cn)
{
    if (pl.splits > 1) {
        if (Cn % 4 != 0 || Cn / 4 > 256) return 0;
        return cdiv(M, splitk_stats_rows_per_block(M));
    }
    if (pl.cfg == 0) return cdiv(M, 128) * 4;                       // 128x32 tiles, four waves stacked along M
    if (pl.cfg == 1 || pl.cfg == 4) return cdiv(M, 128) * 2;        // 128-row tiles, 2 x 2 waves
    return cdiv(M, 64) * 2;                                          // 64x64 tiles, 2 x 2 waves
}

// bf16x3 path (conv_x3_kernel): which problems take it, and what the caller's workspace must hold for it
static int g_conv_x3 = 1;
static int g_x3_var = 0;      // pp_debug_set_x3_variant: experiment forms of conv_x3_kernel<256,128> (see the kernel)
// ok: some bf16x3 kernel serves the problem.  classic: conv_x3_kernel on pre-split A planes (the caller's, or split here by x3_split_kernel) -
// the round-3 rule.  f32: conv_x3f_kernel (conv_x3f.hip) may serve it - the A operand is read as fp32 and split inside the kernel, so
// the layer pays no split launch and needs no A planes; taken whenever the caller holds no A planes, and it opens the path to the
// mid-size layers the split launches priced out (ResNet50 Bottleneck 1x1 / 3x3 at 8192 rows, resnet_models.py:58-94).
struct X3Plan { bool ok, classic, f32; int Kp; int64_t rows_a, a_plane, b_rows, b_plane; size_t bytes, b_off; };
static int g_conv_x3_mid = 1;
static double g_x3_mid_flop = 16e9, g_x3w_flop = 8e9;     // least work of a mid-size layer / a weight gradient (pp_debug_set_x3 bits 9-11 / 14-16)
static int g_x3_mid_tiles = 256;                          // least 128 x 128 tiles of a mid-size layer (bits 12-13)
static int g_x3f = 0;                                     // in-kernel A split (conv_x3f.hip; test build only, pp_debug_set_x3f bit 0 switches it ON: measured slower)
static double g_x3f_flop = 1e9;                           // least work of a layer that takes it without being a classic one
static int g_x3f_tiles = 128, g_x3f_k = 256;              // least 128-row tiles (128 or 64 wide) / least reduction length
static X3Plan x3_plan(const ConvPlan& pl, int64_t M, int64_t rows_a, int Ck, int n_rows, int ntaps_w, int ntaps_live, bool vec)
{
    X3Plan x{};
    // (shallow reductions - the 16 -> 96 expand at 130 x 258 - do not pay for the two split launches: 26 + 14 vs 19 us)
    // large-tile plans, and (ResNet50 at 32 x 64: 8192 rows) the 64x64-tiled layers that give 128-row tiles a full wave of blocks
    // and >= 16 GFLOP (512 -> 512 3x3: 112 TF on the fp32 pipe)
    const int64_t t128 = cdiv(M, 128) * cdiv(n_rows, 128);
    // (measured, FPN-ResNet50: with half a wave of blocks / below 16 GFLOP the operand splits and the idle CUs cost more than the
    // matrix rate gains - 256 -> 256 3x3 at 8192 rows 94 -> 150 us, 2048 -> 256 87 -> 155 us; 512 -> 512 3x3 345 -> 279 us)
    const double flop = 2.0 * (double)M * n_rows * ntaps_live * Ck;
    const bool mid = g_conv_x3_mid && pl.cfg == 2 && t128 >= g_x3_mid_tiles && flop >= g_x3_mid_flop;
    if (!g_conv_x3 || !vec || ntaps_live > 32) return x;
    x.classic = (pl.cfg == 1 || mid) && pl.splits <= 1 && (int64_t)ntaps_live * Ck >= 512;
    // in-kernel split: every classic layer (it replaces the split launch when the caller brings no planes), and whatever fills half the
    // chip with 128-row tiles of either width and has a reduction long enough to amortise the tile's prologue
    const int64_t t64 = n_rows % 64 == 0 ? cdiv(M, 128) * (n_rows / 64) : 0;
    x.f32 = g_x3f && n_rows >= 64 && (x.classic || (flop >= g_x3f_flop && (int64_t)ntaps_live * Ck >= g_x3f_k &&
                                                    std::max(t128, t64) >= g_x3f_tiles && (pl.cfg == 1 || pl.cfg == 2 || pl.cfg == 3 || pl.cfg == 4)));
    if (!x.classic && !x.f32) return x;
    x.Kp = (int)cdiv(Ck, 16) * 16;
    x.rows_a = rows_a;
    x.a_plane = (rows_a + 1) * x.Kp;
    x.b_rows = (int64_t)ntaps_w * n_rows;
    x.b_plane = (x.b_rows + 1) * x.Kp;
    if (x.b_plane * 2 >= (1ll << 32) - 4096) { x.classic = x.f32 = false; return x; }
    if (x.a_plane * 2 >= (1ll << 32) - 4096) x.classic = false;
    if (!x.classic && !x.f32) return x;
    x.b_off = x.classic ? align_up((size_t)3 * x.a_plane * 2, 256) : 0;      // (layers only the in-kernel split serves keep no room for A planes)
    x.bytes = x.b_off + align_up((size_t)3 * x.b_plane * 2, 256);
    x.ok = true;
    return x;
}

// ---- pointwise layers on the plain GEMM kernel (gemm_pw.hip) ----------------------------------------------------------------------
// Which tile form serves an M x N x K pointwise problem, or -1: the one with the least work on the busiest CU (tiles are handed out
// round-robin and a CU shares its matrix pipes among its resident blocks, so a launch lasts ceil(tiles / CUs) tiles), each tile
// priced at its MFMA time over the form's measured efficiency plus a fixed prologue / epilogue (profiles/r06_gemm_pw.txt).
static int g_gemm_pw = 1;                 // 0 off, 1 rule, 2.. force form (v - 2) (pp_debug_set_gemm_pw)
static int64_t g_gemm_pw_rows_min = 4096; // fewer rows: the in-block split-K kernel / the 64x64 tiles fill the chip better
static int device_cus();
static int gemm_pw_choose(int64_t M, int N, int K)
{
    if (!g_gemm_pw || K < 64 || N < 64) return -1;      // (narrower layers have their own whole-row kernels)
    if (g_gemm_pw >= 2) return M >= g_gemm_pw_rows_min ? g_gemm_pw - 2 : -1;
    // Below the row limit the in-block split-K kernel stays (ASPP fuse 1280 -> 256 @2048 rows: 22.9 against 32.6 us for the best tile form
    // here).  The one exception measured - MobileNetV2's 960 -> 320 projection, 26.4 us on 64 x 64 tiles against 30.3 (vendor 24.2) - is NOT
    // taken: that layer's input-affine form (BatchNorm applied where the operand is read) runs on the split-K kernel, and the two must add
    // in the same order (tests/test_bn_on_load_gpu.py: bit-equal to the plain convolution of the materialised input).
    if (M < g_gemm_pw_rows_min) return -1;
    static const double eff[6] = {0.88, 0.86, 0.85, 0.80, 0.78, 0.70}, fixed_us[6] = {10.0, 10.5, 6.0, 4.5, 4.5, 2.5};
    const int cus = device_cus();
    int best = -1;
    double best_t = 1e30;
    for (int f = 0; f < 6; ++f) {
        const int bm = gemm_pw_tile_rows(f), bn = gemm_pw_tile_cols(f);
        const int64_t tiles = cdiv(M, bm) * cdiv(N, bn);
        const double tile_us = 2.0 * bm * bn * (double)K / (4 * 64 * 2.4e3) / eff[f] + fixed_us[f];      // 4 SIMDs x 64 flop / cycle at 2.4 GHz; fill + store per tile
        const double t = (double)cdiv(tiles, cus) * tile_us;
        if (t < best_t) { best_t = t; best = f; }
    }
    return best;
}

template <bool BWD>
static int launch_conv_x3(ConvParams& p, const ConvPlan& pl, const X3Plan& x, int kh_kw, void* workspace, hipStream_t st, bool f32path)
{
    // f32path: conv_x3f_kernel reads the fp32 A operand itself - no x3_split_kernel launch, no A planes
    const uint16_t* ap = p.a_pre ? p.a_pre : reinterpret_cast<uint16_t*>(workspace);
    uint16_t* bp = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(workspace) + x.b_off);
    if (!p.a_pre && !f32path) {
        const int64_t ta = (x.rows_a + 1) * (x.Kp / 8);          // (chunk, row, half) items
        hipLaunchKernelGGL(x3_split_kernel, dim3((unsigned)std::min<int64_t>(cdiv(ta, 256), 4096)), dim3(256), 0, st, p.x, p.ldx, x.rows_a, p.Ck,
                           reinterpret_cast<uint16_t*>(workspace), x.Kp, x.a_plane);
        if (int rc = check_launch("x3_split_kernel")) return rc;
    }
    if (p.b_pre) {
        bp = const_cast<uint16_t*>(p.b_pre);
    } else {
        const int64_t tb = (x.b_rows + 1) * (x.Kp / 8);
        hipLaunchKernelGGL(x3_split_w_kernel, dim3((unsigned)std::min<int64_t>(cdiv(tb, 256), 4096)), dim3(256), 0, st, p.w, kh_kw, p.Cin, p.Cout,
                           BWD ? 0 : 1, bp, x.Kp, x.b_plane);
        if (int rc = check_launch("x3_split_w_kernel")) return rc;
    }
    X3Operands o{ap, bp, x.a_plane, x.b_plane, x.Kp, p.Cn, (uint32_t)(x.rows_a * 32), (uint32_t)(x.b_rows * 32),
                 (uint32_t)((x.rows_a + 1) * 32), (uint32_t)((x.b_rows + 1) * 32), 0};
    p.n_tiles = pl.n_tiles;
    p.splits = 1;
    p.ks_per_split = 0;
    p.part = nullptr;
    // 256-row tiles (eight waves, one block per CU) when that still gives every CU a block (g_conv_x3 bit 1: 128 rows always).
    // Ragged widths (304 = 2 x 128 + 48): the full 128-wide tile columns in one launch, the remainder (<= 64 columns) in a second
    // one with 64-wide tiles - 384 blocks of 256 x 128 on 256 CUs were two rounds for 1.19x the work (349 us; bit 2: that form)
    const int full128 = p.Cn / 128, rem = p.Cn - full128 * 128;
    const bool two = rem > 0 && rem <= 64 && full128 > 0 && !(g_conv_x3 & 4);
    bool n128_only = !two && (rem == 0 || rem > 64 || (g_conv_x3 & 4));
    const int64_t mt256 = cdiv(p.M, 256);
    // a narrow layer of the in-kernel-split path (1024 -> 256 at 8192 rows: 128 tiles of 128 x 128) takes 64-wide tiles when that is what
    // gives every CU a block
    if (f32path && n128_only && rem == 0 && cdiv(p.M, 128) * full128 < 200 && cdiv(p.M, 128) * full128 * 2 >= 128) n128_only = false;
    auto go = [&](bool n128, int ntile, int col_base) {
        // (the 64-wide remainder launch of a ragged width decides for itself: 128 blocks of 256 x 64 leave half of the CUs idle for
        // as long as a full tile column takes - 96 us for 16 % of the layer; 256 blocks of 128 x 64 do it in one short round)
        const bool m256 = !(g_conv_x3 & 2) && mt256 * ntile >= 200;
        const int64_t mtiles = m256 ? mt256 : cdiv(p.M, 128);
        o.col_base = col_base;
        p.n_tiles = ntile;
        const dim3 grid((unsigned)(mtiles * ntile));
#ifdef PP_DEBUG_KNOBS
        if (f32path) {
            (void)launch_conv_x3f(p, o, m256, n128, grid.x, st);
            return;
        }
#endif
        if (m256) {
            if (n128) {
                switch (g_x3_var) {
#ifdef PP_DEBUG_KNOBS        // experiment forms (some of them timing ablations with WRONG results): compiled into the test build only
                    case 1: hipLaunchKernelGGL((conv_x3_kernel<256, 128, 1>), grid, dim3(512), 0, st, p, o); break;
                    case 2: hipLaunchKernelGGL((conv_x3_kernel<256, 128, 2>), grid, dim3(512), 0, st, p, o); break;
                    case 3: hipLaunchKernelGGL((conv_x3_kernel<256, 128, 3>), grid, dim3(512), 0, st, p, o); break;
                    case 4: hipLaunchKernelGGL((conv_x3_kernel<256, 128, 4>), grid, dim3(512), 0, st, p, o); break;
                    case 5: hipLaunchKernelGGL((conv_x3_kernel<256, 128, 5>), grid, dim3(512), 0, st, p, o); break;
                    case 6: hipLaunchKernelGGL((conv_x3_kernel<256, 128, 6>), grid, dim3(512), 0, st, p, o); break;
                    case 7: hipLaunchKernelGGL((conv_x3_kernel<256, 128, 7>), grid, dim3(512), 0, st, p, o); break;
                    case 8: hipLaunchKernelGGL((conv_x3_kernel<256, 128, 8>), grid, dim3(512), 0, st, p, o); break;
                    case 9: hipLaunchKernelGGL((conv_x3_kernel<256, 128, 9>), grid, dim3(512), 0, st, p, o); break;
                    case 10: hipLaunchKernelGGL((conv_x3_kernel<256, 128, 10>), grid, dim3(512), 0, st, p, o); break;
                    case 11: hipLaunchKernelGGL((conv_x3_kernel<256, 128, 11>), grid, dim3(512), 0, st, p, o); break;
#endif
                    default: hipLaunchKernelGGL((conv_x3_kernel<256, 128>), grid, dim3(512), 0, st, p, o);
                }
            } else hipLaunchKernelGGL((conv_x3_kernel<256, 64>), grid, dim3(512), 0, st, p, o);
        } else {
            if (n128) hipLaunchKernelGGL((conv_x3_kernel<128, 128>), grid, dim3(256), 0, st, p, o);
            else      hipLaunchKernelGGL((conv_x3_kernel<128, 64>), grid, dim3(256), 0, st, p, o);
        }
    };
    if (two) {
        go(true, full128, 0);
        if (int rc = check_launch("conv_x3_kernel")) return rc;
        go(false, 1, full128 * 128);
    } else {
        go(n128_only, (int)cdiv(p.Cn, n128_only ? 128 : 64), 0);
    }
    return check_launch("conv_x3_kernel");
}

// Blocks of conv_igemm_dma_kernel<64, 64, false> that can be resident at once (occupancy x CUs); a fused conv + BatchNorm launch
// (spin-waiting blocks, see conv_epilogue_bn) uses at most HALF of it - the rule of the single-launch BatchNorm kernels.
static int device_cus();
static int conv_bn_capacity(int which = 0)        // 0: conv_igemm_dma_kernel<64, 64>, 1: <128, 64>, 2: conv_igemm_kernel<128, 32, 4, 1>
{
    static const int cap[3] = {
        [] { int n = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_igemm_dma_kernel<64, 64, false, true>, kThreads, 0) != hipSuccess) { (void)hipGetLastError(); return 0; } return n * device_cus(); }(),
        [] { int n = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_igemm_dma_kernel<128, 64, false, true>, kThreads, 0) != hipSuccess) { (void)hipGetLastError(); return 0; } return n * device_cus(); }(),
        [] { int n = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_igemm_kernel<128, 32, 4, 1, false, true>, kThreads, 0) != hipSuccess) { (void)hipGetLastError(); return 0; } return n * device_cus(); }()};
    return reserve_scaled(cap[which], device_cus());      // minus the CUs set aside for a resident communication kernel
}
static int g_conv_bn_fuse = 15;     // bit 0: fused conv + BatchNorm epilogue of the tiled kernels, bit 1: of the in-block split-K kernel, bit 2: backward form,
                                                 // bit 3: backward form of the in-block split-K kernel
static int ksplit_bn_capacity(int which)         // 0: <1,1,5>, 1: <2,1,3> (the forward candidates)
{
    static const int cap[2] = {
        [] { int n = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv1x1_ksplit_dma_kernel<1, 1, 5, false>, kThreads, 0) != hipSuccess) { (void)hipGetLastError(); return 0; } return n * device_cus(); }(),
        [] { int n = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv1x1_ksplit_dma_kernel<2, 1, 3, false>, kThreads, 0) != hipSuccess) { (void)hipGetLastError(); return 0; } return n * device_cus(); }()};
    return reserve_scaled(cap[which], device_cus());      // minus the CUs set aside for a resident communication kernel
}
// How a convolution that is to finish a training BatchNorm in its own epilogue would run: kind 0 = no fused kernel for this shape,
// 1 = conv_igemm_dma_kernel<64, 64>, 2 = conv1x1_ksplit_dma_kernel (forward candidates).  R = partial rows per strip = M tiles.
struct BnFusePlan { int kind; int R; int64_t blocks; KsplitCfg kc; };
static BnFusePlan bn_fuse_plan(const ConvParams& p, const ConvPlan& pl, bool vec)
{
    BnFusePlan f{0, 0, 0, {0, 0, 0}};
    if (!g_conv_bn_fuse || !vec || p.stats || p.in_scale || p.bias || p.epi.gamma || p.epi.res || p.epi.act != 0 || p.accumulate || p.bwd_stride > 1 ||
        p.Cn % 32 != 0 || (int64_t)p.B * p.H * p.W * p.ldx >= (1ll << 31) - (1ll << 24))
        return f;
    const bool ksplit = ksplit_shape_ok(p.M, p.Cn, p.Ck, p.taps.n, p.stride) && p.Cout % 4 == 0 && (int64_t)p.Cin * p.Cout < (1ll << 31) - (1ll << 24);
    if (ksplit) {
        if (!(g_conv_bn_fuse & 2)) return f;
        const KsplitCfg kc = ksplit_choose(p.M, p.Cn, false, g_conv_ksplit - 1);
        if (kc.tn != 1) return f;                            // (a forced backward candidate)
        const int64_t blocks = cdiv(p.M, 32 * kc.tm) * cdiv(p.Cn, 32);
        if (blocks > ksplit_bn_capacity(kc.tm == 2 ? 1 : 0) / 2) return f;
        f.kind = 2; f.R = (int)cdiv(p.M, 32 * kc.tm); f.blocks = blocks; f.kc = kc;
        return f;
    }
    const bool dma_ok = g_conv_dma64 && p.taps.n <= 32 && (int64_t)kMaxTaps * p.Cin * p.Cout < (1ll << 31);
    if (!(g_conv_bn_fuse & 1) || pl.splits != 1) return f;
    if (pl.cfg == 2 && dma_ok && pl.tiles <= conv_bn_capacity() / 2) {
        f.kind = 1; f.R = (int)cdiv(p.M, 64); f.blocks = pl.tiles;
    } else if (pl.cfg == 2 && dma_ok && g_conv_dma && cdiv(p.M, 128) * cdiv(p.Cn, 64) <= conv_bn_capacity(1) / 2) {
        f.kind = 3; f.R = (int)cdiv(p.M, 128); f.blocks = cdiv(p.M, 128) * cdiv(p.Cn, 64);       // 128 x 64 tiles: half the blocks
    } else if (pl.cfg == 0 && p.Cn == 32 && pl.tiles <= conv_bn_capacity(2) / 2) {
        f.kind = 4; f.R = (int)cdiv(p.M, 128); f.blocks = pl.tiles;                               // 128 x 32 register-staged kernel
    }
    return f;
}

// backward-data convolution + BatchNorm backward (conv_epilogue_bn_bwd): the 64x64-tiled LDS-DMA kernel, whole grid co-resident
static int conv_bn_bwd_capacity(int which = 0)   // 0: conv_igemm_dma_kernel<64, 64, true, true>, 1: conv_igemm_kernel<128, 32, 4, 1, true, true>
{
    static const int cap[2] = {
        [] { int n = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_igemm_dma_kernel<64, 64, true, true>, kThreads, 0) != hipSuccess) { (void)hipGetLastError(); return 0; } return n * device_cus(); }(),
        [] { int n = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_igemm_kernel<128, 32, 4, 1, true, true>, kThreads, 0) != hipSuccess) { (void)hipGetLastError(); return 0; } return n * device_cus(); }()};
    return reserve_scaled(cap[which], device_cus());      // minus the CUs set aside for a resident communication kernel
}
static int ksplit_bn_bwd_capacity(int which)     // 0: <1,1,5>, 1: <1,2,3> (the backward candidates)
{
    static const int cap[2] = {
        [] { int n = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv1x1_ksplit_dma_kernel<1, 1, 5, true>, kThreads, 0) != hipSuccess) { (void)hipGetLastError(); return 0; } return n * device_cus(); }(),
        [] { int n = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv1x1_ksplit_dma_kernel<1, 2, 3, true>, kThreads, 0) != hipSuccess) { (void)hipGetLastError(); return 0; } return n * device_cus(); }()};
    return reserve_scaled(cap[which], device_cus());      // minus the CUs set aside for a resident communication kernel
}
static BnFusePlan bn_fuse_plan_bwd(const ConvParams& p, const ConvPlan& pl, bool vec)
{
    BnFusePlan f{0, 0, 0, {0, 0, 0}};
    if (!(g_conv_bn_fuse & 4) || !vec || p.stats || p.in_scale || p.bias || p.epi.gamma || p.epi.res || p.epi.act != 0 || p.accumulate || p.bwd_stride > 1 ||
        p.stride != 1 || p.Cn % 32 != 0 || (int64_t)p.B * p.H * p.W * p.ldx >= (1ll << 31) - (1ll << 24))
        return f;
    // the fused form also takes the 1/8-resolution expands (8192 rows, K = 192: 32-channel block outputs) through the in-block split-K
    // kernel: as 64 single-pass 128x32 tiles their K loop is twelve exposed global-load latencies (29 us measured), as grid split-K
    // they need the second launch this fusion removes
    const bool ks_wide = g_conv_ksplit && (g_conv_bn_fuse & 8) && p.taps.n == 1 && p.M <= 2 * (int64_t)g_direct_rows_max && p.Ck >= 128 &&
                         p.Cn >= 32 && p.Cn <= 512 && p.Ck % 4 == 0;
    const bool ks_plain = ksplit_shape_ok(p.M, p.Cn, p.Ck, p.taps.n, p.stride);
    if (ks_plain || ks_wide) {
        // few-row, deep-K pointwise layers (the expand convolutions' backward-data at 1/16 resolution): the in-block split-K kernel
        const KsplitCfg kc = ksplit_choose(p.M, p.Cn, true, g_conv_ksplit - 1);
        const int64_t blocks = cdiv(p.M, 32) * cdiv(p.Cn, 32 * kc.tn);
        const bool ok = (g_conv_bn_fuse & 8) && p.Cout % 4 == 0 && (int64_t)p.Cin * p.Cout < (1ll << 31) - (1ll << 24) && kc.tm == 1 &&
                        blocks <= ksplit_bn_bwd_capacity(kc.tn == 2 ? 1 : 0) / 2;
        if (ok) { f.kind = 2; f.R = (int)cdiv(p.M, 32); f.blocks = blocks; f.kc = kc; return f; }
        if (ks_plain) return f;                              // (the unfused launch would take the split-K kernel: no tiled form for it)
    }
    const bool dma_ok = g_conv_dma64 == 1 && p.taps.n <= 32 && (int64_t)kMaxTaps * p.Cin * p.Cout < (1ll << 31);
    // (a plan with grid split-K - few tiles, deep K - runs as ONE pass here: the split's second launch is what the fusion removes)
    if (pl.cfg == 2 && (pl.splits == 1 || (g_conv_bn_fuse & 8)) && dma_ok && pl.tiles <= conv_bn_bwd_capacity() / 2) {
        f.kind = 1; f.R = (int)cdiv(p.M, 64); f.blocks = pl.tiles;
    } else if (pl.cfg == 0 && p.Cn == 32 && (g_conv_bn_fuse & 8) && pl.tiles <= conv_bn_bwd_capacity(1) / 2) {
        f.kind = 4; f.R = (int)cdiv(p.M, 128); f.blocks = pl.tiles;                               // 128 x 32 register-staged kernel
    }
    return f;
}

template <bool BWD>
static int launch_conv(const ConvParams& p_in, void* workspace, size_t ws_bytes, hipStream_t st, int kh_kw = 0)
{
    EventScope ev(st);
    ConvParams p = p_in;
    p.xcd_remap = g_conv_xcd_remap;
    const bool vec = g_conv_novec == 0 && p.Ck % 4 == 0 && p.Cin % 4 == 0 && p.Cout % 4 == 0 && p.ldx % 4 == 0 &&
                     (reinterpret_cast<uintptr_t>(p.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.w) & 15) == 0;
    ConvPlan pl = plan_conv(p.M, p.Cn, p.Ck, p.taps.n, vec);
    if (p.bn.part) {
        // conv -> training BatchNorm in one launch: only the kernel that implements it may run (the caller asked
        // pp_conv2d_fwd_bn_train_ok first; no silent fallback that would drop the BatchNorm)
        const BnFusePlan f = BWD ? bn_fuse_plan_bwd(p, pl, vec) : bn_fuse_plan(p, pl, vec);
        if (f.kind == 0) return fail(PP_ERR_UNSUPPORTED, "conv + BatchNorm: this shape has no fused kernel");
        p.bn.R = f.R;
        p.splits = 1; p.ks_per_split = 0; p.part = nullptr;
        if constexpr (BWD) {
            if (f.kind == 2) {
                p.n_tiles = (int)cdiv(p.Cn, 32 * f.kc.tn);
                if (f.kc.tn == 2) hipLaunchKernelGGL((conv1x1_ksplit_dma_kernel<1, 2, 3, true>), dim3((unsigned)f.blocks), dim3(kThreads), 0, st, p);
                else              hipLaunchKernelGGL((conv1x1_ksplit_dma_kernel<1, 1, 5, true>), dim3((unsigned)f.blocks), dim3(kThreads), 0, st, p);
                return check_launch("conv1x1_ksplit_dma_kernel<bn bwd>");
            }
            p.tap_inner = g_conv_tap_inner;
            p.n_tiles = pl.n_tiles;
            if (f.kind == 4) {
                hipLaunchKernelGGL((conv_igemm_kernel<128, 32, 4, 1, true, true>), dim3((unsigned)pl.tiles), dim3(kThreads), 0, st, p);
                return check_launch("conv_igemm_kernel<bn bwd>");
            }
            hipLaunchKernelGGL((conv_igemm_dma_kernel<64, 64, true, true>), dim3((unsigned)pl.tiles), dim3(kThreads), 0, st, p);
            return check_launch("conv_igemm_dma_kernel<bn bwd>");
        }
        if constexpr (!BWD) {
            if (f.kind == 2) {
                p.n_tiles = (int)cdiv(p.Cn, 32);
                if (f.kc.tm == 2) hipLaunchKernelGGL((conv1x1_ksplit_dma_kernel<2, 1, 3, false>), dim3((unsigned)f.blocks), dim3(kThreads), 0, st, p);
                else              hipLaunchKernelGGL((conv1x1_ksplit_dma_kernel<1, 1, 5, false>), dim3((unsigned)f.blocks), dim3(kThreads), 0, st, p);
                return check_launch("conv1x1_ksplit_dma_kernel<bn>");
            }
            p.tap_inner = g_conv_tap_inner;
            if (f.kind == 3) {
                p.n_tiles = (int)cdiv(p.Cn, 64);
                hipLaunchKernelGGL((conv_igemm_dma_kernel<128, 64, false, true>), dim3((unsigned)f.blocks), dim3(kThreads), 0, st, p);
            } else if (f.kind == 4) {
                p.n_tiles = pl.n_tiles;
                hipLaunchKernelGGL((conv_igemm_kernel<128, 32, 4, 1, false, true>), dim3((unsigned)pl.tiles), dim3(kThreads), 0, st, p);
            } else {
                p.n_tiles = pl.n_tiles;
                hipLaunchKernelGGL((conv_igemm_dma_kernel<64, 64, false, true>), dim3((unsigned)pl.tiles), dim3(kThreads), 0, st, p);
            }
        }
        return check_launch("conv_igemm_kernel<bn>");
    }
    if (kh_kw > 0 && !p.in_scale && p.bwd_stride <= 1) {
        // large-tile layers: six bf16 MFMAs per product instead of the fp32 MFMA (operands split once into the workspace)
        const X3Plan x = x3_plan(pl, p.M, (int64_t)p.B * p.H * p.W, p.Ck, p.Cn, kh_kw, p.taps.n, vec);
        // the caller's A planes -> conv_x3_kernel; no planes -> the in-kernel split where it applies, else (classic layers) split here
#ifdef PP_DEBUG_KNOBS
        const bool f32path = x.f32 && !p.a_pre && conv_x3f_supported(p);
#else
        const bool f32path = false;
#endif
        const bool ws_ok = (workspace && ws_bytes >= x.bytes && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0) ||
                           (f32path && p.b_pre != nullptr);          // (both operands provided for: nothing is written to the workspace)
        if (x.ok && (f32path || x.classic) && ws_ok && (reinterpret_cast<uintptr_t>(p.b_pre) & 255) == 0)
            return launch_conv_x3<BWD>(p, pl, x, kh_kw, workspace, st, f32path);
    }
    if (pl.splits > 1 && (!workspace || ws_bytes < (size_t)pl.splits * p.M * p.Cn * 4)) {
        pl.splits = 1;                   // no (or too small a) workspace: single pass
    }
    {
        // pointwise layers with enough rows: the plain GEMM kernel (one large tile per CU, gemm_pw.hip).  Forward: B = W [Cin][Cout] as it
        // lies; backward-data: dX = dY x W^T, the same weight read as the TRANSPOSED operand (b[n = Cin][k = Cout])
        const bool pw = p.taps.n == 1 && p.taps.dh[0] == 0 && p.taps.dw[0] == 0 && p.stride == 1 && vec && !p.stats && !p.in_scale && !p.epi.gamma &&
                        !p.epi.res && p.epi.act == 0 && !p.bn.part && p.bwd_stride <= 1 && !p.multi;
        if (pw) {
            // (backward-data: the transposed-operand form stores column by column; it pays where the reduction is long - measured per shape,
            // profiles/r06_gemm_pw.txt - and loses to the implicit-GEMM kernel on wide, shallow problems)
            const bool bt_ok = !BWD || g_gemm_pw >= 2 || (p.Ck >= 512 && 2 * p.Ck >= p.Cn) || p.Cn <= 128;
            const int form = bt_ok ? gemm_pw_choose(p.M, p.Cn, p.Ck) : -1;
            if (form >= 0 && gemm_pw_supported(p.x, p.ldx, p.w, p.Cout, p.y, p.ldy, p.M, p.Cn, p.Ck))
                return launch_gemm_pw(p.x, p.ldx, p.w, p.Cout, p.bias, p.y, p.ldy, p.M, p.Cn, p.Ck, p.accumulate, form, g_conv_xcd_remap, st, BWD ? 1 : 0);
        }
    }
    // few-row, deep-K pointwise layers: in-block split-K over wave-private LDS-DMA rings (conv1x1_ksplit_dma_kernel)
    const bool use_ksplit = ksplit_shape_ok(p.M, p.Cn, p.Ck, p.taps.n, p.stride) && vec && !p.stats && p.bwd_stride <= 1 && p.Cout % 4 == 0 &&
                            p.ldx % 4 == 0 && (int64_t)p.B * p.H * p.W * p.ldx < (1ll << 31) - (1ll << 24) &&
                            (int64_t)p.Cin * p.Cout < (1ll << 31) - (1ll << 24) && !(p.in_scale && p.Ck > 2048);
    if (p.in_scale) {
        // the producer's BatchNorm is applied where the A operand is read: only kernels that do so may run (no silent fallback)
        const bool pad0 = p.taps.n == 1 && p.taps.dh[0] == 0 && p.taps.dw[0] == 0 && p.stride == 1;
        if (BWD || !pad0 || !(use_ksplit || (pl.cfg == 0 && vec)))
            return fail(PP_ERR_UNSUPPORTED, "conv fwd: this shape has no input-affine kernel (ask pp_conv2d_fwd_accepts_affine_in first)");
    }
    if constexpr (!BWD) {
        // the MobileNetV2 stem on an even-sized packed image (conv_stem3x3s2_fwd_kernel)
        const bool stem = g_conv_bwd_rows && p.Cin == 3 && p.Cout == 32 && p.Cn == 32 && kh_kw == 9 && p.taps.n == 9 && p.stride == 2 &&
                          p.taps.dh[0] == -1 && p.taps.dw[0] == -1 && p.taps.dh[8] == 1 && p.taps.dw[8] == 1 && p.W % 2 == 0 && p.H % 2 == 0 &&
                          p.Wo * 2 == p.W && p.Ho * 2 == p.H && p.ldx == 3 && !p.bias && !p.stats && !p.in_scale && !p.epi.gamma && !p.epi.res &&
                          p.epi.act == 0 && !p.accumulate && p.ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(p.y) & 15) == 0 && p.M >= 16384 &&
                          (int64_t)p.B * p.H * p.W * 3 < (1ll << 31);
        if (stem) {
            hipLaunchKernelGGL(conv_stem3x3s2_fwd_kernel, dim3((unsigned)cdiv(p.M, 64)), dim3(256), 0, st, p);
            return check_launch("conv_stem3x3s2_fwd_kernel");
        }
    }
    {
        // narrow pointwise layers on large maps: whole rows through LDS, VALU (conv1x1_rows_kernel / conv1x1_fwd_widen_kernel)
        const bool plain = g_conv_bwd_rows && vec && p.taps.n == 1 && p.stride == 1 && p.bwd_stride <= 1 && !p.stats && !p.in_scale && !p.bias &&
                           !p.epi.gamma && !p.epi.res && p.epi.act == 0 && p.M >= 16384 && p.ldy % 4 == 0 &&
                           (reinterpret_cast<uintptr_t>(p.y) & 15) == 0 && (int64_t)p.B * p.H * p.W * p.ldx < (1ll << 31);
        const bool rows_ok = plain && (p.Cn == 16 || p.Cn == 24 || p.Cn == 32) && p.Ck <= 192 && p.Ck >= 16 && (BWD || p.M >= g_rows_fwd_min);
        if (rows_ok) {
            const size_t lds = (size_t)(64 * (p.Ck + 4) + 64 * p.Cn) * 4;
            const dim3 grid((unsigned)cdiv(p.M, 64));
            const unsigned inv_kq = (1u << 20) / (unsigned)(p.Ck / 4) + 1u;    // e / kq = (e * inv) >> 20 for e < 64 * 48
#define PP_ROWS(CN_) do { if (p.Ck <= 96) hipLaunchKernelGGL((conv1x1_rows_kernel<CN_, 6, BWD>), grid, dim3(256), lds, st, p, inv_kq); \
                          else if (p.Ck <= 144) hipLaunchKernelGGL((conv1x1_rows_kernel<CN_, 9, BWD>), grid, dim3(256), lds, st, p, inv_kq); \
                          else hipLaunchKernelGGL((conv1x1_rows_kernel<CN_, 12, BWD>), grid, dim3(256), lds, st, p, inv_kq); } while (0)
            if (p.Cn == 16)      PP_ROWS(16);
            else if (p.Cn == 24) PP_ROWS(24);
            else                 PP_ROWS(32);
#undef PP_ROWS
            return check_launch("conv1x1_rows_kernel");
        }
        if constexpr (BWD) {
            // backward-data of the project convolutions 96 -> 24 / 144 -> 24 (dy 24 channels wide, dx 96 / 144)
            const bool widen = plain && p.Cout == p.Ck && p.Ck == 24 && (p.Cn == 96 || p.Cn == 144);
            if (widen) {
                const dim3 grid((unsigned)cdiv(p.M, 64));
                if (p.Cn == 96) hipLaunchKernelGGL((conv1x1_fwd_widen_kernel<24, 96, true>), grid, dim3(256), 0, st, p);
                else            hipLaunchKernelGGL((conv1x1_fwd_widen_kernel<24, 144, true>), grid, dim3(256), 0, st, p);
                return check_launch("conv1x1_fwd_widen_kernel<bwd>");
            }
        }
        if constexpr (!BWD) {
            const bool widen = plain && !p.accumulate && ((p.Ck == 16 && p.Cn == 96) || (p.Ck == 24 && p.Cn == 144) || (p.Ck == 32 && p.Cn == 192));
            if (widen) {
                const dim3 grid((unsigned)cdiv(p.M, 64));
                if (p.Ck == 16)      hipLaunchKernelGGL((conv1x1_fwd_widen_kernel<16, 96>), grid, dim3(256), 0, st, p);
                else if (p.Ck == 24) hipLaunchKernelGGL((conv1x1_fwd_widen_kernel<24, 144>), grid, dim3(256), 0, st, p);
                else                 hipLaunchKernelGGL((conv1x1_fwd_widen_kernel<32, 192>), grid, dim3(256), 0, st, p);
                return check_launch("conv1x1_fwd_widen_kernel");
            }
        }
    }
    if (use_ksplit) {
        const KsplitCfg kc = ksplit_choose(p.M, p.Cn, BWD, g_conv_ksplit - 1);
        p.splits = 1;
        p.ks_per_split = 0;
        p.part = nullptr;
        p.n_tiles = (int)cdiv(p.Cn, 32 * kc.tn);
        const dim3 grid((unsigned)(cdiv(p.M, 32 * kc.tm) * p.n_tiles));
#define PP_KSPLIT(TM_, TN_, NST_) hipLaunchKernelGGL((conv1x1_ksplit_dma_kernel<TM_, TN_, NST_, BWD>), grid, dim3(kThreads), 0, st, p)
#define PP_KSPLIT_AFF(TM_, TN_, NST_) hipLaunchKernelGGL((conv1x1_ksplit_dma_kernel<TM_, TN_, NST_, false, true>), grid, dim3(kThreads), 0, st, p)
        if (!BWD && p.in_scale) {
            if (kc.tm == 1 && kc.tn == 1) PP_KSPLIT_AFF(1, 1, 4);          // (one stage less: the tables must fit beside two blocks per CU)
            else if (kc.tm == 2)          PP_KSPLIT_AFF(2, 1, 3);
            else                          PP_KSPLIT_AFF(1, 2, 3);
        } else
        if (kc.tm == 1 && kc.tn == 1) PP_KSPLIT(1, 1, 5);
        else if (kc.tm == 2)          PP_KSPLIT(2, 1, 3);
        else                          PP_KSPLIT(1, 2, 3);
#undef PP_KSPLIT_AFF
#undef PP_KSPLIT
        return check_launch("conv1x1_ksplit_dma_kernel");
    }
    p.tap_inner = g_conv_tap_inner;
    p.n_tiles = pl.n_tiles;
    p.splits = pl.splits;
    p.ks_per_split = pl.ks_per_split;
    p.part = reinterpret_cast<float*>(workspace);
    dim3 grid((unsigned)pl.tiles, (unsigned)pl.splits);
    if (pl.cfg == 0) {
        if (vec) hipLaunchKernelGGL((conv_igemm_kernel<128, 32, 4, 1, BWD, true>), grid, dim3(kThreads), 0, st, p);
        else     hipLaunchKernelGGL((conv_igemm_kernel<128, 32, 4, 1, BWD, false>), grid, dim3(kThreads), 0, st, p);
    } else if (pl.cfg == 1) {
        if (pl.bn64) {
            const bool dma_ok64 = vec && g_conv_dma && (!BWD || g_conv_dma == 1 || g_conv_dma == 3) && p.taps.n <= 32 && (!BWD || p.bwd_stride <= 1) &&
                                  (int64_t)p.B * p.H * p.W * p.ldx < (1ll << 31) - (1ll << 24) &&
                                  (int64_t)kMaxTaps * p.Cin * p.Cout < (1ll << 31);
            if (dma_ok64) hipLaunchKernelGGL((conv_igemm_dma_kernel<128, 64, BWD>), grid, dim3(kThreads), 0, st, p);
            else
            if (vec) hipLaunchKernelGGL((conv_igemm_kernel<128, 64, 2, 2, BWD, true>), grid, dim3(kThreads), 0, st, p);
            else     hipLaunchKernelGGL((conv_igemm_kernel<128, 64, 2, 2, BWD, false>), grid, dim3(kThreads), 0, st, p);
        } else {
            const bool dma_ok = vec && g_conv_dma && (!BWD || g_conv_dma == 1) && p.taps.n <= 32 && (!BWD || p.bwd_stride <= 1) &&
                                (int64_t)p.B * p.H * p.W * p.ldx < (1ll << 31) - (1ll << 24) &&
                                (int64_t)kMaxTaps * p.Cin * p.Cout < (1ll << 31);
            if (dma_ok) hipLaunchKernelGGL((conv_igemm_dma_kernel<128, 128, BWD>), grid, dim3(kThreads), 0, st, p);
            else if (vec) hipLaunchKernelGGL((conv_igemm_kernel<128, 128, 2, 2, BWD, true>), grid, dim3(kThreads), g_conv_lds_pad, st, p);
            else     hipLaunchKernelGGL((conv_igemm_kernel<128, 128, 2, 2, BWD, false>), grid, dim3(kThreads), g_conv_lds_pad, st, p);
        }
    } else if (pl.cfg == 4) {
        hipLaunchKernelGGL((conv_igemm_kernel<128, 128, 2, 2, BWD, true, 32>), grid, dim3(kThreads), 0, st, p);
    } else if (pl.cfg == 3) {
        hipLaunchKernelGGL((conv_igemm_kernel<64, 64, 2, 2, BWD, true, 64>), grid, dim3(kThreads), 0, st, p);
    } else {
        const bool dma_ok = vec && g_conv_dma64 && (!BWD || g_conv_dma64 == 1) && p.taps.n <= 32 && (!BWD || p.bwd_stride <= 1) &&
                            (int64_t)p.B * p.H * p.W * p.ldx < (1ll << 31) - (1ll << 24) &&
                            (int64_t)kMaxTaps * p.Cin * p.Cout < (1ll << 31);
        if (dma_ok)   hipLaunchKernelGGL((conv_igemm_dma_kernel<64, 64, BWD>), grid, dim3(kThreads), 0, st, p);
        else if (vec) hipLaunchKernelGGL((conv_igemm_kernel<64, 64, 2, 2, BWD, true>), grid, dim3(kThreads), 0, st, p);
        else          hipLaunchKernelGGL((conv_igemm_kernel<64, 64, 2, 2, BWD, false>), grid, dim3(kThreads), 0, st, p);
    }
    if (int rc = check_launch("conv_igemm_kernel")) return rc;
    if (pl.splits > 1 && p.stats) {
        const int64_t rpb = splitk_stats_rows_per_block(p.M);
        hipLaunchKernelGGL(splitk_reduce_stats_kernel, dim3((unsigned)cdiv(p.M, rpb)), dim3(256), 0, st, p.part, pl.splits, p.M, p.Cn,
                           p.bias, p.y, p.ldy, p.accumulate, p.stats, rpb);
        return check_launch("splitk_reduce_stats_kernel");
    }
    if (pl.splits > 1 && !(g_conv_ablate_reduce & (BWD ? 2 : 1))) {
        const bool plain = p.epi.gamma == nullptr && p.epi.res == nullptr && p.epi.act == 0 && p.Cn % 4 == 0 && p.ldy % 4 == 0;
        int64_t nb = cdiv(plain ? p.M * (p.Cn / 4) : p.M * p.Cn, 256);
        if (nb > 8192) nb = 8192;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, p.part, pl.splits, p.M, p.Cn, p.bias,
                           p.y, p.ldy, p.epi, p.accumulate);
        return check_launch("splitk_reduce_kernel");
    }
    return PP_OK;
}

static int g_narrow_rows_min = 16, g_narrow_splits_max = 4096;   // narrow-layer weight gradient: split geometry
// These kernels are latency-bound row walks (27 or fewer accumulators per thread), so more, shorter splits win as long as
// the partials stay small: up to 4096 splits while splits x taps x Cin x Cout x 4 B <= 16 MiB, never fewer than 1024
// (measured on the train step: 1024 -> 4096 splits for the stem and the 16/32-channel pointwise layers, 7.08 -> 7.05 ms).
static int64_t narrow_splits_max(int64_t floats_per_split)
{
    int64_t s = (16ll << 20) / (floats_per_split * 4);
    if (s > g_narrow_splits_max) s = g_narrow_splits_max;
    if (s < 1024) s = 1024;
    return s;
}
static int g_wgrad_narrow = 1;
static int g_wgrad_stem = 1;          // pp_debug_set_conv_variant bit 23 switches the specialised stem weight gradient off (A/B)
static int g_wgrad_m64 = 1;
static int g_wgrad_xcd = 1;
static int g_wgrad_dma = 1;          // LDS-DMA weight-gradient kernels: bit 0 = the 128-wide tiles (default on: DeepLab 7.17 -> 7.12 ms/step,
                                     // FPN 26.86 -> 26.77), bit 1 = the 64x64 tiles (off: 7.17 -> 7.20)
static const int g_wgrad_lds_pad_default = 0;
static int g_wgrad_lds_pad = 0;     // unused dynamic LDS per weight-gradient block: caps the blocks per CU (see pp_debug_set_wgrad_target)
static int g_wgrad_target = 1024;   // blocks aimed at by the split-M choice of the MFMA weight-gradient kernels
static int g_wgrad_balance = 1;     // pp_debug_set_wgrad_target bit 24: CU-balanced split choice of the MFMA-bound layers off (A/B)
static int g_wgrad_fold = 0;        // split-K reduction inside the weight-gradient launch: an experiment of the TEST BUILD (pp_debug_set_wgrad_target bit 25
                                    // switches it ON) - bit-identical, measured 20-50 % SLOWER per step (profiles/r06_wgrad_fold.txt)
// a range of `n` arrival counters no launch in flight shares: the ring is handed out in turn (64 K words; a launch takes at most 16 K)
static int* wgrad_counters_take(int n)
{
    static std::atomic<uint32_t> next{0};
    static int* base = nullptr;
    if (!base) {
        void* q = nullptr;
        if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_wgrad_counters)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        base = reinterpret_cast<int*>(q);
    }
    uint32_t at;
    for (;;) {
        uint32_t cur = next.load(std::memory_order_relaxed);
        at = (cur + (uint32_t)n > (uint32_t)kWgradCounters) ? 0u : cur;
        if (next.compare_exchange_weak(cur, at + (uint32_t)n, std::memory_order_relaxed)) break;
    }
    return base + at;
}

static int device_cus()
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

// Split count of an MFMA-bound weight gradient (>= 8 GFLOP: the SegmentHead / ResNet 3x3 layers).  Blocks are handed to CUs
// round-robin and a CU shares its MFMA pipes among its resident blocks, so the launch lasts as long as its busiest CU:
// ceil(tiles * s / CUs) blocks of ceil(M / s) rows each.  The target-block rule (s = target / tiles) lands anywhere between
// 70 % and 98 % balance (36 tiles: 29 slices = 1044 blocks = 4.08 per CU -> 5 on the busiest; 14 slices = 504 blocks -> 2 on
// every CU); this picks the slice count with the least busiest-CU work plus the cost of reducing one more slice of partial
// sums.  Measured on the two SegmentHead layers (4 x 64 x 128 rows): 383 -> 333 us and 482 -> 418 us.
static int64_t wgrad_balanced_splits(int64_t tiles, int64_t M, int bm, int bn, int64_t dw_elems, int64_t max_splits)
{
    const int cus = device_cus();
    const double t_row = 2.0 * bm * bn / 0.46e12;            // one tile row of MFMA work on a CU running at ~0.75 of its fp32 peak
    const double t_red = (double)dw_elems * 8.0 / 4.0e12;    // one more slice of partials written and read back
    int64_t s_lo = cdiv((int64_t)(1.8 * cus), tiles);        // at least ~two blocks per CU (latency hiding)
    if (s_lo > max_splits) s_lo = max_splits;
    if (s_lo < 1) s_lo = 1;
    int64_t best = s_lo;
    double best_cost = 1e30;
    for (int64_t sp = s_lo; sp <= max_splits; ++sp) {
        const int64_t rows = cdiv(cdiv(M, sp), BK) * BK;
        const double cost = (double)cdiv(tiles * sp, cus) * rows * t_row + sp * t_red;
        if (cost < best_cost * 0.999) { best_cost = cost; best = sp; }
    }
    return best;
}

// returns 0 when the layer was handled, 1 when it is not a narrow layer, < 0 on error
static void fill_reduce_job(pp_reduce_job* job, int kind, const float* part, float* dst, int64_t cn, int64_t splits, const ConvTaps& taps)
{
    job->part = part;
    job->dst = dst;
    job->cn = cn;
    job->splits = (int32_t)splits;
    job->ntaps = taps.n;
    job->kind = kind;
    for (int i = 0; i < 12; ++i) job->widx[i] = (uint8_t)(i < taps.n ? taps.widx[i] : 0);
}
// a reduce may be left to pp_wgrad_reduce_batch when the caller asked for it and the batch kernel's table can describe it
static bool reduce_deferrable(const pp_reduce_job* job, const ConvTaps& taps, int64_t cn, int64_t splits, const float* dbias)
{
    return job != nullptr && dbias == nullptr && taps.n <= 12 && (int64_t)taps.n * cn < (1ll << 31) && splits < (1 << 20);
}

static int launch_wgrad_narrow(WgradParams p, int kh, int kw, float* dw, float* dbias, bool* bias_done, void* workspace,
                               size_t ws_bytes, hipStream_t st, pp_reduce_job* job)
{
    *bias_done = false;
    const int nt = p.taps.n;
    int form = 0;            // 1: lanes over Cout (narrow input), 2: lanes over Cin (narrow output)
    // measured against the MFMA path (tools/conv_layer_table.py, B=4 256x512): wins for 3->32 3x3 (182 -> 136 us), 32->16
    // (78 -> 48), 16->96 (82 -> 44), 96->24 (35 -> 25), 144->24 (35 -> 31), 128->19 at full resolution (416 -> 214); loses
    // for 24->144, 32->192 and everything below 32 K pixels.  The 7x7 stem of the ResNets (147 accumulators per thread as one
    // block: slower than MFMA) runs as seven tap rows over grid.y: 763 us on the MFMA path.
    if ((p.Cin == 3 && (nt == 9 || nt == 49)) || (nt == 1 && (p.Cin == 16 || p.Cin == 32) && p.Cout <= 128)) form = 1;
    else if (nt == 1 && kh == 1 && kw == 1 && (p.Cout == 16 || p.Cout == 19 || p.Cout == 24 || p.Cout == 32) && p.Cin <= 256) form = 2;
    if (form == 0 || p.M < 32768) return 1;
    const int lanes_dim = form == 1 ? p.Cout : p.Cin;
    const int NL = lanes_dim >= 256 ? 256 : (int)(cdiv(lanes_dim, 16) * 16);
    const int RL = 256 / NL;
    const int64_t cn0 = (int64_t)p.Cin * p.Cout;
    int64_t splits = cdiv(p.M, g_narrow_rows_min);        // at least this many rows per split
    const int64_t smax = narrow_splits_max(nt * cn0);
    if (splits > smax) splits = smax;
    const int64_t nblk = cdiv(splits, RL);
    splits = nblk * RL;
    const int64_t rows_per_split = cdiv(p.M, splits);
    const int64_t cn = (int64_t)p.Cin * p.Cout;
    const size_t need = (size_t)splits * nt * cn * 4;
    if (!workspace || ws_bytes < need) return 1;           // caller sized the workspace for the MFMA path: use that
    p.part = reinterpret_cast<float*>(workspace);
    // the 3 -> <= 32 channel 3x3 / stride 2 / pad 1 stem on an even-width image with packed pixels: wide loads, one slice per block
    const bool stem = g_wgrad_stem && form == 1 && p.Cin == 3 && nt == 9 && kh == 3 && kw == 3 && p.stride == 2 && p.taps.dh[0] == -1 &&
                      p.taps.dw[0] == -1 && p.W % 2 == 0 && p.Wo * 2 == p.W && p.Ho * 2 == p.H && p.ldx == 3 && p.Cout <= 32 && RL == 8 &&
                      (int64_t)p.B * p.H * p.W * 3 < (1ll << 31);
    // the 3 -> <= 64 channel 7x7 / stride 2 / pad 3 stem of the ResNets on an even-sized packed image: seven waves = seven tap rows, scalar input loads
    const bool stem7 = g_wgrad_stem && form == 1 && p.Cin == 3 && nt == 49 && kh == 7 && kw == 7 && p.stride == 2 && p.taps.dh[0] == -3 &&
                       p.taps.dw[0] == -3 && p.taps.widx[48] == 48 && p.W % 2 == 0 && p.H % 2 == 0 && p.Wo * 2 == p.W && p.Ho * 2 == p.H && p.ldx == 3 &&
                       p.Cout == 64 && p.lddy % 16 == 0 && (reinterpret_cast<uintptr_t>(p.dy) & 63) == 0 && p.M < (1ll << 31) &&
                       (int64_t)p.B * p.H * p.W * 3 < (1ll << 31);
    const bool fuse_bias = form == 2 && dbias != nullptr && ws_bytes >= need + (size_t)splits * p.Cout * 4;
    p.bias_part = fuse_bias ? p.part + (size_t)splits * nt * cn : nullptr;
    if (nt != kh * kw)
        if (hipMemsetAsync(dw, 0, (size_t)kh * kw * cn * 4, st) != hipSuccess) return fail(PP_ERR_LAUNCH, "conv bwd_weight: memset failed");
    dim3 grid((unsigned)nblk), blk(256);
    if (stem) {
        hipLaunchKernelGGL((wgrad_stem3x3s2_kernel<4>), grid, blk, 0, st, p, rows_per_split);
        splits = nblk;                                    // one slice of partial sums per block from here on
    } else if (stem7) {
        // <= 2048 blocks of three waves, at least 32 output pixels each; one slice of partial sums per block (never more than the generic
        // form's slices, which sized the workspace)
        int64_t nb7 = std::min<int64_t>(2048, cdiv(p.M, 32));
        if (nb7 > splits) nb7 = splits;
        const int64_t rpb = cdiv(p.M, nb7);
        nb7 = cdiv(p.M, rpb);
        hipLaunchKernelGGL(wgrad_stem7x7s2_kernel, dim3((unsigned)nb7), dim3(192), 0, st, p, rpb);
        splits = nb7;
    } else if (form == 1) {
        if (p.Cin == 3 && nt == 9)        hipLaunchKernelGGL((wgrad_narrow_in_kernel<3, 9, 4>), grid, blk, 0, st, p, NL, RL, rows_per_split);
        else if (p.Cin == 3 && nt == 49)  // the 7x7 stems (resnet_models.py:115-117): one row of seven taps per grid.y slice
            hipLaunchKernelGGL((wgrad_narrow_in_kernel<3, 7, 4>), dim3((unsigned)nblk, 7), blk, 0, st, p, NL, RL, rows_per_split);
        else if (p.Cin == 16)             hipLaunchKernelGGL((wgrad_narrow_in_kernel<16, 1, 4>), grid, blk, 0, st, p, NL, RL, rows_per_split);
        else                              hipLaunchKernelGGL((wgrad_narrow_in_kernel<32, 1, 4>), grid, blk, 0, st, p, NL, RL, rows_per_split);
    } else {
        if (p.Cout == 16)       hipLaunchKernelGGL((wgrad_narrow_out_kernel<16, 4>), grid, blk, 0, st, p, NL, RL, rows_per_split);
        else if (p.Cout == 19)  hipLaunchKernelGGL((wgrad_narrow_out_kernel<19, 4>), grid, blk, 0, st, p, NL, RL, rows_per_split);
        else if (p.Cout == 24)  hipLaunchKernelGGL((wgrad_narrow_out_kernel<24, 4>), grid, blk, 0, st, p, NL, RL, rows_per_split);
        else                    hipLaunchKernelGGL((wgrad_narrow_out_kernel<32, 4>), grid, blk, 0, st, p, NL, RL, rows_per_split);
    }
    if (int rc = check_launch("wgrad_narrow_kernel")) return rc;
    if (reduce_deferrable(job, p.taps, cn, splits, dbias)) {
        fill_reduce_job(job, 2, p.part, dw, cn, splits, p.taps);
        *bias_done = true;                   // (no bias gradient asked for)
        return PP_OK;
    }
    hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3((unsigned)cdiv((int64_t)nt * cn, 8)), dim3(256), 0, st, p.part, (int)splits,
                       nt, cn, p.taps, dw);
    if (int rc = check_launch("wgrad_reduce_wide_kernel")) return rc;
    if (fuse_bias) {
        hipLaunchKernelGGL(bias_grad_final_kernel, dim3((unsigned)cdiv(p.Cout, 8)), dim3(256), 0, st, p.bias_part, (int)splits, p.Cout, dbias);
        if (int rc = check_launch("bias_grad_final_kernel")) return rc;
        *bias_done = true;
    }
    return 0;
}

// sizes whose 32-bit output-size arithmetic (in + 2 pad - dil (k - 1) - 1) cannot overflow (found by tests/test_abi_asan.py: a hostile
// kernel size wrapped kh * kw below the tap table's bound and build_taps wrote past it)
static inline bool conv_geom_sane(int H, int W, int stride, int pad, int dil)
{
    return H <= (1 << 24) && W <= (1 << 24) && stride <= (1 << 16) && pad >= 0 && pad <= (1 << 20) && dil <= (1 << 16);
}

static int conv_common_check(const void* a, const void* b, const void* c, int B, int H, int W, int Cin, int Cout,
                             int kh, int kw, int stride, int pad, int dil)
{
    if (!a || !b || !c) return fail(PP_ERR_BAD_ARG, "conv: null pointer");
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return fail(PP_ERR_BAD_ARG, "conv: bad shape");
    if (kh < 1 || kw < 1 || kh > kMaxTaps || kw > kMaxTaps || kh * kw > kMaxTaps) return fail(PP_ERR_UNSUPPORTED, "conv: kernel %dx%d unsupported", kh, kw);
    if (stride < 1 || dil < 1 || pad < 0 || !conv_geom_sane(H, W, stride, pad, dil)) return fail(PP_ERR_BAD_ARG, "conv: bad stride/dilation/padding");
    return PP_OK;
}

static inline int out_size(int in, int k, int stride, int pad, int dil) { return (in + 2 * pad - dil * (k - 1) - 1) / stride + 1; }

}  // namespace pp

using namespace pp;

extern "C" {

/* tools/conv_layer_table.py: the plan the forward / backward-data launcher picks for an (M x Cn x K = ntaps*Ck) problem:
 * out = {tile rows, tile cols, tiles, split-K slices} */
#ifdef PP_DEBUG_KNOBS
void pp_debug_conv_plan(int64_t M, int Cn, int Ck, int ntaps, int* out)
{
    const ConvPlan pl = plan_conv(M, Cn, Ck, ntaps, Ck % 4 == 0 && Cn % 4 == 0);
    const int rows[5] = {128, 128, 64, 64, 128}, cols[5] = {32, pl.bn64 ? 64 : 128, 64, 64, 128};
    out[0] = rows[pl.cfg]; out[1] = cols[pl.cfg]; out[2] = (int)pl.tiles; out[3] = pl.splits;
}
#endif

/* A/B of the split-K plan: v = tiles_threshold | target_blocks << 10 | min_k_steps << 20 | min_steps_per_slice << 26 (0: defaults) */
#ifdef PP_DEBUG_KNOBS
void pp_debug_set_splitk(int v)
{
    g_splitk_tiles = (v & 1023) ? (v & 1023) : 192;
    g_splitk_target = ((v >> 10) & 1023) ? ((v >> 10) & 1023) : 512;
    g_splitk_min_nk = ((v >> 20) & 63) ? ((v >> 20) & 63) : 12;
    g_splitk_min_iters = ((v >> 26) & 15) ? ((v >> 26) & 15) : 4;
}
#endif

/* whole-row VALU kernels of the narrow pointwise layers (A/B): bit 0 off, bit 1 forward rows kernel only from 65536 rows */
#ifdef PP_DEBUG_KNOBS
void pp_debug_set_conv_rows(int bits) { g_conv_bwd_rows = (bits & 1) ? 0 : 1; g_rows_fwd_min = (bits & 2) ? 65536 : 16384; }
#endif

/* which shapes pp_conv2d_fwd_bn_train_ok / pp_conv2d_bwd_data_bn_bwd_ok accept (A/B): bit 0 tiled forward kernels, 1 in-block split-K
 * forward, 2 backward form (64x64 tiles), 3 backward form of the in-block split-K and the 128x32 kernels; default 15 */
#ifdef PP_DEBUG_KNOBS
void pp_debug_set_conv_bn_fuse(int bits) { g_conv_bn_fuse = bits & 15; }
#endif

/* 0: fp32 MFMA kernels everywhere; 1 (default): the large-tile forward / backward-data layers run conv_x3_kernel (bf16x3 split) */
/* Yardstick (bench.py roofline_mfma.sustained): nothing but conv_x3_kernel's MFMA stream - six product terms of three A x three B
 * register-resident bf16 fragments on four accumulators, eight waves per CU - with operands of a chosen switching activity (0: zeros,
 * 1: near-constant, 2: hash-random signs / mantissas as the planes of a real activation have).  Same instructions in all three; what
 * differs is the clock the power limit leaves (profiles/r05_conv_x3_power.txt: 2.39-2.5 GHz on zeros, 1.77-1.92 GHz on random
 * operands).  iters x 24 MFMAs per wave. */
int pp_yardstick_mfma_stream(int data_kind, int iters, float* sink, pp_stream_t stream)
{
    if (!sink || iters < 1 || data_kind < 0 || data_kind > 2) return fail(PP_ERR_BAD_ARG, "mfma_stream: bad argument");
    const dim3 grid((unsigned)device_cus()), block(512);
    hipStream_t st = as_stream(stream);
    EventScope ev(st);
    if (data_kind == 0) hipLaunchKernelGGL(mfma_stream_kernel<0>, grid, block, 0, st, sink, iters, 7u);
    else if (data_kind == 1) hipLaunchKernelGGL(mfma_stream_kernel<1>, grid, block, 0, st, sink, iters, 7u);
    else hipLaunchKernelGGL(mfma_stream_kernel<2>, grid, block, 0, st, sink, iters, 7u);
    return check_launch("mfma_stream_kernel");
}

#ifdef PP_DEBUG_KNOBS
void pp_debug_set_gemm_pw(int v)
{
    g_gemm_pw = (v & 15) <= 7 ? (v & 15) : 1;                 // 0 off, 1 rule, 2..7 force tile form 0..5
    g_gemm_pw_rows_min = (v >> 4) > 0 ? (v >> 4) : 4096;      // bits 4..: least rows
}
#endif
#ifdef PP_DEBUG_KNOBS
void pp_debug_set_x3f(int v)
{
    g_x3f = (v & 1) ? 1 : 0;
    static const double gf[8] = {1e9, 0.5e9, 2e9, 4e9, 8e9, 0.25e9, 0.0, 1e9};
    g_x3f_flop = gf[(v >> 1) & 7];
    static const int tl[4] = {128, 64, 192, 256};
    g_x3f_tiles = tl[(v >> 4) & 3];
    static const int kk[4] = {256, 128, 512, 16};
    g_x3f_k = kk[(v >> 6) & 3];
}
#endif
#ifdef PP_DEBUG_KNOBS
void pp_debug_set_x3_variant(int v) { g_x3_var = (v >= 0 && v <= 11) ? v : 0; }
#endif

#ifdef PP_DEBUG_KNOBS
void pp_debug_set_x3(int v)
{
    g_conv_x3 = v & 0xFF;
    g_conv_x3_mid = (v & 256) ? 0 : 1;                     /* bit 8: large-tile plans only */
    static const double gf[8] = {16e9, 12e9, 8e9, 6e9, 4e9, 3e9, 2e9, 1e9};
    g_x3_mid_flop = gf[(v >> 9) & 7];                      /* bits 9-11: least work of a mid-size layer */
    static const int tl[4] = {256, 128, 64, 32};
    g_x3_mid_tiles = tl[(v >> 12) & 3];                    /* bits 12-13: least 128 x 128 tiles of a mid-size layer */
    static const double wf[8] = {8e9, 6e9, 4e9, 3e9, 2e9, 1.5e9, 1e9, 0.5e9};
    g_x3w_flop = wf[(v >> 14) & 7];                        /* bits 14-16: least work of a bf16x3 weight gradient */
}
#endif

#ifdef PP_DEBUG_KNOBS
void pp_debug_set_wgrad_target(int v)
{
    g_wgrad_target = (v & 0xFFFF) > 0 ? (v & 0xFFFF) : 1024;
    g_wgrad_lds_pad = v > 0 ? ((v >> 16) & 0xFF) * 1024 : g_wgrad_lds_pad_default;   // bits 16-23: KiB of LDS padding (A/B)
    g_wgrad_balance = (v > 0 && ((v >> 24) & 1)) ? 0 : 1;                             // bit 24: CU-balanced split choice off
    g_wgrad_fold = (v > 0 && ((v >> 25) & 1)) ? 1 : 0;                                // bit 25: split-K reduction folded into the launch (experiment)
}
#endif
/* v = big_tile_min | wgrad_rows_min << 12 (0 fields: defaults 384 / 128) */
#ifdef PP_DEBUG_KNOBS
void pp_debug_set_conv_thresholds(int v)
{
    g_shortk64 = (v >> 28) & 1 ? 0 : 1;       // bit 28: 128 x 128 tiles also for pointwise layers with K <= 256 (A/B)
    g_big_tile_min = (v & 4095) ? (v & 4095) : 384;
    g_wgrad_rows_min = ((v >> 12) & 4095) ? ((v >> 12) & 4095) : 128;
    const int ns = (v >> 24) & 15;           // bits 24-27: narrow-layer weight gradient, 256 << (ns-1) splits at most, 2048 / max rows at least
    g_narrow_splits_max = ns ? (256 << (ns - 1)) : 4096;
    g_narrow_rows_min = ns ? (ns >= 4 ? 16 : 64) : 16;
}
#endif

#ifdef PP_DEBUG_KNOBS
void pp_debug_set_conv_variant(int v)
{
    g_conv_xcd_remap = (v & 4) ? 0 : 1;      // bit 2 switches the XCD-aware tile order off (A/B)
    g_conv_novec = (v & 8) ? 1 : 0;          // bit 3 forces the conditional-load path (A/B)
    g_conv_lds_pad = (v & 16) ? 40 * 1024 : ((v & 32) ? 70 * 1024 : 0);   // bits 4/5: at most 2 / 1 blocks per CU
    g_conv_splitk = (v & 64) ? 0 : 1;        // bit 6 switches split-K off (A/B)
    g_conv_n64 = (v & 2048) ? 0 : 1;         // bit 11: 64-wide tiles for ragged output widths off (A/B)
    g_conv_tap_inner = (v & 16384) ? 0 : 1;  // bit 14: tap-outer K order of the forward / backward-data kernel (A/B)
    g_wgrad_xcd = (v & 8192) ? 0 : 1;        // bit 13: XCD-aware block order of the weight-gradient kernel off (A/B)
    g_wgrad_m64 = (v & 1024) ? 0 : 1;        // bit 10: 64-row weight-gradient tiles for ragged Cin off (A/B)
    g_wgrad_narrow = (v & 512) ? 0 : 1;      // bit 9 switches the narrow-layer weight-gradient kernels off (A/B)
    g_wgrad_stem = (v & 8388608) ? 0 : 1;    // bit 23: specialised MobileNetV2-stem weight gradient off (A/B)
    g_conv_ablate_reduce = (v >> 16) & 3;    // bits 16/17: timing-only ablation, see above
    g_conv_dma = (v & 256) ? 0 : ((v & 32768) ? 3 : ((v & 4194304) ? 2 : 1));   // bit 15: backward-data only for the 128x64 tiles; bit 22: forward only   // bit 8: LDS-DMA kernel of the 128x128 tiles off; bit 15: also for backward-data
    g_wgrad_dma = ((v >> 20) & 1 ? 0 : 1) | ((v >> 21) & 1 ? 2 : 0);   // bit 20: LDS-DMA weight-gradient kernel of the 128-wide tiles off; bit 21: 64x64 tiles on
    g_conv_dma64 = (v & 262144) ? 0 : ((v & 524288) ? 2 : 1);   // bit 18: LDS-DMA kernel of the 64x64 tiles off; bit 19: forward only
    g_conv_big_bk32 = (v & 4096) ? 1 : 0;    // bit 12: 32-deep K step for the 128x128 tiles (A/B)
    g_conv_deepk = (v & 128) ? 0 : 1;        // bit 7 switches the 64-deep K step of the 64x64 kernel off (A/B)
    g_bwd_phases = (v & 16777216) ? 0 : 1;      // bit 24: strided backward-data as one masked-tap launch instead of s*s phase problems
    g_conv_ksplit = (v & (1 << 25)) ? 0 : 1 + ((v >> 26) & 7);   // bit 25: in-block split-K LDS-DMA 1x1 kernel off; bits 26-28: force tile candidate 1..3 (0: the rule)
    { const int kc[4] = {256, 768, 512, 384}; g_ksplit_k_min = kc[(v >> 29) & 3]; }   // bits 29-30: K threshold 256 (default) / 768 / 512 / 384 (A/B)
    v &= 3;
    g_conv_variant = (v >= 0 && v <= 2) ? v : 0;
}
#endif

size_t pp_conv2d_fwd_workspace_bytes(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil)
{
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || kh < 1 || kw < 1 || kh > kMaxTaps || kw > kMaxTaps || kh * kw > kMaxTaps || !conv_geom_sane(H, W, stride, pad, dil) || stride < 1 || dil < 1) return 0;
    const int Ho = out_size(H, kh, stride, pad, dil), Wo = out_size(W, kw, stride, pad, dil);
    if (Ho < 1 || Wo < 1) return 0;
    ConvTaps t;
    build_taps(t, kh, kw, stride, pad, dil, H, W, Ho, Wo, false);
    const int64_t M = (int64_t)B * Ho * Wo;
    if (ksplit_shape_ok(M, Cout, Cin, t.n, stride)) return 0;             // in-block split-K: no partial sums leave the block
    const bool vec = Cin % 4 == 0 && Cout % 4 == 0;
    const ConvPlan pl = plan_conv(M, Cout, Cin, t.n, vec);
    const X3Plan x3 = x3_plan(pl, M, (int64_t)B * H * W, Cin, Cout, kh * kw, t.n, vec);
    if (x3.ok) return x3.bytes;                                           // bf16x3 operand planes
    return pl.splits > 1 ? align_up((size_t)pl.splits * M * Cout * 4, 256) : 0;
}

int64_t pp_conv2d_fwd_stats_rows(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil)
{
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || kh < 1 || kw < 1 || kh > kMaxTaps || kw > kMaxTaps || kh * kw > kMaxTaps || !conv_geom_sane(H, W, stride, pad, dil) || stride < 1 || dil < 1) return 0;
    const int Ho = out_size(H, kh, stride, pad, dil), Wo = out_size(W, kw, stride, pad, dil);
    if (Ho < 1 || Wo < 1) return 0;
    ConvTaps t;
    build_taps(t, kh, kw, stride, pad, dil, H, W, Ho, Wo, false);
    const int64_t M = (int64_t)B * Ho * Wo;
    const bool vec = Cin % 4 == 0 && Cout % 4 == 0;
    ConvPlan pl = plan_conv(M, Cout, Cin, t.n, vec);
    if (ksplit_shape_ok(M, Cout, Cin, t.n, stride)) pl.splits = 1;      // no workspace is offered for these shapes: single pass (see conv2d_fwd_impl)
    int64_t rows = conv_stats_rows(pl, M, Cout);
    // a layer the bf16x3 kernels may serve (they do when the caller passes the workspace): their wave rows are 64 rows of a 128- or
    // 256-row tile, whatever tile the fp32 plan has - room for both row sets; conv2d_fwd_impl zero-fills the buffer for such a layer
    if (rows > 0 && x3_plan(pl, M, (int64_t)B * H * W, Cin, Cout, kh * kw, t.n, vec).ok) rows = std::max<int64_t>(rows, cdiv(M, 256) * 4);
    return rows;
}

// ---- backward-data of a STRIDED convolution by phases -------------------------------------------------------------------------
// dX(ih, iw) = sum_t dY((ih + pad - th*dil) / s, (iw + pad - tw*dil) / s) W[t] where both divisions are exact.  Run over all of
// dX with every tap masked by divisibility the kernel does s*s times the useful MACs (a 1x1 stride-2 downsample: three of four
// dX pixels are zero; a 3x3 stride-2: 9/4 live taps per pixel on average - resnet_models.py:63-64,142-144).  Instead the s*s
// pixel classes (ih % s, iw % s) = (ph, pw) are s*s ordinary stride-1 problems: class row a <-> ih = a*s + ph reads
// dY(a + (ph + pad - th*dil)/s, .) through exactly the taps whose offset divides, writes a dense [B, Ha, Wb, Cin] slab, and one
// interleave pass puts the slabs (zeros for a class without taps) into dX.  Same sums, tap by tap in the same order.
struct PhaseGeom { int Ha, Wb; ConvTaps taps; };

static void bwd_phase_geom(PhaseGeom& g, int ph, int pw, int kh, int kw, int stride, int pad, int dil, int H, int W, int Ho, int Wo)
{
    g.Ha = ph < H ? (H - ph + stride - 1) / stride : 0;
    g.Wb = pw < W ? (W - pw + stride - 1) / stride : 0;
    g.taps.n = 0;
    for (int th = 0; th < kh; ++th)
        for (int tw = 0; tw < kw; ++tw) {
            const int vh = ph + pad - th * dil, vw = pw + pad - tw * dil;
            if (((vh % stride) + stride) % stride != 0 || ((vw % stride) + stride) % stride != 0) continue;
            const int dh = vh >= 0 ? vh / stride : -((-vh) / stride), dw = vw >= 0 ? vw / stride : -((-vw) / stride);
            // live: some class row / column lands inside dY
            if (!(dh < Ho && dh + g.Ha - 1 >= 0 && dw < Wo && dw + g.Wb - 1 >= 0)) continue;
            g.taps.dh[g.taps.n] = dh;
            g.taps.dw[g.taps.n] = dw;
            g.taps.widx[g.taps.n] = th * kw + tw;
            ++g.taps.n;
        }
}

struct PhaseSlabs { const float* slab[16]; int Ha[16], Wb[16]; };

// dX[b, a*s + ph, bb*s + pw, :] (+)= slab[ph*s + pw][b, a, bb, :]  (a NULL slab is all zeros); one float4 per thread
__global__ __launch_bounds__(256) void bwd_phase_interleave_kernel(PhaseSlabs ps, int s, int B, int H, int W, int cq, float* dx, int64_t lddx,
                                                                   int accumulate)
{
    const int64_t total = (int64_t)B * H * W * cq;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int q = (int)(e % cq);
        int64_t t = e / cq;
        const int iw = (int)(t % W); t /= W;
        const int ih = (int)(t % H);
        const int b = (int)(t / H);
        const int ph = ih % s, pw = iw % s, k = ph * s + pw;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ps.slab[k]) v = *reinterpret_cast<const float4*>(ps.slab[k] + ((((int64_t)b * ps.Ha[k] + ih / s) * ps.Wb[k] + iw / s) * cq + q) * 4);
        float* o = dx + (((int64_t)b * H + ih) * W + iw) * lddx + q * 4;
        if (accumulate) {
            const float4 d = *reinterpret_cast<const float4*>(o);
            v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w;
        }
        *reinterpret_cast<float4*>(o) = v;
    }
}

static bool bwd_phases_apply(int Cin, int Cout, int stride, int64_t lddx)
{
    return g_bwd_phases && stride > 1 && stride <= 4 && Cin % 4 == 0 && Cout % 4 == 0 && lddx % 4 == 0;
}

// workspace of the phase form: the slabs (as many floats as dX has) followed by the largest split-K need of a phase problem
static size_t bwd_phases_workspace(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil, int Ho, int Wo,
                                   size_t* slab_bytes)
{
    size_t split_need = 0;
    for (int ph = 0; ph < stride; ++ph)
        for (int pw = 0; pw < stride; ++pw) {
            PhaseGeom g;
            bwd_phase_geom(g, ph, pw, kh, kw, stride, pad, dil, H, W, Ho, Wo);
            if (g.taps.n == 0 || g.Ha == 0 || g.Wb == 0) continue;
            const int64_t M = (int64_t)B * g.Ha * g.Wb;
            const ConvPlan pl = plan_conv(M, Cin, Cout, g.taps.n, true);
            if (pl.splits > 1) split_need = std::max(split_need, align_up((size_t)pl.splits * M * Cin * 4, 256));
        }
    *slab_bytes = align_up((size_t)B * H * W * Cin * 4, 256);
    return *slab_bytes + split_need;
}

size_t pp_conv2d_bwd_data_workspace_bytes(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil)
{
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || kh < 1 || kw < 1 || kh > kMaxTaps || kw > kMaxTaps || kh * kw > kMaxTaps || !conv_geom_sane(H, W, stride, pad, dil) || stride < 1 || dil < 1) return 0;
    const int Ho = out_size(H, kh, stride, pad, dil), Wo = out_size(W, kw, stride, pad, dil);
    if (Ho < 1 || Wo < 1) return 0;
    ConvTaps t;
    build_taps(t, kh, kw, 1, pad, dil, Ho, Wo, H, W, true, stride);
    const int64_t M = (int64_t)B * H * W;
    const ConvPlan pl = plan_conv(M, Cin, Cout, t.n, Cin % 4 == 0 && Cout % 4 == 0);
    size_t need = pl.splits > 1 ? align_up((size_t)pl.splits * M * Cin * 4, 256) : 0;
    if (ksplit_shape_ok(M, Cin, Cout, t.n, stride)) need = 0;            // in-block split-K (stride 1 only)
    if (stride == 1) {
        const X3Plan x3 = x3_plan(pl, M, (int64_t)B * Ho * Wo, Cout, Cin, kh * kw, t.n, Cin % 4 == 0 && Cout % 4 == 0);
        if (x3.ok) need = std::max(need, x3.bytes);
    }
    if (bwd_phases_apply(Cin, Cout, stride, 4)) {
        size_t slabs = 0;
        need = std::max(need, bwd_phases_workspace(B, H, W, Cin, Cout, kh, kw, stride, pad, dil, Ho, Wo, &slabs));
    }
    return need;
}

static int conv2d_fwd_impl(const float* x, int64_t ldx, int B, int H, int W, int Cin, const float* w, const float* bias,
                           int kh, int kw, int stride, int pad, int dil, float* y, int64_t ldy, int Cout, const Epilogue& epi,
                           void* workspace, size_t ws_bytes, pp_stream_t stream, float* stats = nullptr, size_t stats_floats = 0,
                           const float* in_scale = nullptr, const float* in_shift = nullptr, int in_act = 0, const void* x_planes = nullptr,
                           const void* w_planes = nullptr)
{
    if (int rc = conv_common_check(x, w, y, B, H, W, Cin, Cout, kh, kw, stride, pad, dil)) return rc;
    if (ldx % 4 != 0 && Cin % 4 == 0) return fail(PP_ERR_BAD_ARG, "conv fwd: ldx must be a multiple of 4");
    const int Ho = out_size(H, kh, stride, pad, dil), Wo = out_size(W, kw, stride, pad, dil);
    if (Ho < 1 || Wo < 1) return fail(PP_ERR_BAD_ARG, "conv fwd: empty output");
    ConvParams p{};
    p.x = x; p.w = w; p.bias = bias; p.y = y; p.ldx = ldx; p.ldy = ldy;
    p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.Ck = Cin; p.Cn = Cout; p.Cin = Cin; p.Cout = Cout;
    p.stride = stride; p.M = (int64_t)B * Ho * Wo; p.bwd_stride = 1;
    p.epi = epi;
    p.in_scale = in_scale; p.in_shift = in_shift; p.in_act = in_act;
    p.a_pre = reinterpret_cast<const uint16_t*>(x_planes);
    p.b_pre = reinterpret_cast<const uint16_t*>(w_planes);
    build_taps(p.taps, kh, kw, stride, pad, dil, H, W, Ho, Wo, false);
    if (p.taps.n == 0) return fail(PP_ERR_BAD_ARG, "conv fwd: no live tap");
    if (p.M > 0x7FFFFFFFll) return fail(PP_ERR_UNSUPPORTED, "conv fwd: more than 2^31 output pixels");
    if (stats) {
        const int64_t rows = pp_conv2d_fwd_stats_rows(B, H, W, Cin, Cout, kh, kw, stride, pad, dil);
        if (rows <= 0) return fail(PP_ERR_UNSUPPORTED, "conv fwd: this shape delivers no BatchNorm statistics");
        if (stats_floats < (size_t)rows * 2 * Cout) return fail(PP_ERR_WORKSPACE, "conv fwd: statistics buffer too small");
        const ConvPlan pl = plan_conv(p.M, Cout, Cin, p.taps.n, Cin % 4 == 0 && Cout % 4 == 0);
        p.stats = stats;
        if (x3_plan(pl, p.M, (int64_t)B * H * W, Cin, Cout, kh * kw, p.taps.n, Cin % 4 == 0 && Cout % 4 == 0).ok &&
            hipMemsetAsync(stats, 0, (size_t)rows * 2 * Cout * 4, as_stream(stream)) != hipSuccess)
            return fail(PP_ERR_LAUNCH, "conv fwd: clearing the statistics buffer failed");     // (the kernel that runs writes ITS rows)
        // layers the in-block split-K kernel serves get no workspace (pp_conv2d_fwd_workspace_bytes == 0) and that kernel
        // writes no statistics: with statistics requested they run the tiled kernel as ONE pass (rows = its wave rows)
        if (ksplit_shape_ok(p.M, Cout, Cin, p.taps.n, stride)) return launch_conv<false>(p, nullptr, 0, as_stream(stream));
        if (pl.splits > 1 && (!workspace || ws_bytes < (size_t)pl.splits * p.M * Cout * 4))
            return fail(PP_ERR_WORKSPACE, "conv fwd: the split-K workspace is required when statistics are requested");
    }
    return launch_conv<false>(p, workspace, ws_bytes, as_stream(stream), kh * kw);
}

int pp_conv2d_fwd_stats(const float* x, int64_t ldx, int B, int H, int W, int Cin, const float* w, const float* bias,
                        int kh, int kw, int stride, int pad, int dil, float* y, int64_t ldy, int Cout, void* workspace,
                        size_t ws_bytes, float* stats, size_t stats_floats, pp_stream_t stream)
{
    if (!stats) return fail(PP_ERR_BAD_ARG, "conv fwd: stats is null");
    return conv2d_fwd_impl(x, ldx, B, H, W, Cin, w, bias, kh, kw, stride, pad, dil, y, ldy, Cout, Epilogue{}, workspace,
                           ws_bytes, stream, stats, stats_floats);
}

int pp_conv2d_fwd_accepts_affine_in(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil)
{
    if (B < 1 || H < 1 || W < 1 || Cin < 4 || Cout < 1 || kh != 1 || kw != 1 || stride != 1 || pad != 0 || dil < 1) return 0;
    if (Cin % 4 || Cout % 4) return 0;
    const int64_t M = (int64_t)B * H * W;
    if (ksplit_shape_ok(M, Cout, Cin, 1, 1) && Cin <= 2048) return 1;
    return plan_conv(M, Cout, Cin, 1, true).cfg == 0 ? 1 : 0;
}

int pp_conv2d_fwd_affine_in(const float* x_raw, int64_t ldx, int B, int H, int W, int Cin, const float* in_scale, const float* in_shift,
                            int in_act, const float* w, const float* bias, int kh, int kw, int stride, int pad, int dil, float* y,
                            int64_t ldy, int Cout, void* workspace, size_t ws_bytes, pp_stream_t stream)
{
    if (!in_scale || !in_shift || in_act < 0 || in_act > 2) return fail(PP_ERR_BAD_ARG, "conv fwd: input affine");
    if (!pp_conv2d_fwd_accepts_affine_in(B, H, W, Cin, Cout, kh, kw, stride, pad, dil))
        return fail(PP_ERR_UNSUPPORTED, "conv fwd: this shape has no input-affine kernel");
    if ((reinterpret_cast<uintptr_t>(in_scale) | reinterpret_cast<uintptr_t>(in_shift)) & 15)
        return fail(PP_ERR_BAD_ARG, "conv fwd: input affine vectors must be 16-byte aligned");
    return conv2d_fwd_impl(x_raw, ldx, B, H, W, Cin, w, bias, kh, kw, stride, pad, dil, y, ldy, Cout, Epilogue{}, workspace, ws_bytes,
                           stream, nullptr, 0, in_scale, in_shift, in_act);
}

int pp_conv2d_fwd(const float* x, int64_t ldx, int B, int H, int W, int Cin, const float* w, const float* bias,
                  int kh, int kw, int stride, int pad, int dil, float* y, int64_t ldy, int Cout, void* workspace,
                  size_t ws_bytes, pp_stream_t stream)
{
    return conv2d_fwd_impl(x, ldx, B, H, W, Cin, w, bias, kh, kw, stride, pad, dil, y, ldy, Cout, Epilogue{}, workspace,
                           ws_bytes, stream);
}

int pp_conv2d_fwd_pre(const float* x, int64_t ldx, int B, int H, int W, int Cin, const float* w, const float* bias,
                      int kh, int kw, int stride, int pad, int dil, float* y, int64_t ldy, int Cout, void* workspace,
                      size_t ws_bytes, const void* x_planes, pp_stream_t stream)
{
    return conv2d_fwd_impl(x, ldx, B, H, W, Cin, w, bias, kh, kw, stride, pad, dil, y, ldy, Cout, Epilogue{}, workspace,
                           ws_bytes, stream, nullptr, 0, nullptr, nullptr, 0, x_planes);
}

// The weights' planes off the step's critical path: they change once per optimiser step, so a caller may split them at the START of a
// step on another stream (both layouts: transpose = 1 the forward's [tap][Cout][Kp(Cin)], 0 the backward-data's [tap][Cin][Kp(Cout)]) and
// hand them to every *_pre2 call of the step instead of paying x3_split_w_kernel in front of each convolution (4 x ~10 us per step for
// the two SegmentHead layers).  Ignored by calls that do not run a bf16x3 kernel.
size_t pp_x3_weight_planes_bytes(int kh_kw, int Cin, int Cout, int transpose)
{
    if (kh_kw < 1 || Cin < 1 || Cout < 1) return 0;
    const int n_rows = transpose ? Cout : Cin, K = transpose ? Cin : Cout;
    const int64_t Kp = cdiv(K, 16) * 16, rows = (int64_t)kh_kw * n_rows;
    return align_up((size_t)3 * (size_t)(rows + 1) * (size_t)Kp * 2, 256);
}

int pp_x3_split_weights(const float* w, int kh_kw, int Cin, int Cout, int transpose, void* planes, size_t planes_bytes, pp_stream_t stream)
{
    if (!w || !planes || kh_kw < 1 || Cin < 1 || Cout < 1) return fail(PP_ERR_BAD_ARG, "x3_split_weights: null / empty");
    if (reinterpret_cast<uintptr_t>(planes) & 255) return fail(PP_ERR_BAD_ARG, "x3_split_weights: planes must be 256-byte aligned");
    if (planes_bytes < pp_x3_weight_planes_bytes(kh_kw, Cin, Cout, transpose)) return fail(PP_ERR_WORKSPACE, "x3_split_weights: planes buffer");
    const int n_rows = transpose ? Cout : Cin, K = transpose ? Cin : Cout;
    const int Kp = (int)cdiv(K, 16) * 16;
    const int64_t rows = (int64_t)kh_kw * n_rows, plane = (rows + 1) * Kp;
    if (plane * 2 >= (1ll << 32) - 4096) return fail(PP_ERR_UNSUPPORTED, "x3_split_weights: plane of %lld bytes", (long long)(plane * 2));
    hipStream_t st = as_stream(stream);
    EventScope ev(st);
    const int64_t tb = (rows + 1) * (Kp / 8);
    hipLaunchKernelGGL(x3_split_w_kernel, dim3((unsigned)std::min<int64_t>(cdiv(tb, 256), 4096)), dim3(256), 0, st, w, kh_kw, Cin, Cout, transpose ? 1 : 0,
                       reinterpret_cast<uint16_t*>(planes), Kp, plane);
    return check_launch("x3_split_w_kernel");
}

int pp_conv2d_fwd_pre2(const float* x, int64_t ldx, int B, int H, int W, int Cin, const float* w, const float* bias,
                       int kh, int kw, int stride, int pad, int dil, float* y, int64_t ldy, int Cout, void* workspace,
                       size_t ws_bytes, const void* x_planes, const void* w_planes, pp_stream_t stream)
{
    return conv2d_fwd_impl(x, ldx, B, H, W, Cin, w, bias, kh, kw, stride, pad, dil, y, ldy, Cout, Epilogue{}, workspace,
                           ws_bytes, stream, nullptr, 0, nullptr, nullptr, 0, x_planes, w_planes);
}

// ---- convolution + training BatchNorm (+ residual, activation) in one launch -----------------------------------------------------
static bool conv_bn_shape(ConvParams& p, ConvPlan& pl, int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil,
                          int64_t ldx, int* R_out = nullptr)
{
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || kh < 1 || kw < 1 || kh > kMaxTaps || kw > kMaxTaps || kh * kw > kMaxTaps || !conv_geom_sane(H, W, stride, pad, dil) || stride < 1 || dil < 1) return false;
    const int Ho = out_size(H, kh, stride, pad, dil), Wo = out_size(W, kw, stride, pad, dil);
    if (Ho < 1 || Wo < 1) return false;
    p = ConvParams{};
    p.ldx = ldx; p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.Ck = Cin; p.Cn = Cout; p.Cin = Cin; p.Cout = Cout;
    p.stride = stride; p.M = (int64_t)B * Ho * Wo; p.bwd_stride = 1;
    build_taps(p.taps, kh, kw, stride, pad, dil, H, W, Ho, Wo, false);
    if (p.M > 0x7FFFFFFFll || p.taps.n == 0) return false;
    const bool vec = Cin % 4 == 0 && Cout % 4 == 0 && ldx % 4 == 0;
    pl = plan_conv(p.M, Cout, Cin, p.taps.n, vec);
    const BnFusePlan f = bn_fuse_plan(p, pl, vec);
    if (R_out) *R_out = f.R;
    return f.kind != 0;
}

int pp_conv2d_fwd_bn_train_ok(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil)
{
    ConvParams p; ConvPlan pl;
    return conv_bn_shape(p, pl, B, H, W, Cin, Cout, kh, kw, stride, pad, dil, Cin) ? 1 : 0;
}

size_t pp_conv2d_fwd_bn_train_xchg_bytes(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil)
{
    ConvParams p; ConvPlan pl;
    int R = 0;
    if (!conv_bn_shape(p, pl, B, H, W, Cin, Cout, kh, kw, stride, pad, dil, Cin, &R)) return 0;
    return align_up((size_t)cdiv(Cout, 32) * (size_t)R * 64 * 8, 256);
}

int pp_conv2d_fwd_bn_train(const float* x, int64_t ldx, int B, int H, int W, int Cin, const float* w, int kh, int kw, int stride, int pad,
                           int dil, float* conv_out, int64_t ldc, const float* gamma, const float* beta, float eps, float momentum,
                           float* running_mean, float* running_var, float* mean, float* invstd, const float* residual, int64_t ldr,
                           int act, float* y, int64_t ldy, int Cout, void* xchg, size_t xchg_bytes, int32_t* sync, size_t sync_ints,
                           pp_stream_t stream)
{
    if (int rc = conv_common_check(x, w, conv_out, B, H, W, Cin, Cout, kh, kw, stride, pad, dil)) return rc;
    if (!gamma || !beta || !mean || !invstd || !y || !xchg || !sync) return fail(PP_ERR_BAD_ARG, "conv fwd + BatchNorm: null");
    if (act < 0 || act > 2) return fail(PP_ERR_BAD_ARG, "conv fwd + BatchNorm: act %d", act);
    if ((reinterpret_cast<uintptr_t>(xchg) & 7) != 0 || sync_ints < 2) return fail(PP_ERR_BAD_ARG, "conv fwd + BatchNorm: exchange area");
    ConvParams p; ConvPlan pl;
    int R = 0;
    if (!conv_bn_shape(p, pl, B, H, W, Cin, Cout, kh, kw, stride, pad, dil, ldx, &R))
        return fail(PP_ERR_UNSUPPORTED, "conv fwd + BatchNorm: this shape has no fused kernel (ask pp_conv2d_fwd_bn_train_ok first)");
    const size_t need = (size_t)cdiv(Cout, 32) * (size_t)R * 64 * 8;
    if (xchg_bytes < need) return fail(PP_ERR_WORKSPACE, "conv fwd + BatchNorm: exchange area %zu < %zu", xchg_bytes, need);
    p.x = x; p.w = w; p.bias = nullptr; p.y = conv_out; p.ldy = ldc;
    p.bn = BnTrain{gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, residual, ldr, act, y, ldy,
                   reinterpret_cast<xword*>(xchg), sync, 0};
    return launch_conv<false>(p, nullptr, 0, as_stream(stream), kh * kw);
}

int pp_conv2d_fwd_bn_act(const float* x, int64_t ldx, int B, int H, int W, int Cin, const float* w, const float* bias,
                         int kh, int kw, int stride, int pad, int dil, const float* gamma, const float* beta,
                         const float* running_mean, const float* running_var, float eps, const float* residual,
                         int64_t ldr, int act, float* y, int64_t ldy, int Cout, void* workspace, size_t ws_bytes,
                         pp_stream_t stream)
{
    if (gamma && (!beta || !running_mean || !running_var)) return fail(PP_ERR_BAD_ARG, "conv fwd_bn_act: incomplete BatchNorm");
    if (act < 0 || act > 2) return fail(PP_ERR_BAD_ARG, "conv fwd_bn_act: act %d", act);
    Epilogue e{gamma, beta, running_mean, running_var, eps, residual, ldr, act};
    return conv2d_fwd_impl(x, ldx, B, H, W, Cin, w, bias, kh, kw, stride, pad, dil, y, ldy, Cout, e, workspace, ws_bytes, stream);
}

static int conv2d_bwd_data_impl(const float* dy, int64_t lddy, int B, int Ho, int Wo, int Cout, const float* w, int kh, int kw,
                                int stride, int pad, int dil, float* dx, int64_t lddx, int H, int W, int Cin, int accumulate,
                                void* workspace, size_t ws_bytes, pp_stream_t stream, const void* dy_planes, const void* w_planes = nullptr)
{
    if (int rc = conv_common_check(dy, w, dx, B, H, W, Cin, Cout, kh, kw, stride, pad, dil)) return rc;
    if (Ho != out_size(H, kh, stride, pad, dil) || Wo != out_size(W, kw, stride, pad, dil))
        return fail(PP_ERR_BAD_ARG, "conv bwd_data: inconsistent sizes");
    if (bwd_phases_apply(Cin, Cout, stride, lddx) && (int64_t)B * H * W <= 0x7FFFFFFFll) {
        size_t slab_bytes = 0;
        const size_t need = bwd_phases_workspace(B, H, W, Cin, Cout, kh, kw, stride, pad, dil, Ho, Wo, &slab_bytes);
        if (workspace && ws_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0 && (reinterpret_cast<uintptr_t>(dx) & 15) == 0) {
            hipStream_t st = as_stream(stream);
            float* slabs = reinterpret_cast<float*>(workspace);
            char* split_ws = reinterpret_cast<char*>(workspace) + slab_bytes;
            PhaseSlabs ps{};
            size_t off = 0;
            for (int ph = 0; ph < stride; ++ph)
                for (int pw = 0; pw < stride; ++pw) {
                    const int k = ph * stride + pw;
                    PhaseGeom g;
                    bwd_phase_geom(g, ph, pw, kh, kw, stride, pad, dil, H, W, Ho, Wo);
                    ps.slab[k] = nullptr; ps.Ha[k] = g.Ha; ps.Wb[k] = g.Wb;
                    if (g.taps.n == 0 || g.Ha == 0 || g.Wb == 0) continue;
                    ConvParams q{};
                    q.x = dy; q.w = w; q.bias = nullptr; q.y = slabs + off; q.ldx = lddy; q.ldy = Cin;
                    q.B = B; q.H = Ho; q.W = Wo; q.Ho = g.Ha; q.Wo = g.Wb; q.Ck = Cout; q.Cn = Cin; q.Cin = Cin; q.Cout = Cout;
                    q.stride = 1; q.M = (int64_t)B * g.Ha * g.Wb; q.bwd_stride = 1; q.accumulate = 0;
                    q.taps = g.taps;
                    if (int rc = launch_conv<true>(q, split_ws, ws_bytes - slab_bytes, st)) return rc;
                    ps.slab[k] = slabs + off;
                    off += (size_t)q.M * Cin;
                }
            const int64_t total = (int64_t)B * H * W * (Cin / 4);
            int64_t blocks = cdiv(total, 256 * 4);
            if (blocks > 8192) blocks = 8192;
            hipLaunchKernelGGL(bwd_phase_interleave_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ps, stride, B, H, W, Cin / 4, dx, lddx,
                               accumulate ? 1 : 0);
            return check_launch("bwd_phase_interleave_kernel");
        }
    }
    ConvParams p{};
    p.x = dy; p.w = w; p.bias = nullptr; p.y = dx; p.ldx = lddy; p.ldy = lddx;
    p.B = B; p.H = Ho; p.W = Wo; p.Ho = H; p.Wo = W; p.Ck = Cout; p.Cn = Cin; p.Cin = Cin; p.Cout = Cout;
    p.stride = 1; p.M = (int64_t)B * H * W; p.bwd_stride = stride; p.accumulate = accumulate ? 1 : 0;
    // dX(ih,iw) = sum_t dY((ih + pad - th*dil)/stride, (iw + pad - tw*dil)/stride) W[t]   (where divisible)
    build_taps(p.taps, kh, kw, 1, pad, dil, Ho, Wo, H, W, true, stride);
    if (p.M > 0x7FFFFFFFll) return fail(PP_ERR_UNSUPPORTED, "conv bwd_data: more than 2^31 pixels");
    if (p.taps.n == 0) return fail(PP_ERR_BAD_ARG, "conv bwd_data: no live tap");
    p.a_pre = stride == 1 ? reinterpret_cast<const uint16_t*>(dy_planes) : nullptr;
    p.b_pre = stride == 1 ? reinterpret_cast<const uint16_t*>(w_planes) : nullptr;
    return launch_conv<true>(p, workspace, ws_bytes, as_stream(stream), kh * kw);
}

int pp_conv2d_bwd_data(const float* dy, int64_t lddy, int B, int Ho, int Wo, int Cout, const float* w, int kh, int kw,
                       int stride, int pad, int dil, float* dx, int64_t lddx, int H, int W, int Cin, int accumulate,
                       void* workspace, size_t ws_bytes, pp_stream_t stream)
{
    return conv2d_bwd_data_impl(dy, lddy, B, Ho, Wo, Cout, w, kh, kw, stride, pad, dil, dx, lddx, H, W, Cin, accumulate, workspace, ws_bytes,
                                stream, nullptr);
}

// ---- backward-data of several convolutions that read ONE input, as one implicit GEMM --------------------------------------------------
// aspp.py:49-57: the four ASPP branches (1x1 and three dilated 3x3, stride 1, "same" padding) all convolve the encoder output, so
// its gradient is  dX = sum_b conv_bwd_data(dY_b, W_b)  - four launches, three split-K reduces and three adds in the per-layer form.
// With the branch gradients side by side in one [B,H,W,nb*Cout] buffer the sum IS one convolution whose reduction runs over
// (branch, tap, channel): the merged tap table carries each tap's channel offset into dY and its weight offset behind the lowest
// weight pointer (the branch weights are separate tensors; their distances must fit 31 bits of elements).  Dead taps (a dilation
// larger than the map) are dropped per branch as build_taps does.  Same products, summed in one fp32 accumulation per output instead of
// four rounded partial results added afterwards.
struct MultiBwd { ConvParams p; ConvPlan pl; bool ok; };
static MultiBwd multi_bwd_setup(int B, int H, int W, int Cin, int Cout, int nb, const int* k, const int* d, int64_t lddy)
{
    MultiBwd m{};
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || nb < 2 || nb > 4) return m;
    ConvParams& p = m.p;
    p.ldx = lddy; p.B = B; p.H = H; p.W = W; p.Ho = H; p.Wo = W; p.Ck = Cout; p.Cn = Cin; p.Cin = Cin; p.Cout = Cout;
    p.stride = 1; p.M = (int64_t)B * H * W; p.bwd_stride = 1; p.multi = 1;
    p.taps.n = 0;
    for (int b = 0; b < nb; ++b) {
        if (k[b] < 1 || (k[b] & 1) == 0 || d[b] < 1 || k[b] * k[b] > kMaxTaps) return m;
        ConvTaps t;
        build_taps(t, k[b], k[b], 1, d[b] * (k[b] - 1) / 2, d[b], H, W, H, W, true, 1);
        for (int i = 0; i < t.n; ++i) {
            if (p.taps.n >= 32) return m;
            p.taps.dh[p.taps.n] = t.dh[i]; p.taps.dw[p.taps.n] = t.dw[i];
            p.taps.widx[p.taps.n] = (b << 8) | t.widx[i];        // (branch, tap) until the pointers are known: multi_bwd_bind
            p.tap_coff[p.taps.n] = b * Cout;
            ++p.taps.n;
        }
    }
    if (p.taps.n == 0 || p.M > 0x7FFFFFFFll) return m;
    const bool vec = Cin % 4 == 0 && Cout % 4 == 0 && lddy % 4 == 0;
    m.pl = plan_conv(p.M, Cin, Cout, p.taps.n, vec);
    if (m.pl.cfg == 2 && m.pl.splits > 1) {
        // the merged reduction is 20 - 30 taps deep: aim at one full wave of blocks (4 per CU) rather than plan_conv's 512 - at the
        // BASELINE shape 160 tiles x 4 slices left the CUs with 2 or 3 blocks each (130 us; 6 slices: one even round)
        const int nk = p.taps.n * (int)cdiv(Cout, BK);
        int64_t sp = (int64_t)device_cus() * 4 / m.pl.tiles;
        if (sp > nk / g_splitk_min_iters) sp = nk / g_splitk_min_iters;
        if (sp > 32) sp = 32;
        if (sp > m.pl.splits) {
            m.pl.ks_per_split = (int)cdiv(nk, sp);
            m.pl.splits = (int)cdiv(nk, m.pl.ks_per_split);
        }
    }
    // the kernel that knows the merged addressing: 64 x 64 tiles through the LDS-DMA pipeline
    m.ok = vec && m.pl.cfg == 2 && g_conv_dma64 == 1 && (int64_t)B * H * W * lddy < (1ll << 31) - (1ll << 24);
    return m;
}

size_t pp_conv2d_bwd_data_multi_workspace_bytes(int B, int H, int W, int Cin, int Cout, int nb, int k0, int d0, int k1, int d1, int k2, int d2,
                                                int k3, int d3)
{
    const int k[4] = {k0, k1, k2, k3}, d[4] = {d0, d1, d2, d3};
    const MultiBwd m = multi_bwd_setup(B, H, W, Cin, Cout, nb, k, d, (int64_t)nb * Cout);
    if (!m.ok) return 0;                      // not offered for this shape: the caller keeps the per-layer calls
    return align_up(m.pl.splits > 1 ? (size_t)m.pl.splits * m.p.M * Cin * 4 : 256, 256);
}

int pp_conv2d_bwd_data_multi(const float* dy, int64_t lddy, int B, int H, int W, int Cout, int nb, const float* w0, int k0, int d0,
                             const float* w1, int k1, int d1, const float* w2, int k2, int d2, const float* w3, int k3, int d3, float* dx,
                             int64_t lddx, int Cin, int accumulate, void* workspace, size_t ws_bytes, pp_stream_t stream)
{
    const int k[4] = {k0, k1, k2, k3}, d[4] = {d0, d1, d2, d3};
    const float* w[4] = {w0, w1, w2, w3};
    if (!dy || !dx || lddy < (int64_t)nb * Cout || lddx < Cin) return fail(PP_ERR_BAD_ARG, "conv bwd_data multi: null / leading dimension");
    MultiBwd m = multi_bwd_setup(B, H, W, Cin, Cout, nb, k, d, lddy);
    if (!m.ok) return fail(PP_ERR_UNSUPPORTED, "conv bwd_data multi: not offered for this shape (ask the workspace query first)");
    const float* base = nullptr;
    for (int b = 0; b < nb; ++b) {
        if (!w[b]) return fail(PP_ERR_BAD_ARG, "conv bwd_data multi: null weight");
        if (!base || w[b] < base) base = w[b];
    }
    if ((reinterpret_cast<uintptr_t>(dy) & 15) || (reinterpret_cast<uintptr_t>(dx) & 15) || lddx % 4 != 0)
        return fail(PP_ERR_BAD_ARG, "conv bwd_data multi: dy / dx must be 16-byte aligned");
    ConvParams& p = m.p;
    for (int t = 0; t < p.taps.n; ++t) {
        const int b = p.taps.widx[t] >> 8, lt = p.taps.widx[t] & 255;
        const int64_t off = (w[b] - base) + (int64_t)lt * Cin * Cout;
        if ((reinterpret_cast<uintptr_t>(w[b]) & 15) || off + (int64_t)Cin * Cout >= (1ll << 31))
            return fail(PP_ERR_UNSUPPORTED, "conv bwd_data multi: the branch weights must be 16-byte aligned and within 8 GiB of each other");
        p.tap_woff[t] = (int)off;
        p.taps.widx[t] = 0;
    }
    const size_t need = m.pl.splits > 1 ? (size_t)m.pl.splits * p.M * Cin * 4 : 0;
    if (need && (!workspace || ws_bytes < need)) return fail(PP_ERR_WORKSPACE, "conv bwd_data multi: workspace %zu < %zu", ws_bytes, need);
    hipStream_t st = as_stream(stream);
    EventScope ev(st);
    p.x = dy; p.w = base; p.bias = nullptr; p.y = dx; p.ldy = lddx; p.accumulate = accumulate ? 1 : 0;
    p.xcd_remap = g_conv_xcd_remap;
    p.tap_inner = g_conv_tap_inner;
    p.n_tiles = m.pl.n_tiles;
    p.splits = m.pl.splits;
    p.ks_per_split = m.pl.ks_per_split;
    p.part = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL((conv_igemm_dma_kernel<64, 64, true>), dim3((unsigned)m.pl.tiles, (unsigned)m.pl.splits), dim3(kThreads), 0, st, p);
    if (int rc = check_launch("conv_igemm_dma_kernel<multi>")) return rc;
    if (m.pl.splits > 1) {
        int64_t nblk = cdiv(p.M * (p.Cn / 4), 256);
        if (nblk > 8192) nblk = 8192;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)nblk), dim3(256), 0, st, p.part, m.pl.splits, p.M, p.Cn, p.bias, p.y, p.ldy, p.epi,
                           p.accumulate);
        return check_launch("splitk_reduce_kernel");
    }
    return PP_OK;
}

// ---- bf16x3 operand planes shared between the calls of a layer ---------------------------------------------------------------------
// The forward's A operand (X) is also the weight gradient's, the backward-data's (dY) is the weight gradient's other one: the
// caller may split an activation ONCE (pp_x3_split: three chunk-major bf16 planes, the layout conv_x3_kernel / conv_wgrad_x3_kernel
// DMA from) and hand the planes to every call that reads it.  pp_conv2d_x3_planes_bytes tells whether a call would use them.
size_t pp_x3_planes_bytes(int64_t rows, int C)
{
    if (rows < 1 || C < 1) return 0;
    return align_up((size_t)3 * (size_t)(rows + 1) * (size_t)(cdiv(C, 16) * 16) * 2, 256);
}

int pp_x3_split(const float* x, int64_t ldx, int64_t rows, int C, void* planes, size_t planes_bytes, pp_stream_t stream)
{
    if (!x || !planes || rows < 1 || C < 1) return fail(PP_ERR_BAD_ARG, "x3_split: null / empty");
    if (C % 4 != 0 || ldx % 4 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(planes) & 255))
        return fail(PP_ERR_BAD_ARG, "x3_split: C and ld must be multiples of 4, x 16-byte and planes 256-byte aligned");
    const int Kp = (int)cdiv(C, 16) * 16;
    const int64_t plane = (rows + 1) * Kp;
    if (plane * 2 >= (1ll << 32) - 4096) return fail(PP_ERR_UNSUPPORTED, "x3_split: plane of %lld bytes", (long long)(plane * 2));
    if (planes_bytes < pp_x3_planes_bytes(rows, C)) return fail(PP_ERR_WORKSPACE, "x3_split: planes buffer");
    const int64_t ta = (rows + 1) * (Kp / 8);
    hipLaunchKernelGGL(x3_split_kernel, dim3((unsigned)std::min<int64_t>(cdiv(ta, 256), 4096)), dim3(256), 0, as_stream(stream), x, ldx, rows, C,
                       reinterpret_cast<uint16_t*>(planes), Kp, plane);
    return check_launch("x3_split_kernel");
}

// backward-data + the BatchNorm backward of the layer in front (see conv_epilogue_bn_bwd)
static bool conv_bn_bwd_shape(ConvParams& p, ConvPlan& pl, int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil,
                              int64_t lddy, int* R_out)
{
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || kh < 1 || kw < 1 || kh > kMaxTaps || kw > kMaxTaps || kh * kw > kMaxTaps || !conv_geom_sane(H, W, stride, pad, dil) || stride != 1 || dil < 1) return false;
    const int Ho = out_size(H, kh, stride, pad, dil), Wo = out_size(W, kw, stride, pad, dil);
    if (Ho < 1 || Wo < 1) return false;
    p = ConvParams{};
    p.ldx = lddy; p.B = B; p.H = Ho; p.W = Wo; p.Ho = H; p.Wo = W; p.Ck = Cout; p.Cn = Cin; p.Cin = Cin; p.Cout = Cout;
    p.stride = 1; p.M = (int64_t)B * H * W; p.bwd_stride = 1;
    build_taps(p.taps, kh, kw, 1, pad, dil, Ho, Wo, H, W, true, 1);
    if (p.M > 0x7FFFFFFFll || p.taps.n == 0) return false;
    const bool vec = Cin % 4 == 0 && Cout % 4 == 0 && lddy % 4 == 0;
    pl = plan_conv(p.M, Cin, Cout, p.taps.n, vec);
    const BnFusePlan f = bn_fuse_plan_bwd(p, pl, vec);
    if (R_out) *R_out = f.R;
    return f.kind != 0;
}

int pp_conv2d_bwd_data_bn_bwd_ok(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil)
{
    ConvParams p; ConvPlan pl;
    return conv_bn_bwd_shape(p, pl, B, H, W, Cin, Cout, kh, kw, stride, pad, dil, Cout, nullptr) ? 1 : 0;
}

size_t pp_conv2d_bwd_data_bn_bwd_xchg_bytes(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil)
{
    ConvParams p; ConvPlan pl;
    int R = 0;
    if (!conv_bn_bwd_shape(p, pl, B, H, W, Cin, Cout, kh, kw, stride, pad, dil, Cout, &R)) return 0;
    return align_up((size_t)cdiv(Cin, 32) * (size_t)R * 64 * 8, 256);
}

int pp_conv2d_bwd_data_bn_bwd(const float* dy, int64_t lddy, int B, int Ho, int Wo, int Cout, const float* w, int kh, int kw, int stride,
                              int pad, int dil, int H, int W, int Cin, const float* bn_x, int64_t ldbx, const float* mean,
                              const float* invstd, const float* gamma, const float* beta, int act, float* dgamma, float* dbeta,
                              float* dx_bn, int64_t lddx, const float* grad_in, int64_t ldgi, float* dres, int64_t lddr,
                              void* xchg, size_t xchg_bytes, int32_t* sync, size_t sync_ints, pp_stream_t stream)
{
    if (int rc = conv_common_check(dy, w, dx_bn, B, H, W, Cin, Cout, kh, kw, stride, pad, dil)) return rc;
    if (Ho != out_size(H, kh, stride, pad, dil) || Wo != out_size(W, kw, stride, pad, dil))
        return fail(PP_ERR_BAD_ARG, "conv bwd_data + BatchNorm bwd: inconsistent sizes");
    if (!bn_x || !mean || !invstd || !gamma || !beta || !dgamma || !dbeta || !xchg || !sync) return fail(PP_ERR_BAD_ARG, "conv bwd_data + BatchNorm bwd: null");
    if (act < 0 || act > 2) return fail(PP_ERR_BAD_ARG, "conv bwd_data + BatchNorm bwd: act %d", act);
    if ((reinterpret_cast<uintptr_t>(xchg) & 7) != 0 || sync_ints < 2) return fail(PP_ERR_BAD_ARG, "conv bwd_data + BatchNorm bwd: exchange area");
    ConvParams p; ConvPlan pl;
    int R = 0;
    if (!conv_bn_bwd_shape(p, pl, B, H, W, Cin, Cout, kh, kw, stride, pad, dil, lddy, &R))
        return fail(PP_ERR_UNSUPPORTED, "conv bwd_data + BatchNorm bwd: this shape has no fused kernel (ask pp_conv2d_bwd_data_bn_bwd_ok first)");
    const size_t need = (size_t)cdiv(Cin, 32) * (size_t)R * 64 * 8;
    if (xchg_bytes < need) return fail(PP_ERR_WORKSPACE, "conv bwd_data + BatchNorm bwd: exchange area %zu < %zu", xchg_bytes, need);
    p.x = dy; p.w = w; p.bias = nullptr; p.y = dx_bn; p.ldy = lddx;
    p.bn = BnTrain{gamma, beta, 0.0f, 0.0f, nullptr, nullptr, const_cast<float*>(mean), const_cast<float*>(invstd), nullptr, 0, act, dx_bn, lddx,
                   reinterpret_cast<xword*>(xchg), sync, 0, bn_x, ldbx, dgamma, dbeta, grad_in, ldgi, dres, lddr};
    return launch_conv<true>(p, nullptr, 0, as_stream(stream), kh * kw);
}

int pp_conv2d_bwd_data_pre(const float* dy, int64_t lddy, int B, int Ho, int Wo, int Cout, const float* w, int kh, int kw,
                           int stride, int pad, int dil, float* dx, int64_t lddx, int H, int W, int Cin, int accumulate,
                           void* workspace, size_t ws_bytes, const void* dy_planes, pp_stream_t stream)
{
    return conv2d_bwd_data_impl(dy, lddy, B, Ho, Wo, Cout, w, kh, kw, stride, pad, dil, dx, lddx, H, W, Cin, accumulate, workspace, ws_bytes,
                                stream, dy_planes);
}

int pp_conv2d_bwd_data_pre2(const float* dy, int64_t lddy, int B, int Ho, int Wo, int Cout, const float* w, int kh, int kw,
                            int stride, int pad, int dil, float* dx, int64_t lddx, int H, int W, int Cin, int accumulate,
                            void* workspace, size_t ws_bytes, const void* dy_planes, const void* w_planes, pp_stream_t stream)
{
    return conv2d_bwd_data_impl(dy, lddy, B, Ho, Wo, Cout, w, kh, kw, stride, pad, dil, dx, lddx, H, W, Cin, accumulate, workspace, ws_bytes,
                                stream, dy_planes, w_planes);
}

// bf16x3 weight gradient (conv_wgrad_x3_kernel): the MFMA-bound layers (>= 8 GFLOP, both channel counts > 64, 128-wide outputs)
struct X3WPlan { bool ok; int Cin_p, Cout_p; int64_t x_plane, dy_plane; size_t bytes; };
static X3WPlan x3w_plan(int B, int H, int W, int Cin, int Cout, int64_t M, int ntaps, bool shape_ok)
{
    X3WPlan x{};
    if (!g_conv_x3 || (g_conv_x3 & 8) || !shape_ok || ntaps > 32 || 2.0 * ntaps * Cin * Cout * (double)M < g_x3w_flop) return x;
    x.Cin_p = (int)cdiv(Cin, 16) * 16;
    x.Cout_p = (int)cdiv(Cout, 16) * 16;
    const int64_t rows_x = (int64_t)B * H * W;
    x.x_plane = (rows_x + 1) * x.Cin_p;
    x.dy_plane = (M + 1) * x.Cout_p;
    if (x.x_plane * 2 >= (1ll << 32) - 4096 || x.dy_plane * 2 >= (1ll << 32) - 4096) return x;
    x.bytes = align_up((size_t)3 * x.x_plane * 2, 256) + align_up((size_t)3 * x.dy_plane * 2, 256);
    x.ok = true;
    return x;
}

size_t pp_conv2d_bwd_weight_workspace_bytes(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad,
                                            int dil)
{
    // (found by the AddressSanitizer driver, tests/test_abi_asan.py: a zero stride divided by zero here - every other query checks first)
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || kh < 1 || kw < 1 || kh > kMaxTaps || kw > kMaxTaps || kh * kw > kMaxTaps || !conv_geom_sane(H, W, stride, pad, dil) || stride < 1 || dil < 1 || pad < 0) return 0;
    const int Ho = out_size(H, kh, stride, pad, dil), Wo = out_size(W, kw, stride, pad, dil);
    const int64_t M = (int64_t)B * Ho * Wo;
    // splits chosen in pp_conv2d_bwd_weight never exceed 64
    (void)M;
    size_t w = (size_t)64 * kh * kw * Cin * Cout * 4, b = (size_t)256 * Cout * 4;
    const bool narrow = Cin == 3 || ((kh == 1 && kw == 1) && (Cin <= 32 || Cout <= 32) && Cin <= 256 && Cout <= 256);
    if (narrow) {               // wgrad_narrow_*: up to narrow_splits_max (+ one block of row lanes) splits
        const size_t sp = (size_t)narrow_splits_max((int64_t)kh * kw * Cin * Cout) + 16;
        const size_t n = sp * kh * kw * Cin * Cout * 4 + sp * Cout * 4;
        if (n > w) w = n;
    }
    w += (size_t)64 * Cout * 4;            // bias-gradient partials ride behind the weight partials
    size_t total = align_up(w > b ? w : b, 256);
    {
        const bool big = Cin > 64 && Cout > 64, vec = Cin % 4 == 0 && Cout % 4 == 0;
        const int remn = Cout % 128;
        const bool narrow_n = big && remn != 0 && remn <= 64 && (cdiv(Cout, 128) * 128 - Cout) * 100 > 12 * Cout;
        const X3WPlan xw = x3w_plan(B, H, W, Cin, Cout, M, kh * kw, vec && big && !narrow_n);
        if (xw.ok) total += 256 + xw.bytes;
    }
    return total;
}

static int conv2d_bwd_weight_impl(const float* x, int64_t ldx, int B, int H, int W, int Cin, const float* dy, int64_t lddy,
                                  int Cout, int kh, int kw, int stride, int pad, int dil, float* dw, float* dbias,
                                  void* workspace, size_t ws_bytes, pp_stream_t stream, pp_reduce_job* job,
                                  const void* x_planes = nullptr, const void* dy_planes = nullptr)
{
    if (job) job->kind = 0;
    if (int rc = conv_common_check(x, dy, dw, B, H, W, Cin, Cout, kh, kw, stride, pad, dil)) return rc;
    const int Ho = out_size(H, kh, stride, pad, dil), Wo = out_size(W, kw, stride, pad, dil);
    hipStream_t st = as_stream(stream);
    WgradParams p{};
    p.x = x; p.dy = dy; p.ldx = ldx; p.lddy = lddy;
    p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.Cin = Cin; p.Cout = Cout; p.stride = stride;
    p.M = (int64_t)B * Ho * Wo;
    build_taps(p.taps, kh, kw, stride, pad, dil, H, W, Ho, Wo, false);
    if (p.M > 0x7FFFFFFFll) return fail(PP_ERR_UNSUPPORTED, "conv bwd_weight: more than 2^31 output pixels");
    p.pointwise = (kh == 1 && kw == 1 && stride == 1 && pad == 0) ? 1 : 0;
    p.xcd_remap = g_wgrad_xcd;
    if (g_wgrad_narrow) {
        bool bias_done = false;
        const int rc = launch_wgrad_narrow(p, kh, kw, dw, dbias, &bias_done, workspace, ws_bytes, st, job);
        if (rc != 1) {                       // 0: handled, < 0: error, 1: not a narrow layer
            if (rc < 0) return rc;
            if (bias_done) return PP_OK;
            goto bias_part;
        }
    }
    {
    const bool big = Cin > 64 && Cout > 64;
    // 128-row tiles waste (128 - Cin % 128) rows of the last tile: 304 input channels (SegmentHead, decoders.py:107) fill
    // 2.4 of 3 tiles; 64-row tiles waste 5 % instead of 21 %
    const int rem = Cin % 128;
    const int remn = Cout % 128;
    const bool narrow_n = big && g_wgrad_m64 && remn != 0 && remn <= 64 && (cdiv(Cout, 128) * 128 - Cout) * 100 > 12 * Cout;
    const bool vec = g_conv_novec == 0 && Cin % 4 == 0 && Cout % 4 == 0 && ldx % 4 == 0 && lddy % 4 == 0 &&
                     (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0;
    // MFMA-bound layers run the bf16x3 kernel (operand planes behind the partial sums in the workspace); a bias gradient then takes
    // the separate column-sum pass at the end.  Its 128-row tiles skip the waves whose rows lie past Cin, so it keeps them for a
    // ragged Cin (measured on 304 -> 256: 300 us against 312 us with 64-row tiles, profiles/r03_conv_x3.txt)
    const X3WPlan xw = x3w_plan(B, H, W, Cin, Cout, p.M, p.taps.n, vec && big && !narrow_n);
    const bool narrow_m = big && !xw.ok && g_wgrad_m64 && rem != 0 && rem <= 64 && (cdiv(Cin, 128) * 128 - Cin) * 100 > 12 * Cin;
    const int bm = big ? (narrow_m ? 64 : 128) : 64, bn = big ? (narrow_n ? 64 : 128) : 64;
    const int64_t tiles = cdiv(Cin, bm) * cdiv(Cout, bn) * p.taps.n;
    int64_t splits = cdiv(g_wgrad_target, tiles);
    const int64_t max_by_m = cdiv(p.M, g_wgrad_rows_min);       // at least this many pixels per split
    if (splits > max_by_m) splits = max_by_m;
    if (splits > 64) splits = 64;
    if (splits < 1) splits = 1;
    if (g_wgrad_balance && big && 2.0 * p.taps.n * Cin * Cout * (double)p.M >= 8e9)
        splits = wgrad_balanced_splits(tiles, p.M, bm, bn, (int64_t)p.taps.n * Cin * Cout, max_by_m < 64 ? max_by_m : 64);
    p.m_per_split = cdiv(cdiv(p.M, splits), BK) * BK;
    splits = cdiv(p.M, p.m_per_split);
    const size_t need = (size_t)splits * p.taps.n * Cin * Cout * 4;
    if (!workspace || ws_bytes < need) return fail(PP_ERR_WORKSPACE, "conv bwd_weight: workspace %zu < %zu", ws_bytes, need);
    p.part = reinterpret_cast<float*>(workspace);
    const size_t x3_off = align_up(need + (size_t)64 * Cout * 4, 256);
    const bool use_x3 = xw.ok && ws_bytes >= x3_off + xw.bytes && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0;
    // bias gradient = column sums of dy: the blocks of tile column 0 read every dy tile anyway (no second pass over dy)
    const bool fuse_bias = !use_x3 && dbias != nullptr && ws_bytes >= need + (size_t)splits * Cout * 4;
    p.bias_part = fuse_bias ? p.part + (size_t)splits * p.taps.n * Cin * Cout : nullptr;
    if (p.taps.n != kh * kw)
        if (hipMemsetAsync(dw, 0, (size_t)kh * kw * Cin * Cout * 4, st) != hipSuccess)
            return fail(PP_ERR_LAUNCH, "conv bwd_weight: memset failed");
    dim3 grid((unsigned)(cdiv(Cin, bm) * p.taps.n), (unsigned)cdiv(Cout, bn), (unsigned)splits);
    // the split-K reduction inside the launch: the last block of a tile to arrive sums the tile's slices in slice order (bit-identical
    // to the reduce launch it replaces: wgrad_reduce4_kernel, 38 launches of a DeepLab step).  Not with a fused bias gradient (its
    // partials have their own final launch) and not when the caller defers the reduce into a batch
    p.counters = nullptr; p.dw = dw; p.splits = (int)splits;
    if (g_wgrad_fold && splits > 1 && !fuse_bias && !job && (int64_t)Cin * Cout % 4 == 0 && Cout % 4 == 0 &&
        (reinterpret_cast<uintptr_t>(p.part) & 15) == 0 && (reinterpret_cast<uintptr_t>(dw) & 15) == 0) {
        const int64_t tiles_ctr = (int64_t)grid.x * grid.y;
        if (tiles_ctr <= kWgradCounters / 4) p.counters = wgrad_counters_take((int)tiles_ctr);
    }
    // LDS-DMA kernels: vector operands, no fused bias gradient, 32-bit safe row pitch; bits of g_wgrad_dma: 1 = 128-wide tiles, 2 = 64x64
    const bool dma = vec && p.bias_part == nullptr && (int64_t)p.M * std::max(ldx, lddy) < (1ll << 40);
    {
        if (use_x3) {
            uint16_t* xw_ = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(workspace) + x3_off);
            uint16_t* dw_ = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(xw_) + align_up((size_t)3 * xw.x_plane * 2, 256));
            const uint16_t* xp = x_planes ? reinterpret_cast<const uint16_t*>(x_planes) : xw_;
            const uint16_t* dp = dy_planes ? reinterpret_cast<const uint16_t*>(dy_planes) : dw_;
            const int64_t rows_x = (int64_t)B * H * W;
            if (!x_planes)
                hipLaunchKernelGGL(x3_split_kernel, dim3((unsigned)std::min<int64_t>(cdiv((rows_x + 1) * (xw.Cin_p / 8), 256), 4096)), dim3(256), 0, st,
                                   x, ldx, rows_x, Cin, xw_, xw.Cin_p, xw.x_plane);
            if (!dy_planes)
                hipLaunchKernelGGL(x3_split_kernel, dim3((unsigned)std::min<int64_t>(cdiv((p.M + 1) * (xw.Cout_p / 8), 256), 4096)), dim3(256), 0, st,
                                   dy, lddy, p.M, Cout, dw_, xw.Cout_p, xw.dy_plane);
            if (int rc = check_launch("x3_split_kernel")) return rc;
            X3WOperands o{xp, dp, xw.x_plane, xw.dy_plane, (uint32_t)((rows_x + 1) * 32), (uint32_t)((p.M + 1) * 32),
                          (uint32_t)(rows_x * 32), (uint32_t)(p.M * 32)};
            dim3 gx((unsigned)(cdiv(Cin, 128) * p.taps.n), (unsigned)cdiv(Cout, 128), (unsigned)splits);
            hipLaunchKernelGGL((conv_wgrad_x3_kernel<128>), gx, dim3(kThreads), 0, st, p, o);
            goto reduce_partials;
        }
    }
    if (dma && (g_wgrad_dma & 1) && big && !(narrow_m && narrow_n)) {
        if (narrow_m)      hipLaunchKernelGGL((conv_wgrad_dma_kernel<64, 128>), grid, dim3(kThreads), 0, st, p);
        else if (narrow_n) hipLaunchKernelGGL((conv_wgrad_dma_kernel<128, 64>), grid, dim3(kThreads), 0, st, p);
        else               hipLaunchKernelGGL((conv_wgrad_dma_kernel<128, 128>), grid, dim3(kThreads), 0, st, p);
    } else if (dma && (g_wgrad_dma & 2) && (!big || (narrow_m && narrow_n))) {
        hipLaunchKernelGGL((conv_wgrad_dma_kernel<64, 64>), grid, dim3(kThreads), 0, st, p);
    } else
    if (big && narrow_m && narrow_n) {
        if (vec) hipLaunchKernelGGL((conv_wgrad_kernel<64, 64, 2, 2, true>), grid, dim3(kThreads), g_wgrad_lds_pad, st, p);
        else     hipLaunchKernelGGL((conv_wgrad_kernel<64, 64, 2, 2, false>), grid, dim3(kThreads), g_wgrad_lds_pad, st, p);
    } else if (big && narrow_m) {
        if (vec) hipLaunchKernelGGL((conv_wgrad_kernel<64, 128, 2, 2, true>), grid, dim3(kThreads), g_wgrad_lds_pad, st, p);
        else     hipLaunchKernelGGL((conv_wgrad_kernel<64, 128, 2, 2, false>), grid, dim3(kThreads), g_wgrad_lds_pad, st, p);
    } else if (big && narrow_n) {
        if (vec) hipLaunchKernelGGL((conv_wgrad_kernel<128, 64, 2, 2, true>), grid, dim3(kThreads), g_wgrad_lds_pad, st, p);
        else     hipLaunchKernelGGL((conv_wgrad_kernel<128, 64, 2, 2, false>), grid, dim3(kThreads), g_wgrad_lds_pad, st, p);
    } else if (big) {
        if (vec) hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, 2, 2, true>), grid, dim3(kThreads), g_wgrad_lds_pad, st, p);
        else     hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, 2, 2, false>), grid, dim3(kThreads), g_wgrad_lds_pad, st, p);
    } else {
        if (vec) hipLaunchKernelGGL((conv_wgrad_kernel<64, 64, 2, 2, true>), grid, dim3(kThreads), g_wgrad_lds_pad, st, p);
        else     hipLaunchKernelGGL((conv_wgrad_kernel<64, 64, 2, 2, false>), grid, dim3(kThreads), g_wgrad_lds_pad, st, p);
    }
reduce_partials:
    if (int rc = check_launch("conv_wgrad_kernel")) return rc;
    const int64_t cn = (int64_t)Cin * Cout;
    if (p.counters) {                         // the launch reduced its own slices (wgrad_fold_tail)
        if (job) job->kind = 0;
        goto bias_part;
    }
    if (cn % 4 == 0 && (reinterpret_cast<uintptr_t>(p.part) & 15) == 0 && (reinterpret_cast<uintptr_t>(dw) & 15) == 0 &&
        reduce_deferrable(job, p.taps, cn, splits, dbias)) {
        fill_reduce_job(job, 1, p.part, dw, cn, splits, p.taps);
        return PP_OK;
    }
    if (cn % 4 == 0 && (reinterpret_cast<uintptr_t>(p.part) & 15) == 0 && (reinterpret_cast<uintptr_t>(dw) & 15) == 0)
        hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3((unsigned)cdiv(p.taps.n * (cn / 4), 256)), dim3(256), 0, st, p.part,
                           (int)splits, p.taps.n, cn, p.taps, dw);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv(p.taps.n * cn, 256)), dim3(256), 0, st, p.part,
                           (int)splits, p.taps.n, cn, p.taps, dw);
    if (int rc = check_launch("wgrad_reduce_kernel")) return rc;
    if (fuse_bias) {
        hipLaunchKernelGGL(bias_grad_final_kernel, dim3((unsigned)cdiv(Cout, 8)), dim3(256), 0, st, p.bias_part, (int)splits, Cout, dbias);
        if (int rc = check_launch("bias_grad_final_kernel")) return rc;
        return PP_OK;
    }
    }
bias_part:
    if (dbias) {
        // partials reuse the (already consumed) head of the workspace: stream order makes that safe
        int nblk = (int)cdiv(p.M, 512);
        if (nblk > 256) nblk = 256;
        const int64_t rpb = cdiv(p.M, nblk);
        nblk = (int)cdiv(p.M, rpb);
        if ((size_t)nblk * Cout * 4 > ws_bytes) return fail(PP_ERR_WORKSPACE, "conv bwd_weight: workspace (bias)");
        float* bpart = reinterpret_cast<float*>(workspace);
        hipLaunchKernelGGL(bias_grad_partial_kernel, dim3((unsigned)cdiv(Cout, 64), (unsigned)nblk), dim3(256), 0, st, dy, p.M,
                           Cout, lddy, rpb, bpart);
        if (int rc = check_launch("bias_grad_partial_kernel")) return rc;
        hipLaunchKernelGGL(bias_grad_final_kernel, dim3((unsigned)cdiv(Cout, 8)), dim3(256), 0, st, bpart, nblk, Cout, dbias);
        if (int rc = check_launch("bias_grad_final_kernel")) return rc;
    }
    return PP_OK;
}

int pp_conv2d_bwd_weight(const float* x, int64_t ldx, int B, int H, int W, int Cin, const float* dy, int64_t lddy,
                         int Cout, int kh, int kw, int stride, int pad, int dil, float* dw, float* dbias,
                         void* workspace, size_t ws_bytes, pp_stream_t stream)
{
    return conv2d_bwd_weight_impl(x, ldx, B, H, W, Cin, dy, lddy, Cout, kh, kw, stride, pad, dil, dw, dbias, workspace, ws_bytes, stream, nullptr);
}

int pp_conv2d_bwd_weight_pre(const float* x, int64_t ldx, int B, int H, int W, int Cin, const float* dy, int64_t lddy,
                             int Cout, int kh, int kw, int stride, int pad, int dil, float* dw, float* dbias,
                             void* workspace, size_t ws_bytes, const void* x_planes, const void* dy_planes, pp_stream_t stream)
{
    return conv2d_bwd_weight_impl(x, ldx, B, H, W, Cin, dy, lddy, Cout, kh, kw, stride, pad, dil, dw, dbias, workspace, ws_bytes, stream, nullptr,
                                  x_planes, dy_planes);
}

// 0: the call would not run a bf16x3 kernel on such planes (they would be ignored); else the bytes of the planes it reads
// (which: 0 forward -> A planes of x, 1 backward-data -> A planes of dy, 2 weight gradient -> planes of x AND of dy are used;
//  3 / 4: forward / backward-data -> the WEIGHT planes in that direction's layout - also asked for by the layers that split their
//  activations inside the kernel (conv_x3f.hip) and therefore want no A planes: 0 / 1 answer 0 for those)
size_t pp_conv2d_x3_planes_bytes(int which, int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil)
{
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || kh < 1 || kw < 1 || kh > kMaxTaps || kw > kMaxTaps || kh * kw > kMaxTaps || !conv_geom_sane(H, W, stride, pad, dil) || stride < 1 || dil < 1) return 0;
    const int Ho = out_size(H, kh, stride, pad, dil), Wo = out_size(W, kw, stride, pad, dil);
    if (Ho < 1 || Wo < 1) return 0;
    const bool vec = Cin % 4 == 0 && Cout % 4 == 0;
    ConvTaps t;
    if (which == 0 || which == 3) {
        build_taps(t, kh, kw, stride, pad, dil, H, W, Ho, Wo, false);
        const int64_t M = (int64_t)B * Ho * Wo;
        if (ksplit_shape_ok(M, Cout, Cin, t.n, stride)) return 0;
        const ConvPlan pl = plan_conv(M, Cout, Cin, t.n, vec);
        const X3Plan xf = x3_plan(pl, M, (int64_t)B * H * W, Cin, Cout, kh * kw, t.n, vec);
        if (which == 3) return xf.ok ? pp_x3_weight_planes_bytes(kh * kw, Cin, Cout, 1) : 0;
        return xf.classic ? pp_x3_planes_bytes((int64_t)B * H * W, Cin) : 0;
    }
    if (which == 1 || which == 4) {
        if (stride != 1) return 0;
        build_taps(t, kh, kw, 1, pad, dil, Ho, Wo, H, W, true, stride);
        const int64_t M = (int64_t)B * H * W;
        if (ksplit_shape_ok(M, Cin, Cout, t.n, stride)) return 0;
        const ConvPlan pl = plan_conv(M, Cin, Cout, t.n, vec);
        const X3Plan xb = x3_plan(pl, M, (int64_t)B * Ho * Wo, Cout, Cin, kh * kw, t.n, vec);
        if (which == 4) return xb.ok ? pp_x3_weight_planes_bytes(kh * kw, Cin, Cout, 0) : 0;
        return xb.classic ? pp_x3_planes_bytes((int64_t)B * Ho * Wo, Cout) : 0;
    }
    if (which == 2) {
        build_taps(t, kh, kw, stride, pad, dil, H, W, Ho, Wo, false);
        const int64_t M = (int64_t)B * Ho * Wo;
        const bool big = Cin > 64 && Cout > 64;
        const int remn = Cout % 128;
        const bool narrow_n = big && remn != 0 && remn <= 64 && (cdiv(Cout, 128) * 128 - Cout) * 100 > 12 * Cout;
        return x3w_plan(B, H, W, Cin, Cout, M, t.n, vec && big && !narrow_n).ok ? pp_x3_planes_bytes((int64_t)B * H * W, Cin) : 0;
    }
    return 0;
}

int pp_conv2d_bwd_weight_partials(const float* x, int64_t ldx, int B, int H, int W, int Cin, const float* dy, int64_t lddy,
                                  int Cout, int kh, int kw, int stride, int pad, int dil, float* dw, float* dbias,
                                  void* workspace, size_t ws_bytes, pp_reduce_job* job, pp_stream_t stream)
{
    if (!job) return fail(PP_ERR_BAD_ARG, "conv bwd_weight_partials: job is NULL");
    return conv2d_bwd_weight_impl(x, ldx, B, H, W, Cin, dy, lddy, Cout, kh, kw, stride, pad, dil, dw, dbias, workspace, ws_bytes, stream, job);
}

int pp_wgrad_reduce_batch(const pp_reduce_job* jobs, int n, pp_stream_t stream)
{
    if (n < 0 || (n > 0 && !jobs)) return fail(PP_ERR_BAD_ARG, "wgrad_reduce_batch: jobs");
    hipStream_t st = as_stream(stream);
    int i = 0;
    while (i < n) {
        ReduceBatch b{};
        int m = 0;
        int64_t blocks = 0;
        for (; i < n && m < kBatchJobs; ++i) {
            const pp_reduce_job& j = jobs[i];
            if (j.kind == 0) continue;                       // nothing was deferred for this layer
            if (j.kind < 1 || j.kind > 3 || !j.part || !j.dst || j.cn < 1 || j.splits < 1 || (j.kind != 3 && (j.ntaps < 1 || j.ntaps > 12)))
                return fail(PP_ERR_BAD_ARG, "wgrad_reduce_batch: job %d is malformed", i);
            const int64_t total = j.kind == 3 ? j.cn : (int64_t)j.ntaps * j.cn;
            if (total >= (1ll << 31) || (j.kind == 1 && (j.cn % 4 != 0 || ((reinterpret_cast<uintptr_t>(j.part) | reinterpret_cast<uintptr_t>(j.dst)) & 15))))
                return fail(PP_ERR_BAD_ARG, "wgrad_reduce_batch: job %d: size / alignment", i);
            const int64_t nb = j.kind == 1 ? cdiv(total / 4, 256) : cdiv(total, 8);
            if (blocks + nb >= (1ll << 31)) break;
            b.part[m] = j.part; b.dst[m] = j.dst; b.cn[m] = (int)j.cn; b.splits[m] = j.splits;
            b.ntaps[m] = (uint8_t)(j.kind == 3 ? 1 : j.ntaps); b.kind[m] = (uint8_t)j.kind;
            for (int k = 0; k < 12; ++k) b.widx[m][k] = j.widx[k];
            b.start[m] = (int)blocks;
            blocks += nb;
            ++m;
        }
        if (m == 0) continue;
        b.start[m] = (int)blocks;
        b.n = m;
        hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, st, b);
        if (int rc = check_launch("wgrad_reduce_batch_kernel")) return rc;
    }
    return PP_OK;
}

}  // extern "C"
