// pp_common.h — shared host/device helpers for libpixelpick_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "pixelpick_hip.h"

namespace pp {

constexpr int kWave = 64;  // CDNA wavefront

// thread-local error text behind pp_last_error()
char* err_buf();
int fail(int code, const char* fmt, ...);

inline hipStream_t as_stream(pp_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(PP_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return PP_OK;
}

// Optional profiling hook (bench.py): caller-owned hipEvent pairs recorded right around a dominant kernel's
// launch (acq_kernel, conv_igemm_kernel), pair i for the i-th such launch after pp_set_kernel_events().
struct EventHook {
    hipEvent_t* start;
    hipEvent_t* stop;
    int n, i;
};
EventHook& event_hook();

struct EventScope {
    hipStream_t st;
    bool on;
    explicit EventScope(hipStream_t s) : st(s), on(event_hook().i < event_hook().n)
    {
        if (on) (void)hipEventRecord(event_hook().start[event_hook().i], st);
    }
    ~EventScope()
    {
        if (on) (void)hipEventRecord(event_hook().stop[event_hook().i++], st);
    }
};

// CUs the caller sets aside for kernels that stay resident beside ours for a long time (RCCL's channel blocks during an overlapped
// all-reduce): pp_set_comm_cu_reserve / PIXELPICK_COMM_CU_RESERVE.  Every launch whose blocks wait for each other (single-launch
// BatchNorm, convolution + BatchNorm epilogues) sizes itself against occupancy x (CUs - reserve) instead of occupancy x CUs.
int comm_cu_reserve();
inline int reserve_scaled(int cap, int cus)
{
    const int r = comm_cu_reserve();
    if (r <= 0 || cus <= 0) return cap;
    if (r >= cus) return 0;
    return (int)((int64_t)cap * (cus - r) / cus);
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Inference epilogue shared by the dense and depthwise convolutions: eval-mode BatchNorm folded per output channel
// (scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale: the arithmetic of pp_bn_eval_affine +
// pp_scale_shift_act, so the fused and the three-launch forms agree bit for bit), optional residual, activation.
struct Epilogue {
    const float* gamma;        // NULL: no BatchNorm (plain bias / identity)
    const float* beta;
    const float* mean;
    const float* var;
    float eps;
    const float* res;          // NULL: no residual
    int64_t ldr;
    int act;                   // 0 none, 1 ReLU, 2 ReLU6
};

// ---- bilinear source coordinates (torch upsample_bilinear2d index arithmetic, fp32) ---------------------
//   align_corners:  src = scale*dst,                      scale = (in-1)/(out-1)   (0 if out == 1)
//   otherwise:      src = max(scale*(dst+0.5)-0.5, 0),    scale = in/out (or 1/scale_factor)
struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp lerp_src(int dst, int in, float scale, int align)
{
    float src = align ? __fmul_rn(scale, (float)dst) : fmaxf(__fsub_rn(__fmul_rn(scale, (float)dst + 0.5f), 0.5f), 0.0f);
    Lerp L;
    L.i0 = (int)src;
    if (L.i0 > in - 1) L.i0 = in - 1;
    L.i1 = L.i0 + (L.i0 < in - 1 ? 1 : 0);
    L.l1 = src - (float)L.i0;
    L.l0 = 1.0f - L.l1;
    return L;
}

// h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11) with the contraction spelled out, so that every kernel that
// interpolates (pp_bilinear_fwd and the fused low-resolution acquisition) produces the same bits.
__device__ __forceinline__ float bilerp(float h0, float h1, float w0, float w1, float v00, float v01, float v10, float v11)
{
    const float top = __fmaf_rn(w1, v01, __fmul_rn(w0, v00));
    const float bot = __fmaf_rn(w1, v11, __fmul_rn(w0, v10));
    return __fmaf_rn(h1, bot, __fmul_rn(h0, top));
}

#ifdef __HIPCC__
__device__ __forceinline__ float epi_act(float z, int act)
{
    if (act == 1) return fmaxf(z, 0.0f);
    if (act == 2) return fminf(fmaxf(z, 0.0f), 6.0f);
    return z;
}
// ---- wave64 reductions -------------------------------------------------------------------------
// DPP row_shr 1/2/4/8 builds the row maximum in lane 15 of each 16-lane row, row_bcast15/31 carry it
// to lane 63 (canonical GFX9 reduction); result broadcast with readlane.  No LDS traffic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_umax_step(uint32_t v)
{
    uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
    return v > t ? v : t;
}

__device__ __forceinline__ uint32_t wave_umax_dpp(uint32_t v)
{
    v = dpp_umax_step<0x111, 0xf>(v);  // row_shr:1
    v = dpp_umax_step<0x112, 0xf>(v);  // row_shr:2
    v = dpp_umax_step<0x114, 0xf>(v);  // row_shr:4
    v = dpp_umax_step<0x118, 0xf>(v);  // row_shr:8
    v = dpp_umax_step<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
    v = dpp_umax_step<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ uint32_t wave_umax_shfl(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t t = (uint32_t)__shfl_xor((int)v, o, 64);
        v = v > t ? v : t;
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_umax(uint32_t v, int mode)
{
    return mode == 0 ? wave_umax_dpp(v) : wave_umax_shfl(v);
}

// ---- order-preserving float <-> u32 keys --------------------------------------------------------
// Larger key == selected earlier.  Policy: NaN first for largest, last for smallest;
// -0.0 == +0.0; key 0 is reserved for "no element".
__device__ __forceinline__ uint32_t order_key(float v, bool largest)
{
    uint32_t u;
    if (v != v) {
        u = 0xFFFFFFFFu;
    } else {
        v = v + 0.0f;
        u = __float_as_uint(v);
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    }
    u = largest ? u : ~u;
    return u == 0u ? 1u : u;  // only NaN under !largest maps to 0 -> 1 (still below every real key)
}

__device__ __forceinline__ float key_to_float(uint32_t key, bool largest)
{
    uint32_t u = largest ? key : ~key;
    if (!largest && key == 1u) return __uint_as_float(0x7FC00000u);  // NaN under !largest
    if (u == 0xFFFFFFFFu) return __uint_as_float(0x7FC00000u);
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    return __uint_as_float(u);
}

// Fixed-order fp64 sum of `nblk` partial values per output, 32 lanes per output (8 outputs per 256-thread
// block): lane l adds blocks l, l+32, ... then an LDS tree combines the lanes.  Deterministic.
__device__ __forceinline__ double lanes32_sum(const float* part, int nblk, int64_t stride_b, int64_t idx, bool valid,
                                              double* sh /*[256]*/)
{
    const int t = threadIdx.x, lane = t >> 3;
    double s = 0.0;
    if (valid) {
        int b = lane;
        for (; b + 96 < nblk; b += 128) {      // 4 independent loads in flight per lane
            const float v0 = part[(int64_t)b * stride_b + idx], v1 = part[(int64_t)(b + 32) * stride_b + idx];
            const float v2 = part[(int64_t)(b + 64) * stride_b + idx], v3 = part[(int64_t)(b + 96) * stride_b + idx];
            s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
        }
        for (; b < nblk; b += 32) s += (double)part[(int64_t)b * stride_b + idx];
    }
    sh[t] = s;
    __syncthreads();
    for (int off = 128; off >= 8; off >>= 1) {
        if (t < off) sh[t] += sh[t + off];
        __syncthreads();
    }
    const double r = sh[t & 7];
    __syncthreads();
    return r;
}

#endif  // __HIPCC__

}  // namespace pp
