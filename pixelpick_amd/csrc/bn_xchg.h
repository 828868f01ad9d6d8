// bn_xchg.h - the cross-block exchange of the single-launch BatchNorm kernels (nn_ops.hip) and of the convolution kernels that
// finish a training BatchNorm in their own epilogue (conv_igemm.hip).  Device code only.
#pragma once
#include "pp_common.h"

namespace pp {

constexpr int kXT = 256;      // threads per block of every kernel that uses the exchange

// Partials cross XCDs.  Every partial is ONE 64-bit word {launch tag, value} written and read with agent-scope
// atomic accesses in FINE-GRAINED device memory: the value carries its own "ready" flag, so no ordering between a
// data store and a separate arrival counter is needed (that ordering needs a release fence = a `buffer_wbl2` per
// block, 0.09 us x blocks serialised; without the fence a counter can be seen before the data - observed as a
// run-to-run difference in ~1 of 10 twelve-step runs).  A reader spins until the tag of the word equals this
// launch's tag.  The tag is epoch+1, the epoch lives in sync[0] and is advanced by the last block of the launch to
// finish reading (counted in sync[1]), so consecutive launches - also replays of a captured graph - never share one.
typedef unsigned long long xword;

__device__ __forceinline__ void xchg_put(xword* p, float v, unsigned tag)
{
    __hip_atomic_store(p, ((xword)tag << 32) | (xword)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float xchg_get(const xword* p, unsigned tag)
{
    xword w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while ((unsigned)(w >> 32) != tag) {
        __builtin_amdgcn_s_sleep(2);
        w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return __uint_as_float((unsigned)w);
}
__device__ __forceinline__ void publish_partial(xword* p, int nch, const float4& s0, const float4& s1, unsigned tag)
{
    xchg_put(p + 0, s0.x, tag); xchg_put(p + 1, s0.y, tag); xchg_put(p + 2, s0.z, tag); xchg_put(p + 3, s0.w, tag);
    xchg_put(p + nch + 0, s1.x, tag); xchg_put(p + nch + 1, s1.y, tag); xchg_put(p + nch + 2, s1.z, tag); xchg_put(p + nch + 3, s1.w, tag);
}

// this launch's tag, read once per block (before any block of the launch can have advanced the epoch: the epoch moves
// only when every block has gone through launch_done).  The load is issued at kernel entry by thread 0 and only consumed
// after the statistics pass (tag_share in front of rowlane_tree, whose barriers publish it): its ~1 us of latency runs
// under the pass instead of in front of it.
__device__ __forceinline__ unsigned tag_issue(const int* sync)
{
    return threadIdx.x == 0 ? (unsigned)__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u : 0u;
}
__device__ __forceinline__ void tag_share(unsigned tag0, unsigned* sh_tag)
{
    if (threadIdx.x == 0) *sh_tag = tag0;
}

// called by every block after its last xchg_get: the last one through advances the epoch and re-arms the counter
__device__ __forceinline__ void launch_done(int* sync)
{
    if (threadIdx.x == 0) {
        const int gone = __hip_atomic_fetch_add(sync + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gone == (int)gridDim.x - 1) {
            __hip_atomic_store(sync + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// fixed-order fp64 sum over the strip's R partial rows; tot[o] for o < nout = 8*bq (stat-major: [2][bq*4])
__device__ __forceinline__ void strip_combine(const xword* part, int strip, int R, int nout, unsigned tag, double* shd /*[256]*/,
                                              double* tot /*[64]*/)
{
    const int t = threadIdx.x;
    const int nsub = kXT / nout;
    const int o = t % nout, sub = t / nout;
    double s = 0.0;
    if (sub < nsub) {
        const xword* p = part + (int64_t)strip * R * nout + o;
        int c = sub;
        // narrow strips of long maps (C <= 64: one or two strips x 128-256 row chunks): sixteen words in flight per thread, or the
        // combine is R / (4 nsub) dependent round trips to fine-grained memory (8.7 us of a 23 us launch on the 32-channel stem map)
        for (; c + 15 * nsub < R; c += 16 * nsub) {
            xword w[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) w[j] = __hip_atomic_load(p + (int64_t)(c + j * nsub) * nout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int j = 0; j < 16; j += 4) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    v[i] = (unsigned)(w[j + i] >> 32) == tag ? __uint_as_float((unsigned)w[j + i]) : xchg_get(p + (int64_t)(c + (j + i) * nsub) * nout, tag);
                s += ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]);      // the four-at-a-time loop's association
            }
        }
        for (; c + 3 * nsub < R; c += 4 * nsub) {
            // four words in flight; a word whose tag is not this launch's yet is re-read by xchg_get
            const xword* p0 = p + (int64_t)c * nout;
            const xword* p1 = p + (int64_t)(c + nsub) * nout;
            const xword* p2 = p + (int64_t)(c + 2 * nsub) * nout;
            const xword* p3 = p + (int64_t)(c + 3 * nsub) * nout;
            const xword w0 = __hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const xword w1 = __hip_atomic_load(p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const xword w2 = __hip_atomic_load(p2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const xword w3 = __hip_atomic_load(p3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float v0 = (unsigned)(w0 >> 32) == tag ? __uint_as_float((unsigned)w0) : xchg_get(p0, tag);
            const float v1 = (unsigned)(w1 >> 32) == tag ? __uint_as_float((unsigned)w1) : xchg_get(p1, tag);
            const float v2 = (unsigned)(w2 >> 32) == tag ? __uint_as_float((unsigned)w2) : xchg_get(p2, tag);
            const float v3 = (unsigned)(w3 >> 32) == tag ? __uint_as_float((unsigned)w3) : xchg_get(p3, tag);
            s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
        }
        for (; c < R; c += nsub) s += (double)xchg_get(p + (int64_t)c * nout, tag);
    }
    shd[t] = s;
    __syncthreads();
    if (t < nout) {
        double a = shd[t];
        for (int k = 1; k < nsub; ++k) a += shd[k * nout + t];
        tot[t] = a;
    }
    __syncthreads();
}


// derivative mask from the activation OUTPUT y (relu: y>0, relu6: 0<y<6)
__device__ __forceinline__ float act_mask(float y, int act)
{
    if (act == 1) return y > 0.0f ? 1.0f : 0.0f;
    if (act == 2) return (y > 0.0f && y < 6.0f) ? 1.0f : 0.0f;
    return 1.0f;
}


// dx of one element; no fma contraction, so that every kernel variant (and the mask / dropout scaling in front of it, which
// the row-cached variant applies in the reduction pass) rounds identically
__device__ __forceinline__ float bn_dx(float u, float v, float mu, float is, float ga, float db, float dg, float inv_count)
{
#pragma clang fp contract(off)
    return ga * is * (u - db * inv_count - (v - mu) * is * dg * inv_count);
}


}  // namespace pp
