// capi.hip — version / error plumbing of the C ABI (include/pixelpick_hip.h).
#include "pp_common.h"

#include <stdlib.h>

namespace pp {

char* err_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

EventHook& event_hook()
{
    static thread_local EventHook h{nullptr, nullptr, 0, 0};   // per calling thread, like every other debugging knob
    return h;
}

static int g_comm_cu_reserve = [] {
    const char* e = getenv("PIXELPICK_COMM_CU_RESERVE");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 0;
}();
int comm_cu_reserve() { return g_comm_cu_reserve; }

// Stand-in for a communication kernel that stays resident (tests/test_dist_gpu.py): every block takes a whole CU's LDS (no second
// block of anything that uses LDS fits beside it) and spins until *stop != 0 or `max_ticks` of the constant 100 MHz clock have
// passed - it cannot outlive its time limit whatever the host does.
__global__ __launch_bounds__(1024) void occupy_kernel(const int* stop, unsigned long long max_ticks, unsigned long long* started)
{
    extern __shared__ int occ_lds[];
    if (threadIdx.x == 0) {
        occ_lds[0] = (int)blockIdx.x;
        atomicAdd(started, 1ull);
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0 && wall_clock64() - t0 < max_ticks)
            __builtin_amdgcn_s_sleep(64);
    }
    __syncthreads();
}

// Yardstick (measurement only): what a kernel that does nothing but READ a buffer reaches - float4 per lane, eight loads in flight,
// one wave per SIMD (the form tools/probe/hbm_rw.hip found fastest on this chip: 6.3-6.7 TB/s).  bench.py times it on the very logits
// buffer acq_kernel scores and reports acq_kernel against it next to the 8 TB/s specification figure.
typedef float f32x4_ntl __attribute__((ext_vector_type(4)));
// NT: non-temporal loads (what the single-pass acquisition scorers use since round 5)
template <bool NT>
__global__ __launch_bounds__(256) void stream_read_kernel(const float4* __restrict__ x, size_t n4, float* out)
{
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    auto ld = [&](size_t j) -> float4 {
        if constexpr (NT) {
            const f32x4_ntl q = __builtin_nontemporal_load(reinterpret_cast<const f32x4_ntl*>(x) + j);
            return make_float4(q.x, q.y, q.z, q.w);
        } else {
            return x[j];
        }
    };
    for (; i + 7 * stride < n4; i += 8 * stride) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ld(i + u * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; i < n4; i += stride) { const float4 v = ld(i); acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;       // (never true for the bench's data: keeps the loads alive)
}

}  // namespace pp

extern "C" {

int pp_yardstick_stream_read(const void* x, size_t bytes, int blocks, float* sink, pp_stream_t stream)
{
    // blocks < 0: -blocks blocks with non-temporal loads
    const bool nt = blocks < 0;
    if (nt) blocks = -blocks;
    if (!x || !sink || bytes < 16 || (reinterpret_cast<uintptr_t>(x) & 15)) return pp::fail(PP_ERR_BAD_ARG, "stream_read: buffer");
    if (blocks < 1) blocks = 256;
    if (nt)
        hipLaunchKernelGGL(pp::stream_read_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           reinterpret_cast<const float4*>(x), bytes / 16, sink);
    else
        hipLaunchKernelGGL(pp::stream_read_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           reinterpret_cast<const float4*>(x), bytes / 16, sink);
    return hipGetLastError() == hipSuccess ? PP_OK : pp::fail(PP_ERR_LAUNCH, "stream_read_kernel launch failed");
}

void pp_set_comm_cu_reserve(int cus) { pp::g_comm_cu_reserve = cus > 0 ? cus : 0; }
int pp_get_comm_cu_reserve(void) { return pp::g_comm_cu_reserve; }

#ifdef PP_DEBUG_KNOBS
int pp_debug_occupy_cus(int blocks, const int* stop, uint64_t max_ticks, uint64_t* started, pp_stream_t stream)
{
    if (blocks < 1 || blocks > 256 || !stop || !started || max_ticks == 0 || max_ticks > 6000000000ull)
        return pp::fail(PP_ERR_BAD_ARG, "occupy_cus: blocks in [1, 256], a stop flag, a start counter and at most 60 s of ticks");
    static bool attr = false;
    const int lds = 160 * 1024;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(pp::occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return pp::fail(PP_ERR_LAUNCH, "occupy_cus: cannot ask for %d bytes of LDS", lds);
        attr = true;
    }
    hipLaunchKernelGGL(pp::occupy_kernel, dim3((unsigned)blocks), dim3(1024), lds, reinterpret_cast<hipStream_t>(stream), stop,
                       (unsigned long long)max_ticks, reinterpret_cast<unsigned long long*>(started));
    return hipGetLastError() == hipSuccess ? PP_OK : pp::fail(PP_ERR_LAUNCH, "occupy_kernel launch failed");
}
#endif

void pp_set_kernel_events(void** starts, void** stops, int n)
{
    pp::EventHook& h = pp::event_hook();
    h.start = reinterpret_cast<hipEvent_t*>(starts);
    h.stop = reinterpret_cast<hipEvent_t*>(stops);
    h.n = (starts && stops) ? n : 0;
    h.i = 0;
}

int pp_version(void) { return 100; }  // 0.1.0

const char* pp_last_error(void) { return pp::err_buf(); }

}
