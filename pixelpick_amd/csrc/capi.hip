// capi.hip — version / error plumbing of the C ABI (include/pixelpick_hip.h).
#include "pp_common.h"

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

// Yardstick (measurement only): what a kernel that does nothing but READ a buffer reaches - float4 per lane, eight loads in flight,
// one wave per SIMD (the form tools/probe/hbm_rw.hip found fastest on this chip: 6.3-6.7 TB/s).  bench.py times it on the very logits
// buffer acq_kernel scores and reports acq_kernel against it next to the 8 TB/s specification figure.
__global__ __launch_bounds__(256) void stream_read_kernel(const float4* __restrict__ x, size_t n4, float* out)
{
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n4; i += 8 * stride) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; i < n4; i += stride) { const float4 v = x[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;       // (never true for the bench's data: keeps the loads alive)
}

}  // namespace pp

extern "C" {

int pp_debug_stream_read(const void* x, size_t bytes, int blocks, float* sink, pp_stream_t stream)
{
    if (!x || !sink || bytes < 16 || (reinterpret_cast<uintptr_t>(x) & 15)) return pp::fail(PP_ERR_BAD_ARG, "stream_read: buffer");
    if (blocks < 1) blocks = 256;
    hipLaunchKernelGGL(pp::stream_read_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4*>(x), bytes / 16, sink);
    return hipGetLastError() == hipSuccess ? PP_OK : pp::fail(PP_ERR_LAUNCH, "stream_read_kernel launch failed");
}

void pp_debug_set_kernel_events(void** starts, void** stops, int n)
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
