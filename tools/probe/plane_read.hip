// probe: what a kernel that only READS NCHW logits the way acq_kernel has to (every pixel's C class values are C streams one
// plane apart) reaches, as a function of the plane size - against a linear read of the same bytes.  Answers the round-3 question
// "is 0.63-0.70 of 8 TB/s on the 8 MB planes of 1024 x 2048 (and on C = 21 / C = 11) the scorer or the layout?".
//   hipcc --offload-arch=gfx950 -O3 tools/probe/plane_read.hip -o /tmp/plane_read && /tmp/plane_read
// Variants: G groups of 4 pixels per lane with all C loads of a group in flight (acq_kernel's shape), block -> pixel-chunk map
// linear / XCD-contiguous (the blocks of one XCD walk one contiguous eighth of the launch) / image-interleaved, waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float f32x4_nt __attribute__((ext_vector_type(4)));
// NT (round 5): non-temporal loads - a stream that is read once should not allocate in L2 / MALL
template <bool NT> __device__ inline float4 ld4(const float4* p)
{
    if constexpr (NT) { const f32x4_nt q = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(p)); return make_float4(q.x, q.y, q.z, q.w); }
    else return *p;
}
template <int C, int G, int OCC, bool NT>
__global__ __launch_bounds__(256, OCC) void plane_read_k(const float4* __restrict__ x, int64_t plane4, int64_t chunks_per_image, int map, float* out)
{
    // plane4 = H*W/4 float4 per plane; a block owns chunk = 256*G consecutive float4 of ONE image (all C planes of them)
    int64_t b = blockIdx.x;
    const int64_t nb = gridDim.x;
    if (map == 1) {                       // XCD-contiguous: hardware deals block b to XCD b % 8
        const int64_t per = (nb + 7) / 8;
        b = (b % 8) * per + b / 8;
        if (b >= nb) return;
    }
    const int64_t img = b / chunks_per_image, ch = b % chunks_per_image;
    const float4* base = x + img * (int64_t)C * plane4 + ch * 256 * G + threadIdx.x;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float4 v[C];
        const int64_t off = (int64_t)g * 256;
        const bool ok = ch * 256 * G + off + threadIdx.x < plane4;
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = ok ? ld4<NT>(base + (int64_t)c * plane4 + off) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < C; ++c) { acc.x += v[c].x; acc.y += v[c].y; acc.z += v[c].z; acc.w += v[c].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

// the same bytes as one linear stream (eight float4 in flight per lane, 256 blocks x OCC): the read-only yardstick of bench.py
template <bool NT>
__global__ __launch_bounds__(256) void linear_read_k(const float4* __restrict__ x, size_t n4, float* out)
{
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n4; i += 8 * stride) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ld4<NT>(x + i + u * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; i < n4; i += stride) { const float4 v = ld4<NT>(x + i); acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <typename F>
static double time_ms(F&& launch, int reps = 12)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    std::vector<float> ms;
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms[ms.size() / 2];
}

template <int C, int G, int OCC>
static void run_case(const float4* x, float* out, int B, int H, int W, const char* tag)
{
    const int64_t plane4 = (int64_t)H * W / 4;
    const int64_t cpi = (plane4 + 256 * G - 1) / (256 * G);
    const double bytes = (double)B * C * plane4 * 16;
    for (int map = 0; map < 2; ++map) {
        const double ms = time_ms([&] { hipLaunchKernelGGL((plane_read_k<C, G, OCC, false>), dim3((unsigned)(B * cpi)), dim3(256), 0, 0, x, plane4, cpi, map, out); });
        const double mn = time_ms([&] { hipLaunchKernelGGL((plane_read_k<C, G, OCC, true>), dim3((unsigned)(B * cpi)), dim3(256), 0, 0, x, plane4, cpi, map, out); });
        printf("  %-28s C=%2d G=%d occ=%d map=%-6s  %8.3f ms  %7.1f GB/s  %.3f of 8 TB/s | non-temporal loads %8.3f ms  %.3f\n", tag, C, G, OCC, map ? "xcd" : "linear", ms,
               bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0, mn, bytes / mn / 1e6 / 8000.0);
    }
}

int main()
{
    const size_t cap = (size_t)2304 << 20;                 // 2.25 GB: larger than every case below (VOC: 2.2 GB), >> 256 MB MALL
    float4* x; float* out;
    CK(hipMalloc(&x, cap)); CK(hipMalloc(&out, 64));
    CK(hipMemset(x, 0, cap));
    struct Shape { int B, H, W; const char* tag; };
    const Shape shapes[] = {{128, 256, 512, "256x512 (512 KB planes)"}, {32, 512, 1024, "512x1024 (2 MB planes)"},
                            {8, 1024, 2048, "1024x2048 (8 MB planes)"}, {8, 1000, 2048, "1000x2048"}};
    for (const Shape& s : shapes) {
        const size_t n4 = (size_t)s.B * 19 * s.H * s.W / 4;
        const double ms = time_ms([&] { hipLaunchKernelGGL(linear_read_k<false>, dim3(256), dim3(256), 0, 0, x, n4, out); });
        const double msn = time_ms([&] { hipLaunchKernelGGL(linear_read_k<true>, dim3(256), dim3(256), 0, 0, x, n4, out); });
        printf("%s B=%d: linear read of the same bytes %8.3f ms %7.1f GB/s | non-temporal %8.3f ms %7.1f GB/s\n", s.tag, s.B, ms, n4 * 16.0 / ms / 1e6, msn, n4 * 16.0 / msn / 1e6);
        run_case<19, 2, 3>(x, out, s.B, s.H, s.W, s.tag);
        run_case<19, 2, 2>(x, out, s.B, s.H, s.W, s.tag);
        run_case<19, 1, 3>(x, out, s.B, s.H, s.W, s.tag);
        run_case<19, 1, 4>(x, out, s.B, s.H, s.W, s.tag);
        run_case<19, 2, 4>(x, out, s.B, s.H, s.W, s.tag);
        run_case<19, 2, 5>(x, out, s.B, s.H, s.W, s.tag);
        run_case<19, 4, 2>(x, out, s.B, s.H, s.W, s.tag);
    }
    {   // VOC 320x320 C=21 B=256 and CamVid 360x480 C=11 B=128
        const size_t n4 = (size_t)256 * 21 * 320 * 320 / 4;
        double ms = time_ms([&] { hipLaunchKernelGGL(linear_read_k<true>, dim3(256), dim3(256), 0, 0, x, n4, out); });
        printf("VOC 320x320 C=21 B=256: linear %8.3f ms %7.1f GB/s\n", ms, n4 * 16.0 / ms / 1e6);
        run_case<21, 2, 2>(x, out, 256, 320, 320, "voc 320x320");
        run_case<21, 1, 3>(x, out, 256, 320, 320, "voc 320x320");
        run_case<21, 1, 4>(x, out, 256, 320, 320, "voc 320x320");
        const size_t m4 = (size_t)128 * 11 * 360 * 480 / 4;
        ms = time_ms([&] { hipLaunchKernelGGL(linear_read_k<true>, dim3(256), dim3(256), 0, 0, x, m4, out); });
        printf("CamVid 360x480 C=11 B=128: linear %8.3f ms %7.1f GB/s\n", ms, m4 * 16.0 / ms / 1e6);
        run_case<11, 2, 3>(x, out, 128, 360, 480, "camvid 360x480");
        run_case<11, 2, 4>(x, out, 128, 360, 480, "camvid 360x480");
        run_case<11, 4, 3>(x, out, 128, 360, 480, "camvid 360x480");
    }
    hipFree(x); hipFree(out);
    return 0;
}
