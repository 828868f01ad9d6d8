// yardstick: what a kernel that only READS (sum), only WRITES (fill) or copies reaches on this chip, as a function of waves per
// SIMD and 16-byte accesses in flight per lane - the ceiling the bandwidth-bound kernels (acq_kernel: read-only; BatchNorm:
// a read phase, then a write phase) are graded against.  hipcc --offload-arch=gfx950 -O3 tools/probe/hbm_rw.hip -o /tmp/hbm_rw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

template <int U>
__global__ __launch_bounds__(256) void read_k(const float4* __restrict__ x, size_t n4, float* out)
{
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; i < n4; i += stride) { const float4 v = x[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
template <int U>
__global__ __launch_bounds__(256) void write_k(float4* __restrict__ y, size_t n4, float v0)
{
    const float4 v = make_float4(v0, v0, v0, v0);
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) y[i + u * stride] = v;
    }
    for (; i < n4; i += stride) y[i] = v;
}
template <int U>
__global__ __launch_bounds__(256) void copy_k(const float4* __restrict__ x, float4* __restrict__ y, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) y[i + u * stride] = v[u];
    }
    for (; i < n4; i += stride) y[i] = x[i];
}

// BatchNorm's access pattern on an [M][C] fp32 map: a block owns a strip of `bq` float4 columns (bq * 16 bytes of every row)
// and a chunk of rows; thread (ql = t % bq, rl = t / bq) walks rows rl, rl + 256/bq, ... of the chunk.  mode 0 read, 1 write.
__global__ __launch_bounds__(256) void strip_k(float4* __restrict__ x, int64_t M, int cq, int bq, int R, int mode, float* out)
{
    const int nstrips = cq / bq, nrl = 256 / bq;
    const int strip = blockIdx.x % nstrips, chunk = blockIdx.x / nstrips;
    const int ql = threadIdx.x % bq, rl = threadIdx.x / bq;
    const int64_t rpc = (M + R - 1) / R, r0 = chunk * rpc, r1 = r0 + rpc < M ? r0 + rpc : M;
    float4* xq = x + strip * bq + ql;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t r = r0 + rl; r < r1; r += (int64_t)nrl * 4) {
        if (mode == 0) {
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = xq[(r + j * nrl < r1 ? r + j * nrl : r) * cq];
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (r + j * nrl < r1) xq[(r + j * nrl) * cq] = make_float4(1.f, 2.f, 3.f, 4.f);
        }
    }
    if (mode == 0 && acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <typename F>
static double time_us(F&& f, int iters = 20)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    std::vector<float> ts;
    for (int i = 0; i < iters; ++i) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); ts.push_back(ms * 1e3f); }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main()
{
    {
        // the SegmentHead BatchNorm map: 32768 rows x 256 channels (33.6 MB), 256 blocks, strips of bq float4 columns
        const int64_t M = 32768; const int cq = 64;
        float4* x; float* out; float4* fl;
        hipMalloc(&x, M * cq * 16); hipMalloc(&out, 4); hipMalloc(&fl, (size_t)512 << 20);
        printf("== BatchNorm-shaped strips on a 32768 x 256 fp32 map, 256 blocks (us per launch incl. ~4.5 us dispatch; cold = 512 MiB written in between)\n");
        for (int bq : {8, 16, 32, 64}) {
            const int R = 256 / (cq / bq);
            double rd = time_us([&] { hipLaunchKernelGGL(strip_k, dim3(256), dim3(256), 0, 0, x, M, cq, bq, R, 0, out); });
            double wr = time_us([&] { hipLaunchKernelGGL(strip_k, dim3(256), dim3(256), 0, 0, x, M, cq, bq, R, 1, out); });
            // cold read: evict first (timed separately and subtracted is not possible with one event pair: time only the strip kernel)
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            std::vector<float> ts;
            for (int i = 0; i < 10; ++i) {
                hipMemsetAsync(fl, i, (size_t)512 << 20, 0);
                hipEventRecord(a); hipLaunchKernelGGL(strip_k, dim3(256), dim3(256), 0, 0, x, M, cq, bq, R, 0, out); hipEventRecord(b);
                hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); ts.push_back(ms * 1e3f);
            }
            std::sort(ts.begin(), ts.end());
            printf("  strip %4d B x %3d row chunks: read warm %5.1f us, read cold %5.1f us, write %5.1f us\n", bq * 16, R, rd, ts[5], wr);
        }
        hipFree(x); hipFree(out); hipFree(fl);
    }
    for (size_t mb : {32, 256, 2560}) {
        const size_t bytes = mb << 20, n4 = bytes / 16;
        float4 *x, *y; float* out;
        hipMalloc(&x, bytes); hipMalloc(&y, bytes); hipMalloc(&out, 4);
        hipMemset(x, 0, bytes); hipMemset(y, 0, bytes);
        printf("== %zu MiB per tensor (blocks of 256 threads; 256 CUs: 256 blocks = 1 wave / SIMD)\n", mb);
        for (int blocks : {256, 512, 1024, 2048, 4096}) {
            double r4 = time_us([&] { hipLaunchKernelGGL(read_k<4>, dim3(blocks), dim3(256), 0, 0, x, n4, out); });
            double r8 = time_us([&] { hipLaunchKernelGGL(read_k<8>, dim3(blocks), dim3(256), 0, 0, x, n4, out); });
            double r16 = time_us([&] { hipLaunchKernelGGL(read_k<16>, dim3(blocks), dim3(256), 0, 0, x, n4, out); });
            double w4 = time_us([&] { hipLaunchKernelGGL(write_k<4>, dim3(blocks), dim3(256), 0, 0, y, n4, 1.0f); });
            double w16 = time_us([&] { hipLaunchKernelGGL(write_k<16>, dim3(blocks), dim3(256), 0, 0, y, n4, 1.0f); });
            double c4 = time_us([&] { hipLaunchKernelGGL(copy_k<4>, dim3(blocks), dim3(256), 0, 0, x, y, n4); });
            double c8 = time_us([&] { hipLaunchKernelGGL(copy_k<8>, dim3(blocks), dim3(256), 0, 0, x, y, n4); });
            printf("  blocks %4d  read x4 %6.2f x8 %6.2f x16 %6.2f TB/s | write x4 %6.2f x16 %6.2f TB/s | copy (r+w) x4 %6.2f x8 %6.2f TB/s\n", blocks,
                   bytes / r4 / 1e6, bytes / r8 / 1e6, bytes / r16 / 1e6, bytes / w4 / 1e6, bytes / w16 / 1e6, 2.0 * bytes / c4 / 1e6, 2.0 * bytes / c8 / 1e6);
        }
        hipFree(x); hipFree(y); hipFree(out);
    }
    return 0;
}
