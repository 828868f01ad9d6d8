// yardstick: what the matrix pipe SUSTAINS on this chip when a kernel does nothing but issue independent MFMAs from registers -
// the ceiling conv_x3_kernel (v_mfma_f32_32x32x16_bf16, six per fp32 product) and the fp32 kernels (v_mfma_f32_32x32x2_f32) are
// graded against, as a function of waves per SIMD.  The spec peaks (MI355X_MICROARCH.md: 2516.6 TF bf16, 157.3 TF fp32) assume
// 2.4 GHz; a chip-wide MFMA load runs at whatever clock the power limit leaves.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_peak.hip -o tools/probe/mfma_peak && tools/probe/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void bf16_k(float* out, int iters, float seed)
{
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x * 0.001f + i); b[i] = (__bf16)(seed - i); }
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 12345.678f) out[0] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void f32_k(float* out, int iters, float seed)
{
    float a = seed + threadIdx.x * 0.001f, b = seed - 1.f;
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 12345.678f) out[0] = s;
}

template <typename F>
static double time_ms(F launch, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    std::vector<float> ts;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main()
{
    float* out; hipMalloc(&out, 256);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    printf("%s: %d CUs, clock %d MHz\n", pr.gcnArchName, cus, pr.clockRate / 1000);
    // durations: ~0.2 ms (a convolution's length) and ~5 ms (sustained)
    for (int wps : {1, 2, 4}) {            // waves per SIMD: blocks of 256 threads = one wave per SIMD each
        for (int iters : {2000, 60000}) {
            const int blocks = cus * wps;
            const double fl_b = (double)blocks * 4 * iters * 4 * 2.0 * 32 * 32 * 16;       // 4 waves x NACC=4 MFMAs
            double ms = time_ms([&] { hipLaunchKernelGGL(bf16_k<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); }, 7);
            printf("bf16 32x32x16  %d wave(s)/SIMD  %6d iters: %8.3f ms  %7.1f TFLOP/s (%.3f of 2516.6)\n", wps, iters, ms, fl_b / ms / 1e9, fl_b / ms / 1e9 / 2516.6);
            const int it32 = iters / 2;
            const double fl_f = (double)blocks * 4 * it32 * 4 * 2.0 * 32 * 32 * 2;
            ms = time_ms([&] { hipLaunchKernelGGL(f32_k<4>, dim3(blocks), dim3(256), 0, 0, out, it32, 1.0f); }, 7);
            printf("f32  32x32x2   %d wave(s)/SIMD  %6d iters: %8.3f ms  %7.1f TFLOP/s (%.3f of 157.3)\n", wps, it32, ms, fl_f / ms / 1e9, fl_f / ms / 1e9 / 157.3);
        }
    }
    return 0;
}
