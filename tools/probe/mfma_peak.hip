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

// The same six-term product order as conv_x3_kernel (three A planes x three B planes -> hi*lo, lo*hi, mid*mid, mid*hi, hi*mid, hi*hi on
// four accumulators), operands held in registers.  DATA 0: all-zero operands; 1: the near-constant values of bf16_k; 2: operands drawn
// from a hash - random signs, exponents within 2^-2..2^1 and full mantissas, i.e. what the planes of a real activation look like.
// Same instruction stream in all three: what differs is the switching activity, i.e. the clock the power limit leaves.
__device__ inline unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int DATA>
__global__ __launch_bounds__(512) void bf16_x3_k(float* out, int iters, unsigned seed)
{
    bf16x8_t a[3][2], b[3][2];
    for (int pl = 0; pl < 3; ++pl)
        for (int t = 0; t < 2; ++t)
            for (int i = 0; i < 8; ++i) {
                float va = 0.f, vb = 0.f;
                if (DATA == 1) { va = 1.0f + threadIdx.x * 0.001f + i; vb = 1.0f - i; }
                if (DATA == 2) {
                    const unsigned h = hash32(seed + ((blockIdx.x * 512 + threadIdx.x) * 3 + pl) * 32 + t * 8 + i), g = hash32(h + 0x9e3779b9u);
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
    // conv_x3_kernel's launch shape: one block of eight waves per CU (two waves per SIMD), 24 MFMAs per iteration and wave
    for (int iters : {150, 4000}) {
        const double fl = (double)cus * 8 * iters * 24 * 2.0 * 32 * 32 * 16;
        const char* names[3] = {"zero operands", "near-constant operands", "random operands"};
        for (int d = 0; d < 3; ++d) {
            double ms = time_ms([&] {
                if (d == 0) hipLaunchKernelGGL(bf16_x3_k<0>, dim3(cus), dim3(512), 0, 0, out, iters, 7u);
                else if (d == 1) hipLaunchKernelGGL(bf16_x3_k<1>, dim3(cus), dim3(512), 0, 0, out, iters, 7u);
                else hipLaunchKernelGGL(bf16_x3_k<2>, dim3(cus), dim3(512), 0, 0, out, iters, 7u);
            }, 9);
            printf("bf16 six-term stream, 2 waves/SIMD, %-24s %5d iters: %8.3f ms  %7.1f TFLOP/s (%.3f of 2516.6)\n", names[d], iters, ms, fl / ms / 1e9,
                   fl / ms / 1e9 / 2516.6);
        }
    }
    return 0;
}
