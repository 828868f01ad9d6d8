// metrics.hip — device-side step metrics: argmax over classes + confusion-matrix histogram.
//
// Replaces, per train/val step, `logits.argmax(dim=1)`, two full-map D2H copies and a numpy bincount
// (model.py:124-125,194-196; utils/metrics.py:168-177 RunningScore._fast_hist/update): only the C x C
// histogram ever leaves the device.  Integer atomics: deterministic.
#include "pp_common.h"

namespace pp {

__global__ __launch_bounds__(256) void confusion_kernel(const float* logits, const int64_t* target, int B, int C, int64_t HW,
                                                        int64_t sB, int64_t sC, unsigned long long* hist)
{
    extern __shared__ unsigned int sh[];   // C*C block-private counters
    for (int i = threadIdx.x; i < C * C; i += 256) sh[i] = 0u;
    __syncthreads();
    const int64_t total = (int64_t)B * HW;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t t = target[e];
        if (t < 0 || t >= C) continue;                       // utils/metrics.py:169 mask
        const int64_t b = e / HW, pix = e - b * HW;
        const float* px = logits + b * sB + pix;
        float m = px[0];
        int am = 0;
        for (int c = 1; c < C; ++c) {
            const float v = px[c * sC];
            if (v > m) { m = v; am = c; }                    // first maximum, like torch.argmax on CPU
        }
        atomicAdd(&sh[(int)t * C + am], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += 256)
        if (sh[i]) atomicAdd(&hist[i], (unsigned long long)sh[i]);
}

}  // namespace pp

using namespace pp;

extern "C" {

int pp_confusion_matrix_update(const float* logits, int B, int C, int64_t HW, int64_t sB, int64_t sC, const int64_t* target,
                               int64_t* hist, pp_stream_t stream)
{
    if (!logits || !target || !hist) return fail(PP_ERR_BAD_ARG, "confusion_matrix: null");
    if (C < 1 || C > 104) return fail(PP_ERR_UNSUPPORTED, "confusion_matrix: C=%d (LDS histogram holds up to 104 classes)", C);
    int64_t nblk = cdiv((int64_t)B * HW, 256 * 8);
    if (nblk > 2048) nblk = 2048;
    if (nblk < 1) nblk = 1;
    hipLaunchKernelGGL(confusion_kernel, dim3((unsigned)nblk), dim3(256), (size_t)C * C * 4, as_stream(stream), logits, target, B, C,
                       HW, sB, sC, reinterpret_cast<unsigned long long*>(hist));
    return check_launch("confusion_kernel");
}

}  // extern "C"
