// augment.hip — the training-time data augmentation of datasets/base_dataset.py:48-141 on the device.
//
// The reference augments on the host through torchvision wrappers around PIL (geometric: TF.resize / TF.pad / TF.crop /
// TF.hflip on PIL images and uint8 tensors; photometric: ColorJitter / RandomGrayscale = PIL ImageEnhance / convert("L"))
// and cv2.GaussianBlur.  The kernels below reproduce that arithmetic on uint8 images exactly:
//   * PIL's BILINEAR resize is a separable triangle filter with support max(scale, 1), coefficients normalised in double
//     and quantised to 22 fractional bits, one uint8 rounding after the horizontal and one after the vertical pass
//     (libImaging/Resample.c).  The host builds the (bounds, coefficient) tables exactly as precompute_coeffs() does; the
//     kernels only evaluate  clip8((2^21 + sum px*k) >> 22).
//   * label maps are resized with PIL's NEAREST (index = int(a*(x+0.5)) accumulated in double) and query masks with
//     torch's (index = floor(x * float(in/out))): the reference really uses both; the host passes each as an index table.
//   * pad (mean colour / ignore_index / 0), crop and horizontal flip are folded into the gather of the second pass.
//   * ImageEnhance blends truncate towards zero after clipping; "L" = (19595 R + 38470 G + 7471 B + 0x8000) >> 16.
// Random parameters are drawn on the host in the reference's order (pixelpick_amd/augment.py).
#include "pp_common.h"

namespace pp {

constexpr int kAT = 256;
constexpr int kPrecisionBits = 22;       // libImaging/Resample.c PRECISION_BITS = 32 - 8 - 2

__device__ __forceinline__ uint8_t clip8(int64_t v)
{
    v >>= kPrecisionBits;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: dst[r][xx][c] = clip8(2^21 + sum_k src[r][xmin+k][c] * kk[xx][k])
__global__ __launch_bounds__(kAT) void aug_resample_h_kernel(const uint8_t* src, int H, int W, const int32_t* bounds, const int32_t* kk,
                                                            int ksize, int Wout, uint8_t* dst)
{
    const int64_t total = (int64_t)H * Wout;
    for (int64_t e = (int64_t)blockIdx.x * kAT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kAT) {
        const int r = (int)(e / Wout), xx = (int)(e - (int64_t)r * Wout);
        const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
        const int32_t* k = kk + (int64_t)xx * ksize;
        const uint8_t* s = src + ((int64_t)r * W + xmin) * 3;
        int64_t a0 = 1 << (kPrecisionBits - 1), a1 = a0, a2 = a0;
        for (int x = 0; x < n; ++x) {
            const int64_t w = k[x];
            a0 += (int64_t)s[3 * x + 0] * w;
            a1 += (int64_t)s[3 * x + 1] * w;
            a2 += (int64_t)s[3 * x + 2] * w;
        }
        uint8_t* d = dst + ((int64_t)r * Wout + xx) * 3;
        d[0] = clip8(a0); d[1] = clip8(a1); d[2] = clip8(a2);
    }
}

struct CropGeom {
    int h_rs, w_rs;          // size of the (virtual) resized image; beyond it lies the constant padding
    int start_h, start_w;    // crop origin in the padded image
    int ch, cw;              // output size
    int flip;                // horizontal flip of the crop
};

// vertical pass evaluated only at the crop: out[i][j][c] = resized[i+start_h][jj+start_w][c] or the fill colour
__global__ __launch_bounds__(kAT) void aug_vcrop_kernel(const uint8_t* tmp /*[H, w_rs, 3]*/, const int32_t* bounds, const int32_t* kk,
                                                       int ksize, CropGeom g, int fill_r, int fill_g, int fill_b, uint8_t* out)
{
    const int64_t total = (int64_t)g.ch * g.cw;
    for (int64_t e = (int64_t)blockIdx.x * kAT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kAT) {
        const int i = (int)(e / g.cw), j = (int)(e - (int64_t)i * g.cw);
        const int jj = g.flip ? g.cw - 1 - j : j;
        const int r = i + g.start_h, c = jj + g.start_w;
        uint8_t* d = out + e * 3;
        if (r >= g.h_rs || c >= g.w_rs) { d[0] = (uint8_t)fill_r; d[1] = (uint8_t)fill_g; d[2] = (uint8_t)fill_b; continue; }
        const int ymin = bounds[2 * r], n = bounds[2 * r + 1];
        const int32_t* k = kk + (int64_t)r * ksize;
        int64_t a0 = 1 << (kPrecisionBits - 1), a1 = a0, a2 = a0;
        for (int y = 0; y < n; ++y) {
            const uint8_t* s = tmp + ((int64_t)(ymin + y) * g.w_rs + c) * 3;
            const int64_t w = k[y];
            a0 += (int64_t)s[0] * w; a1 += (int64_t)s[1] * w; a2 += (int64_t)s[2] * w;
        }
        d[0] = clip8(a0); d[1] = clip8(a1); d[2] = clip8(a2);
    }
}

// label map (PIL NEAREST tables) and query mask (torch nearest tables) through the same pad / crop / flip
__global__ __launch_bounds__(kAT) void aug_labels_kernel(const uint8_t* y, const uint8_t* q, int W, const int32_t* ty, const int32_t* tx,
                                                        const int32_t* qy, const int32_t* qx, CropGeom g, int ignore_index,
                                                        int64_t* y_out, uint8_t* q_out)
{
    const int64_t total = (int64_t)g.ch * g.cw;
    for (int64_t e = (int64_t)blockIdx.x * kAT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kAT) {
        const int i = (int)(e / g.cw), j = (int)(e - (int64_t)i * g.cw);
        const int jj = g.flip ? g.cw - 1 - j : j;
        const int r = i + g.start_h, c = jj + g.start_w;
        const bool pad = r >= g.h_rs || c >= g.w_rs;
        if (y_out) y_out[e] = pad ? (int64_t)ignore_index : (int64_t)y[(int64_t)ty[r] * W + tx[c]];
        if (q_out) q_out[e] = pad ? (uint8_t)0 : (uint8_t)(q[(int64_t)qy[r] * W + qx[c]] != 0);
    }
}

__device__ __forceinline__ int luma(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

// ImagingBlend(degenerate, image, factor) on one value: in1 + alpha*(in2-in1) in float, clipped, truncated
__device__ __forceinline__ uint8_t blend8(int deg, int px, float alpha)
{
#pragma clang fp contract(off)              // PIL rounds the product before the add (plain C on the host, no fma)
    const float prod = alpha * (float)(px - deg);
    const float t = (float)deg + prod;
    if (alpha >= 0.0f && alpha <= 1.0f) return (uint8_t)t;
    return t <= 0.0f ? (uint8_t)0 : (t >= 255.0f ? (uint8_t)255 : (uint8_t)t);
}

__global__ __launch_bounds__(kAT) void aug_luma_sum_kernel(const uint8_t* img, int64_t n, unsigned long long* sum)
{
    __shared__ unsigned long long sh[kAT];
    unsigned long long s = 0;
    for (int64_t e = (int64_t)blockIdx.x * kAT + threadIdx.x; e < n; e += (int64_t)gridDim.x * kAT)
        s += (unsigned long long)luma(img[3 * e], img[3 * e + 1], img[3 * e + 2]);
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = kAT / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(sum, sh[0]);          // integer sum: order-independent, deterministic
}

// op: 0 brightness, 1 contrast (degenerate = int(mean L + 0.5), from *lsum / n), 2 saturation, 3 hue, 4 grayscale
__global__ __launch_bounds__(kAT) void aug_jitter_kernel(uint8_t* img, int64_t n, int op, float factor, const unsigned long long* lsum)
{
    int mean_l = 0;
    if (op == 1) mean_l = (int)((double)(*lsum) / (double)n + 0.5);
    const int hshift = op == 3 ? (int)(uint8_t)(int)(factor * 255.0f) : 0;       // np.uint8(hue_factor * 255): wraps
    for (int64_t e = (int64_t)blockIdx.x * kAT + threadIdx.x; e < n; e += (int64_t)gridDim.x * kAT) {
        uint8_t* p = img + 3 * e;
        const int r = p[0], g = p[1], b = p[2];
        if (op == 0) {
            p[0] = blend8(0, r, factor); p[1] = blend8(0, g, factor); p[2] = blend8(0, b, factor);
        } else if (op == 1) {
            p[0] = blend8(mean_l, r, factor); p[1] = blend8(mean_l, g, factor); p[2] = blend8(mean_l, b, factor);
        } else if (op == 2) {
            const int l = luma(r, g, b);
            p[0] = blend8(l, r, factor); p[1] = blend8(l, g, factor); p[2] = blend8(l, b, factor);
        } else if (op == 4) {
            const uint8_t l = (uint8_t)luma(r, g, b);
            p[0] = l; p[1] = l; p[2] = l;
        } else {
            // libImaging/Convert.c rgb2hsv_row -> uint8 H += shift (wraps) -> hsv2rgb
            const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
            uint8_t uh = 0, us = 0;
            const uint8_t uv = (uint8_t)maxc;
            if (minc != maxc) {
                // libImaging/Convert.c rgb2hsv_row, including which intermediates are float and which double
                const float cr = (float)(maxc - minc);
                const float s = cr / (float)maxc;
                const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
                float h;
                if (r == maxc) h = bc - gc;
                else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
                else h = (float)(4.0 + (double)gc - (double)rc);
                h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
                const int ih = (int)((double)h * 255.0), is = (int)((double)s * 255.0);
                uh = (uint8_t)(ih < 0 ? 0 : (ih > 255 ? 255 : ih));
                us = (uint8_t)(is < 0 ? 0 : (is > 255 ? 255 : is));
            }
            uh = (uint8_t)(uh + hshift);
            if (us == 0) { p[0] = uv; p[1] = uv; p[2] = uv; continue; }
            const float hh = (float)uh * 6.0f / 255.0f;
            const int i = (int)floorf(hh);
            const float f = hh - (float)i;
            const float fs = (float)us / 255.0f;
            const float v = (float)uv;
            auto rnd = [](float x) { int t = (int)roundf(x); return (uint8_t)(t < 0 ? 0 : (t > 255 ? 255 : t)); };
            const uint8_t pp = rnd(v * (1.0f - fs)), qq = rnd(v * (1.0f - fs * f)), tt = rnd(v * (1.0f - fs * (1.0f - f)));
            uint8_t ro, go, bo;
            switch (i % 6) {
                case 0: ro = uv; go = tt; bo = pp; break;
                case 1: ro = qq; go = uv; bo = pp; break;
                case 2: ro = pp; go = uv; bo = tt; break;
                case 3: ro = pp; go = qq; bo = uv; break;
                case 4: ro = tt; go = pp; bo = uv; break;
                default: ro = uv; go = pp; bo = qq; break;
            }
            p[0] = ro; p[1] = go; p[2] = bo;
        }
    }
}

// one pass of a separable Gaussian blur on HWC uint8 (cv2.GaussianBlur semantics: float kernel, BORDER_REFLECT_101,
// round-half-to-even saturate); axis 1 = along W, axis 0 = along H; float intermediate between the passes
__global__ __launch_bounds__(kAT) void aug_blur_kernel(const uint8_t* src8, const float* srcf, int H, int W, const float* k, int ks, int axis,
                                                      float* dstf, uint8_t* dst8)
{
    const int64_t total = (int64_t)H * W;
    const int half = ks / 2;
    for (int64_t e = (int64_t)blockIdx.x * kAT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kAT) {
        const int r = (int)(e / W), c = (int)(e - (int64_t)r * W);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        const int n = axis == 1 ? W : H;
        for (int t = 0; t < ks; ++t) {
            int p = (axis == 1 ? c : r) + t - half;
            if (n == 1) p = 0;
            else { while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p; }        // reflect 101
            const int64_t o = ((axis == 1 ? (int64_t)r * W + p : (int64_t)p * W + c)) * 3;
            const float w = k[t];
            if (src8) { a0 += w * (float)src8[o]; a1 += w * (float)src8[o + 1]; a2 += w * (float)src8[o + 2]; }
            else { a0 += w * srcf[o]; a1 += w * srcf[o + 1]; a2 += w * srcf[o + 2]; }
        }
        if (dstf) { dstf[3 * e] = a0; dstf[3 * e + 1] = a1; dstf[3 * e + 2] = a2; }
        else {
            auto sat = [](float x) { const float t = rintf(x); return (uint8_t)(t < 0.f ? 0.f : (t > 255.f ? 255.f : t)); };
            dst8[3 * e] = sat(a0); dst8[3 * e + 1] = sat(a1); dst8[3 * e + 2] = sat(a2);
        }
    }
}

// cv2.GaussianBlur on 8-bit images as OpenCV >= 3.4.2 / 4.x computes it (smooth.dispatch.cpp GaussianBlurFixedPoint): taps in 8.8 fixed
// point (sum exactly 256), row pass exact in 16 bits (8.8), column pass in 32 bits (16.16), (acc + 0x8000) >> 16 - round half up.
// Pure integer arithmetic: no dependence on summation order.  axis 1: src8 -> dst16 (row pass), axis 0: src16 -> dst8 (column pass).
__global__ __launch_bounds__(kAT) void aug_blur_q8_kernel(const uint8_t* src8, const uint16_t* src16, int H, int W, const uint16_t* k, int ks,
                                                         int axis, uint16_t* dst16, uint8_t* dst8)
{
    const int64_t total = (int64_t)H * W;
    const int half = ks / 2;
    for (int64_t e = (int64_t)blockIdx.x * kAT + threadIdx.x; e < total; e += (int64_t)gridDim.x * kAT) {
        const int r = (int)(e / W), c = (int)(e - (int64_t)r * W);
        uint32_t a0 = 0, a1 = 0, a2 = 0;
        const int n = axis == 1 ? W : H;
        for (int t = 0; t < ks; ++t) {
            int p = (axis == 1 ? c : r) + t - half;
            if (n == 1) p = 0;
            else { while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p; }        // BORDER_REFLECT_101
            const int64_t o = ((axis == 1 ? (int64_t)r * W + p : (int64_t)p * W + c)) * 3;
            const uint32_t w = k[t];
            if (src8) { a0 += w * src8[o]; a1 += w * src8[o + 1]; a2 += w * src8[o + 2]; }
            else { a0 += w * src16[o]; a1 += w * src16[o + 1]; a2 += w * src16[o + 2]; }
        }
        if (dst16) { dst16[3 * e] = (uint16_t)a0; dst16[3 * e + 1] = (uint16_t)a1; dst16[3 * e + 2] = (uint16_t)a2; }
        else {
            auto sat = [](uint32_t x) { const uint32_t t = (x + 0x8000u) >> 16; return (uint8_t)(t > 255u ? 255u : t); };
            dst8[3 * e] = sat(a0); dst8[3 * e + 1] = sat(a1); dst8[3 * e + 2] = sat(a2);
        }
    }
}

// TF.normalize(TF.to_tensor(x), mean, std): HWC uint8 -> CHW float32, (v / 255 - mean) / std
__global__ __launch_bounds__(kAT) void aug_to_tensor_kernel(const uint8_t* img, int64_t n, float m0, float m1, float m2, float s0, float s1,
                                                           float s2, float* out)
{
    for (int64_t e = (int64_t)blockIdx.x * kAT + threadIdx.x; e < n; e += (int64_t)gridDim.x * kAT) {
        out[e] = ((float)img[3 * e] / 255.0f - m0) / s0;
        out[n + e] = ((float)img[3 * e + 1] / 255.0f - m1) / s1;
        out[2 * n + e] = ((float)img[3 * e + 2] / 255.0f - m2) / s2;
    }
}

static inline unsigned aug_grid(int64_t total)
{
    int64_t b = (total + kAT - 1) / kAT;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace pp

using namespace pp;

extern "C" {

int pp_aug_resample_h(const uint8_t* src, int H, int W, const int32_t* bounds, const int32_t* kk, int ksize, int Wout, uint8_t* dst,
                      pp_stream_t stream)
{
    if (!src || !bounds || !kk || !dst) return fail(PP_ERR_BAD_ARG, "aug_resample_h: null");
    if (H <= 0 || W <= 0 || Wout <= 0 || ksize <= 0) return fail(PP_ERR_BAD_ARG, "aug_resample_h: bad shape");
    hipLaunchKernelGGL(aug_resample_h_kernel, dim3(aug_grid((int64_t)H * Wout)), dim3(kAT), 0, as_stream(stream), src, H, W, bounds, kk,
                       ksize, Wout, dst);
    return check_launch("aug_resample_h_kernel");
}

int pp_aug_vcrop(const uint8_t* tmp, const int32_t* bounds, const int32_t* kk, int ksize, int h_rs, int w_rs, int start_h, int start_w,
                 int ch, int cw, int flip, int fill_r, int fill_g, int fill_b, uint8_t* out, pp_stream_t stream)
{
    if (!tmp || !bounds || !kk || !out) return fail(PP_ERR_BAD_ARG, "aug_vcrop: null");
    if (h_rs <= 0 || w_rs <= 0 || ch <= 0 || cw <= 0 || start_h < 0 || start_w < 0) return fail(PP_ERR_BAD_ARG, "aug_vcrop: bad geometry");
    const CropGeom g{h_rs, w_rs, start_h, start_w, ch, cw, flip ? 1 : 0};
    hipLaunchKernelGGL(aug_vcrop_kernel, dim3(aug_grid((int64_t)ch * cw)), dim3(kAT), 0, as_stream(stream), tmp, bounds, kk, ksize, g,
                       fill_r, fill_g, fill_b, out);
    return check_launch("aug_vcrop_kernel");
}

int pp_aug_labels(const uint8_t* y, const uint8_t* q, int W, const int32_t* ty, const int32_t* tx, const int32_t* qy, const int32_t* qx,
                  int h_rs, int w_rs, int start_h, int start_w, int ch, int cw, int flip, int ignore_index, int64_t* y_out,
                  uint8_t* q_out, pp_stream_t stream)
{
    if ((y_out && (!y || !ty || !tx)) || (q_out && (!q || !qy || !qx)) || (!y_out && !q_out)) return fail(PP_ERR_BAD_ARG, "aug_labels: null");
    const CropGeom g{h_rs, w_rs, start_h, start_w, ch, cw, flip ? 1 : 0};
    hipLaunchKernelGGL(aug_labels_kernel, dim3(aug_grid((int64_t)ch * cw)), dim3(kAT), 0, as_stream(stream), y, q, W, ty, tx, qy, qx, g,
                       ignore_index, y_out, q_out);
    return check_launch("aug_labels_kernel");
}

int pp_aug_jitter(uint8_t* img, int64_t n_pixels, int op, float factor, unsigned long long* scratch_sum, pp_stream_t stream)
{
    if (!img || n_pixels <= 0 || op < 0 || op > 4) return fail(PP_ERR_BAD_ARG, "aug_jitter: bad argument");
    hipStream_t st = as_stream(stream);
    if (op == 1) {
        if (!scratch_sum) return fail(PP_ERR_WORKSPACE, "aug_jitter: contrast needs an 8-byte scratch word");
        if (hipMemsetAsync(scratch_sum, 0, 8, st) != hipSuccess) return fail(PP_ERR_LAUNCH, "aug_jitter: memset failed");
        hipLaunchKernelGGL(aug_luma_sum_kernel, dim3(aug_grid(n_pixels) > 256 ? 256 : aug_grid(n_pixels)), dim3(kAT), 0, st, img, n_pixels,
                           scratch_sum);
        if (int rc = check_launch("aug_luma_sum_kernel")) return rc;
    }
    hipLaunchKernelGGL(aug_jitter_kernel, dim3(aug_grid(n_pixels)), dim3(kAT), 0, st, img, n_pixels, op, factor, scratch_sum);
    return check_launch("aug_jitter_kernel");
}

int pp_aug_blur(uint8_t* img, int H, int W, const float* kernel, int ks, float* scratch /*[H*W*3]*/, pp_stream_t stream)
{
    if (!img || !kernel || !scratch || H <= 0 || W <= 0 || ks < 1 || ks % 2 == 0) return fail(PP_ERR_BAD_ARG, "aug_blur: bad argument");
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(aug_blur_kernel, dim3(aug_grid((int64_t)H * W)), dim3(kAT), 0, st, img, (const float*)nullptr, H, W, kernel, ks, 1,
                       scratch, (uint8_t*)nullptr);
    if (int rc = check_launch("aug_blur_kernel")) return rc;
    hipLaunchKernelGGL(aug_blur_kernel, dim3(aug_grid((int64_t)H * W)), dim3(kAT), 0, st, (const uint8_t*)nullptr, scratch, H, W, kernel, ks,
                       0, (float*)nullptr, img);
    return check_launch("aug_blur_kernel");
}

int pp_aug_blur_q8(uint8_t* img, int H, int W, const uint16_t* kernel_q8, int ks, uint16_t* scratch /*[H*W*3]*/, pp_stream_t stream)
{
    if (!img || !kernel_q8 || !scratch || H <= 0 || W <= 0 || ks < 1 || ks % 2 == 0) return fail(PP_ERR_BAD_ARG, "aug_blur_q8: bad argument");
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(aug_blur_q8_kernel, dim3(aug_grid((int64_t)H * W)), dim3(kAT), 0, st, img, (const uint16_t*)nullptr, H, W, kernel_q8, ks, 1,
                       scratch, (uint8_t*)nullptr);
    if (int rc = check_launch("aug_blur_q8_kernel")) return rc;
    hipLaunchKernelGGL(aug_blur_q8_kernel, dim3(aug_grid((int64_t)H * W)), dim3(kAT), 0, st, (const uint8_t*)nullptr, scratch, H, W, kernel_q8, ks,
                       0, (uint16_t*)nullptr, img);
    return check_launch("aug_blur_q8_kernel");
}

int pp_aug_to_tensor(const uint8_t* img, int64_t n_pixels, const float* mean3, const float* std3, float* out, pp_stream_t stream)
{
    if (!img || !out || !mean3 || !std3 || n_pixels <= 0) return fail(PP_ERR_BAD_ARG, "aug_to_tensor: bad argument");
    hipLaunchKernelGGL(aug_to_tensor_kernel, dim3(aug_grid(n_pixels)), dim3(kAT), 0, as_stream(stream), img, n_pixels, mean3[0], mean3[1],
                       mean3[2], std3[0], std3[1], std3[2], out);
    return check_launch("aug_to_tensor_kernel");
}

}  // extern "C"
