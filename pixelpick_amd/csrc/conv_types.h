// conv_types.h - the types and the store epilogue shared by the convolution kernels of conv_igemm.hip and conv_x3f.hip
// (one definition: the kernels take ConvParams / X3Operands by value, so the translation units must agree on them bit for bit).
#pragma once
#include "pp_common.h"
#include "bn_xchg.h"

namespace pp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kThreads = 256;
constexpr int kMaxTaps = 49;

struct ConvTaps {
    int n;                  // number of live taps
    int dh[kMaxTaps];       // input row offset of tap (already includes -pad / flip); int: wave-uniform s_load
    int dw[kMaxTaps];
    int widx[kMaxTaps];     // index of the tap in the weight tensor (kh*KW + kw)
};

// Training BatchNorm finished in the convolution's own epilogue (conv_epilogue_bn): the blocks of a column strip exchange their
// column sums exactly as the blocks of the single-launch BatchNorm kernel do (bn_xchg.h), then every block normalises the tile it
// still holds in registers.  part == NULL: off.
struct BnTrain {
    const float* gamma; const float* beta; float eps, momentum;
    float* running_mean; float* running_var; float* mean; float* invstd;
    const float* res; int64_t ldr; int act;
    float* y; int64_t ldy;              // the normalised (+ residual, activation) output; ConvParams::y receives the raw convolution
    xword* part; int* sync; int R;      // exchange area [strips][R][64] words, launch epoch, M tiles of the grid
    // backward form (a backward-data convolution that also runs the BatchNorm backward of the layer in FRONT of it, conv_epilogue_bn_bwd):
    // bx = the BatchNorm's input, mean / invstd are inputs, y receives the gradient of that input, dgamma / dbeta the parameter gradients
    const float* bx; int64_t ldbx; float* dgamma; float* dbeta;
    // backward form, a BatchNorm output with MORE consumers / a residual input: gin = the gradient the output already holds from the
    // consumers whose backward ran earlier (added to the tile before anything else, the `accumulate` of a plain backward-data), dres =
    // where the gradient of the BatchNorm's residual input goes (the masked gradient itself).  NULL: none.
    const float* gin; int64_t ldgin; float* dres; int64_t lddr;
};

struct ConvParams {
    const uint16_t* a_pre;   // bf16x3 planes of the A operand the CALLER already holds (pp_x3_split), or NULL: split here
    const uint16_t* b_pre;   // bf16x3 planes of the WEIGHTS the caller already holds (pp_x3_split_weights, the layout of this direction), or NULL
    const float* x;   // A-side activations (X for fwd/wgrad, dY for bwd-data)
    const float* w;   // HWIO weights
    const float* bias;
    float* y;         // output (Y, dX)
    int64_t ldx, ldy;
    int B, H, W;      // A-side spatial size
    int Ho, Wo;       // output spatial size (rows of the GEMM)
    int Ck;           // reduction channels (Cin for fwd, Cout for bwd-data)
    int Cn;           // output channels  (Cout for fwd, Cin for bwd-data)
    int Cin, Cout;    // weight tensor dims (for addressing)
    int stride;
    int64_t M;        // B*Ho*Wo
    int bwd_stride;   // backward-data of a strided conv: source row = (row + dh) / bwd_stride when divisible (else 1)
    int n_tiles;      // tiles along the output-channel axis (grid.x is 1-D: m_tiles * n_tiles blocks)
    int xcd_remap;
    int splits;       // split-K: grid.y slices of the (tap, channel-chunk) loop; > 1 -> partial sums go to `part`
    int ks_per_split;
    float* part;      // [splits][M][Cn] partial outputs (no bias)
    Epilogue epi;     // inference only: folded BatchNorm + residual + activation (all NULL / 0 in training)
    int accumulate;   // y += result (backward-data into a gradient that already holds the residual branch's part)
    int tap_inner;    // K loop order (A/B knob)
    const float* in_scale;   // forward of a 1x1 / pad-0 convolution BEHIND a training BatchNorm whose apply pass was skipped: the A
    const float* in_shift;   // operand is act(fma(x, in_scale[c], in_shift[c])) (bn_apply_kernel's arithmetic), applied where the
    int in_act;              // operand is read.  NULL: x as it is.  (conv_igemm_kernel VEC path, conv1x1_ksplit_dma_kernel)
    // several convolutions' backward-data as ONE implicit GEMM (pp_conv2d_bwd_data_multi: the ASPP branches, aspp.py:49-57, all read
    // one input): tap t of the merged reduction reads the A operand tap_coff[t] channels into its row and its weights tap_woff[t]
    // elements behind `w` (conv_igemm_dma_kernel only; 0: the tap table's own addressing)
    int multi;
    int tap_coff[32], tap_woff[32];
    BnTrain bn;
    float* stats;     // training forward in front of a BatchNorm: per-wave column sums / sums of squares of the stored outputs,
                      // [rows_partial][2][Cn], rows_partial = m0 / (TM*32) + wm (see conv_epilogue); NULL: none
    ConvTaps taps;
};

// ---- epilogue shared by the register-staged and the LDS-DMA kernels --------------------------------------
template <int TM, int TN>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x16 (&acc)[TM][TN], int64_t m0, int n0, int wm, int wn)
{
    const int tid = threadIdx.x;
    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + (wn * TN + tn) * 32 + l31;
        if (n >= p.Cn) continue;
        const bool final_pass = p.splits <= 1;
        const float bv = (p.bias && final_pass) ? p.bias[n] : 0.0f;
        const bool affine = final_pass && p.epi.gamma != nullptr;
        float sc = 1.0f, sf = 0.0f;
        if (affine) {
            const float is = 1.0f / sqrtf(p.epi.var[n] + p.epi.eps);
            sc = p.epi.gamma[n] * is;
            sf = p.epi.beta[n] - p.epi.mean[n] * sc;
        }
        const float* res = final_pass ? p.epi.res : nullptr;
        const int act = final_pass ? p.epi.act : 0;
        float* out = p.splits > 1 ? p.part + (int64_t)blockIdx.y * p.M * p.Cn : p.y;
        const int64_t ldo = p.splits > 1 ? (int64_t)p.Cn : p.ldy;
        float s1 = 0.0f, s2 = 0.0f;           // column sum / sum of squares of what this lane stores (BatchNorm statistics)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (m < p.M) {
                    float o = acc[tm][tn][r] + bv;
                    if (affine) o = fmaf(o, sc, sf);
                    if (res) o += res[m * p.epi.ldr + n];
                    if (final_pass && p.accumulate) o += out[m * ldo + n];
                    o = epi_act(o, act);
                    out[m * ldo + n] = o;
                    s1 += o;
                    s2 = fmaf(o, o, s2);
                }
            }
        }
        if (p.stats && final_pass) {
            // lanes l and l+32 hold the same column (rows 4*hh apart): one fixed-order add, then one store per column
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (hh == 0) {
                const int64_t pr = m0 / (TM * 32) + wm;
                p.stats[(pr * 2 + 0) * p.Cn + n] = s1;
                p.stats[(pr * 2 + 1) * p.Cn + n] = s2;
            }
        }
    }
}

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

struct X3Operands {
    const uint16_t* a;        // A planes [3][Kp/16][rows_a + 1][16] (row rows_a = zeros)
    const uint16_t* b;        // B planes [3][Kp/16][ntaps * n_rows + 1][16] (last row = zeros); row = tap_w * n_rows + n
    int64_t a_plane, b_plane; // elements per plane
    int Kp;                   // reduction channels per tap, padded to a multiple of 16 (pads are zeros in both operands)
    int n_rows;               // B rows per tap (= output channels of this GEMM)
    uint32_t a_zero, b_zero;  // byte offset of the zero row inside a plane (chunk 0)
    uint32_t a_chunk, b_chunk;   // bytes per 16-channel chunk: (rows + 1) * 32
    int col_base;             // first output column of this launch (ragged widths run as a 128-wide launch + a 64-wide one)
};


// conv_x3f.hip (test build only): conv_x3_kernel's tile forms with the A operand read as fp32 and split into its three bf16 planes INSIDE the kernel
// (no x3_split_kernel launch, no A planes in memory).  m256: 256-row tiles (eight waves), else 128; n128: 128-column tiles, else 64.
int launch_conv_x3f(const ConvParams& p, const X3Operands& o, bool m256, bool n128, unsigned blocks, hipStream_t st);
int conv_x3f_supported(const ConvParams& p);

// gemm_pw.hip: pointwise convolutions as a plain row-major GEMM (tile forms 0..5, see the file)
int launch_gemm_pw(const float* a, int64_t lda, const float* b, int64_t ldb, const float* bias, float* c, int64_t ldc, int64_t M, int N, int K,
                   int accumulate, int form, int xcd_remap, hipStream_t st, int b_transposed = 0);
int gemm_pw_supported(const float* a, int64_t lda, const float* b, int64_t ldb, const float* c, int64_t ldc, int64_t M, int N, int K);
int gemm_pw_tile_rows(int form);
int gemm_pw_tile_cols(int form);

}  // namespace pp
