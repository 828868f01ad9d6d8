// gemm_pw.hip — pointwise (1x1, stride 1, no padding) convolutions as a plain row-major GEMM on the fp32 MFMA pipe.
//
//   C[M][N] (+)= A[M][K] x B[K][N] (+ bias[N])      A = NHWC activations (pixel stride lda), B = the HWIO weight of a 1x1 convolution
//                                                    ([Cin][Cout] as stored), C = NHWC output (pixel stride ldc)
//
// Replaces conv_igemm_dma_kernel for the dense pointwise layers of the reference's networks that have enough rows to give every CU a
// large tile: networks/backbones/resnet_models.py:58-94 (Bottleneck conv1 / conv3 and the downsample), networks/decoders.py:25-77,
// networks/aspp.py:49,73-75, networks/mobilenet_v2.py:42,56 - and, through a transposed copy of the weight, their backward-data
// (model.py:121).
//
// Why a second kernel (profiles/r06_vendor_sgemm_kernels.txt): the implicit-GEMM kernel (64x64 / 128x128 tiles, K steps of 16, one
// barrier per step, 3-4 blocks per CU) reaches 0.46-0.60 of the fp32 MFMA peak on the ResNet50 Bottleneck shapes at 8192 rows, where
// the vendor library reaches 0.67-0.86 with ONE work-group per CU: macro tiles chosen so that the tile count is the CU count, K steps
// of 32-64 per barrier, accumulators in AGPRs.  This kernel is that design, hand-written for gfx950:
//   * 256 threads = 4 waves as WM x WN, a wave owns TM x TN MFMA tiles of 32 x 32 (v_mfma_f32_32x32x2_f32): wave tiles up to 64 x 128
//     (128 accumulator registers - the compiler keeps them in AGPRs at one wave per SIMD);
//   * K step 32 (the 128 x 256 / 256 x 128 tiles: 128 MFMAs per wave between barriers) or 64 (the smaller tiles) per barrier, ring of
//     three LDS stages filled by LDS-DMA (global_load_lds_dwordx4) two steps ahead, one piece behind each MFMA group, counted s_waitcnt vmcnt;
//   * A tile in MK form (k contiguous, 128- / 256-byte rows, quad slots XOR-swizzled by (row >> 1) & 7 / row & 15): one conflict-free ds_read_b128 per
//     row tile gives the four k values a lane feeds to four consecutive MFMAs (lane half h consumes k = 8c + 4h + j);
//   * B tile in KN form exactly as it lies in memory (every DMA piece is one contiguous KiB of a weight row): a lane reads TN
//     CONSECUTIVE columns of one k with one ds_read_b(32 TN), i.e. lane l of column tile t owns column TN*l + t of the wave's strip -
//     six LDS reads per 32 MFMAs instead of eighteen, and the epilogue stores TN consecutive floats per row (float4 for TN = 4)
//     instead of 64 scattered dwords;
//   * no tap table, no im2col arithmetic: one multiply-add per DMA piece and step.
// Results are bit-identical to conv_igemm_dma_kernel's (same MFMA, same k order within an accumulator: k ascending).
#include <type_traits>

#include "conv_types.h"

namespace pp {

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__device__ __attribute__((aligned(16))) float g_gemm_zero[4] = {0.f, 0.f, 0.f, 0.f};

struct GemmParams {
    const float* a; const float* b; const float* bias; float* c;
    int64_t lda, ldb, ldc;
    int64_t M;
    int N, K;
    int n_tiles;          // tiles along N (grid is 1-D: m_tiles * n_tiles)
    int xcd_remap;
    int accumulate;       // c += result
};

__device__ __forceinline__ void gemm_glds16(const float* gsrc, uint32_t lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// BT: the B operand is given TRANSPOSED - b[n][k], k contiguous (backward-data of a 1x1 convolution: dX = dY x W^T with W = [Cin][Cout], i.e.
// b = W, n = Cin, k = Cout).  Its tile then lies in LDS as the A tile does (rows = n, swizzled quads of k, one ds_read_b128 per column tile
// and chunk) and lane l of column tile t owns column 32 t + l as in the implicit-GEMM kernels: scalar stores in the epilogue.
template <int TM, int TN, int WM, int WN, int BK, bool BT = false>
__global__ __launch_bounds__(256, 1) void gemm_pw_kernel(GemmParams p)
{
    static_assert(WM * WN == 4, "four waves");
    static_assert(BT || TN == 1 || TN == 2 || TN == 4, "a lane reads TN consecutive columns");
    static_assert(BK == 32 || BK == 64, "K step");
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NSTAGE = 3;
    constexpr int NCH = BK / 8;                                        // chunks of 8 k per step
    constexpr int QR = BK / 4;                                         // 16-byte quads per A row of a stage (8 | 16)
    constexpr int A_FLOATS = BM * BK, B_FLOATS = BK * BN, STAGE_FLOATS = A_FLOATS + B_FLOATS;
    constexpr int PA = A_FLOATS / 256 / 4, PB = B_FLOATS / 256 / 4, PW = PA + PB;      // 1-KiB DMA pieces per wave and step
    static_assert(PA >= 1 && PB >= 1 && A_FLOATS % 1024 == 0 && B_FLOATS % 1024 == 0, "tile");
    constexpr int NSLOT = (NCH - 1) * 4;                               // MFMA groups of a step behind which a piece of the DMA goes out
    static_assert(PW <= 2 * NSLOT, "at most two pieces behind one MFMA group");
    constexpr int NQB = BN / 4;                                        // quads per k row of the B tile
    __shared__ __attribute__((aligned(1024))) float smem[NSTAGE * STAGE_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int mt, nt;
    {
        const int ntn = p.n_tiles, nblk = gridDim.x, bid = blockIdx.x, per_xcd = nblk / 8;
        if (p.xcd_remap && per_xcd * 8 == nblk) {
            const int lin = (bid & 7) * per_xcd + (bid >> 3);
            mt = lin / ntn; nt = lin - mt * ntn;
        } else {
            mt = bid / ntn; nt = bid - mt * ntn;
        }
    }
    const int64_t m0 = (int64_t)mt * BM;
    const int n0 = nt * BN;
    const int K = __builtin_amdgcn_readfirstlane(p.K);
    const int nsteps = (K + BK - 1) / BK, nfull = K / BK;
    const float* zero = g_gemm_zero;
    asm volatile("" : "+v"(zero));
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);

    // ---- DMA addressing.  A piece i of this wave = rows (wave + 4 i) * (64 / QR) ..., lane -> (row, quad slot); the slot holds logical
    //      quad q = slot ^ (swizzle of the row).  B piece j = linear quads (wave + 4 j) * 64 + lane of the [BK][BN] image.  Every piece
    //      keeps a RUNNING source pointer: a full step costs it one 64-bit add (rows / columns outside the problem point at a zero quad
    //      and do not move); only a ragged LAST step checks k.
    constexpr int RPP = 64 / QR;                                       // A rows per piece (8 | 4)
    const float* a_cur[PA]; int a_inc[PA], a_k[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int piece = wave + 4 * i;
        const int row = piece * RPP + lane / QR;
        const int q = (lane % QR) ^ (QR == 8 ? ((row >> 1) & 7) : (row & 15));
        a_k[i] = q * 4;
        const bool ok = m0 + row < p.M;
        a_cur[i] = ok ? p.a + (m0 + row) * p.lda + q * 4 : zero;
        a_inc[i] = ok ? BK : 0;
    }
    const float* b_cur[PB]; int b_inc[PB], b_k[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        if constexpr (BT) {
            const int piece = wave + 4 * j;
            const int row = piece * RPP + lane / QR;                   // column n of the tile
            const int q = (lane % QR) ^ (QR == 8 ? ((row >> 1) & 7) : (row & 15));
            b_k[j] = q * 4;
            const bool ok = n0 + row < p.N;
            b_cur[j] = ok ? p.b + (int64_t)(n0 + row) * p.ldb + q * 4 : zero;
            b_inc[j] = ok ? BK : 0;
        } else {
            const int P = (wave + 4 * j) * 64 + lane;
            const int k = P / NQB, nq = P - k * NQB;
            b_k[j] = k;
            const bool ok = n0 + nq * 4 < p.N;
            b_cur[j] = ok ? p.b + (int64_t)k * p.ldb + n0 + nq * 4 : zero;
            b_inc[j] = ok ? (int)p.ldb * BK : 0;
        }
    }
    // piece idx of this wave's PW pieces (A pieces first) of the NEXT step to issue, into ring slot `stage`; CHECK: the step may be ragged
    auto issue_piece = [&](auto idx_tag, auto check_tag, int step, int stage) {
        constexpr int idx = decltype(idx_tag)::value;
        constexpr bool CHECK = decltype(check_tag)::value;
        const uint32_t la = lds0 + (uint32_t)(stage * STAGE_FLOATS * 4);
        if constexpr (idx < PA) {
            const float* src = a_cur[idx];
            if constexpr (CHECK) { if (step * BK + a_k[idx] >= K) src = zero; }
            gemm_glds16(src, la + (uint32_t)((wave + 4 * idx) * 1024));
            a_cur[idx] += a_inc[idx];
        } else if constexpr (idx < PW) {
            constexpr int j = idx - PA;
            const float* src = b_cur[j];
            if constexpr (CHECK) { if (step * BK + b_k[j] >= K) src = zero; }
            gemm_glds16(src, la + (uint32_t)(A_FLOATS * 4 + (wave + 4 * j) * 1024));
            b_cur[j] += b_inc[j];
        }
    };
    // what goes out behind MFMA group `slot` (0 .. NSLOT-1) of a step: pieces slot and slot + NSLOT
    auto issue_slot = [&](auto slot_tag, auto check_tag, int step, int stage) {
        constexpr int sl = decltype(slot_tag)::value;
        issue_piece(std::integral_constant<int, sl>{}, check_tag, step, stage);
        issue_piece(std::integral_constant<int, sl + NSLOT>{}, check_tag, step, stage);
    };
    auto issue_all = [&](int step, int stage) {          // (prologue) the whole step, checked
        auto go = [&](auto self, auto idx_tag) {
            constexpr int idx = decltype(idx_tag)::value;
            if constexpr (idx < PW) {
                issue_piece(idx_tag, std::true_type{}, step, stage);
                self(self, std::integral_constant<int, idx + 1>{});
            }
        };
        go(go, std::integral_constant<int, 0>{});
    };

    // ---- fragments of one K chunk of 8: a[tm] = four k of this lane's row, b[j] = TN consecutive columns of k = 8c + 4h + j
    struct Frags { f32x4_t a[TM]; float b[4][TN]; };
    int a_row_slot[TM], a_swz[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int r = (wm * TM + t) * 32 + l31;
        a_row_slot[t] = r * QR;
        a_swz[t] = QR == 8 ? ((r >> 1) & 7) : (r & 15);
    }
    int bt_row_slot[TN], bt_swz[TN];                       // (BT) the B tile's rows are columns n = (wn * TN + t) * 32 + l31
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int r = (wn * TN + t) * 32 + l31;
        bt_row_slot[t] = r * QR;
        bt_swz[t] = QR == 8 ? ((r >> 1) & 7) : (r & 15);
    }
    const int b_col = BT ? wn * (TN * 32) + l31 : wn * (TN * 32) + TN * l31;
    auto read_frags = [&](int stage, int c, Frags& F) {
        const f32x4_t* As4 = reinterpret_cast<const f32x4_t*>(smem + stage * STAGE_FLOATS);
        const float* Bs = smem + stage * STAGE_FLOATS + A_FLOATS;
#pragma unroll
        for (int t = 0; t < TM; ++t) F.a[t] = As4[a_row_slot[t] + ((2 * c + h) ^ a_swz[t])];
        if constexpr (BT) {
            const f32x4_t* Bs4 = reinterpret_cast<const f32x4_t*>(Bs);
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const f32x4_t v = Bs4[bt_row_slot[t] + ((2 * c + h) ^ bt_swz[t])];
                F.b[0][t] = v.x; F.b[1][t] = v.y; F.b[2][t] = v.z; F.b[3][t] = v.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* src = Bs + (8 * c + 4 * h + j) * BN + b_col;
                if constexpr (TN == 4) {
                    const f32x4_t v = *reinterpret_cast<const f32x4_t*>(src);
                    F.b[j][0] = v.x; F.b[j][1] = v.y; F.b[j][2] = v.z; F.b[j][3] = v.w;
                } else if constexpr (TN == 2) {
                    const f32x2_t v = *reinterpret_cast<const f32x2_t*>(src);
                    F.b[j][0] = v.x; F.b[j][1] = v.y;
                } else {
                    F.b[j][0] = *src;
                }
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    auto mma_group = [&](const Frags& F, int j) {          // the TM x TN MFMAs of one k pair {8c + j, 8c + 4 + j}
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(F.a[tm][j], F.b[j][tn], acc[tm][tn], 0, 0, 0);
    };

    // One K step = NCH chunks of four MFMA groups.  The fragments of chunk c+2 are read (into the registers chunk c just used) behind
    // chunk c's MFMAs, so every chunk starts on fragments read a whole chunk earlier; behind the groups of chunks 0 .. NCH-2 the DMA
    // of step k+2 goes out piece by piece into the ring slot the barrier of step k-1 freed; behind chunk NCH-2 the wave waits for its
    // own pieces of step k+1 (behind them it issued exactly step k+2's PW pieces) and meets the others at the barrier, then reads the
    // first fragments of step k+1 while chunk NCH-1 multiplies.
    // DM: 0 no DMA (the last two steps), 1 unchecked (the issued step is a full one), 2 checked
    auto kstep = [&](auto dm_tag, auto last_tag, int k, int stage, Frags& F0, Frags& F1) {
        constexpr int DM = decltype(dm_tag)::value;
        constexpr bool LAST = decltype(last_tag)::value;           // no step behind this one: no wait, no barrier, no reads ahead
        const int sn = stage + 1 == NSTAGE ? 0 : stage + 1;
        const int sd = stage == 0 ? NSTAGE - 1 : stage - 1;        // ring slot of step k+2 (= that of step k-1)
        auto chunk = [&](auto c_tag) {
            constexpr int c = decltype(c_tag)::value;
            Frags& F = (c & 1) ? F1 : F0;
            auto group = [&](auto j_tag) {
                constexpr int j = decltype(j_tag)::value;
                mma_group(F, j);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (DM != 0 && c < NCH - 1)
                    issue_slot(std::integral_constant<int, c * 4 + j>{}, std::integral_constant<bool, DM == 2>{}, k + 2, sd);
                __builtin_amdgcn_sched_barrier(0);
            };
            group(std::integral_constant<int, 0>{}); group(std::integral_constant<int, 1>{});
            group(std::integral_constant<int, 2>{}); group(std::integral_constant<int, 3>{});
            if constexpr (c + 2 < NCH) {
                read_frags(stage, c + 2, F);
            } else if constexpr (!LAST) {
                if constexpr (c == NCH - 2) {
                    if constexpr (DM == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PW) : "memory");
                    else                   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                read_frags(sn, c + 2 - NCH, F);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto run = [&](auto self, auto c_tag) {
            constexpr int c = decltype(c_tag)::value;
            if constexpr (c < NCH) { chunk(c_tag); self(self, std::integral_constant<int, c + 1>{}); }
        };
        run(run, std::integral_constant<int, 0>{});
    };

    Frags F0, F1;
    if (nsteps > 0) issue_all(0, 0);
    if (nsteps > 1) issue_all(1, 1);
    if (nsteps > 0) {
        if (nsteps > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PW) : "memory");
        else            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        read_frags(0, 0, F0);
        read_frags(0, 1, F1);
    }
    using D0 = std::integral_constant<int, 0>; using D1 = std::integral_constant<int, 1>; using D2 = std::integral_constant<int, 2>;
    int k = 0;
    for (; k + 2 < nfull; ++k) kstep(D1{}, std::false_type{}, k, k % NSTAGE, F0, F1);            // step k+2 is a full step
    for (; k + 2 < nsteps; ++k) kstep(D2{}, std::false_type{}, k, k % NSTAGE, F0, F1);           // (at most one: the ragged last step)
    for (; k + 1 < nsteps; ++k) kstep(D0{}, std::false_type{}, k, k % NSTAGE, F0, F1);
    if (k < nsteps) kstep(D0{}, std::true_type{}, k, k % NSTAGE, F0, F1);

    if constexpr (BT) {
        // ---- epilogue, transposed-B form: lane (l31, h) of column tile tn holds rows (r & 3) + 8 (r >> 2) + 4 h of column 32 tn + l31
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = n0 + b_col + tn * 32;
            if (col >= p.N) continue;
            const float bv = p.bias ? p.bias[col] : 0.0f;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t m = m0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (m < p.M) {
                        float* dst = p.c + m * p.ldc + col;
                        float o = acc[tm][tn][r] + bv;
                        if (p.accumulate) o += *dst;
                        *dst = o;
                    }
                }
        }
        return;
    }
    // ---- epilogue: lane (l31, h) of column tile tn holds rows (r & 3) + 8 (r >> 2) + 4 h of column TN * l31 + tn: TN consecutive floats per row
    const int col = n0 + b_col;
    if (col < p.N) {
        float bv[TN];
#pragma unroll
        for (int t = 0; t < TN; ++t) bv[t] = p.bias ? p.bias[col + t] : 0.0f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < p.M) {
                    float* dst = p.c + m * p.ldc + col;
                    float o[TN];
#pragma unroll
                    for (int t = 0; t < TN; ++t) o[t] = acc[tm][t][r] + bv[t];
                    if constexpr (TN == 4) {
                        if (p.accumulate) { const f32x4_t v = *reinterpret_cast<const f32x4_t*>(dst); o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w; }
                        *reinterpret_cast<f32x4_t*>(dst) = f32x4_t{o[0], o[1], o[2], o[3]};
                    } else if constexpr (TN == 2) {
                        if (p.accumulate) { const f32x2_t v = *reinterpret_cast<const f32x2_t*>(dst); o[0] += v.x; o[1] += v.y; }
                        *reinterpret_cast<f32x2_t*>(dst) = f32x2_t{o[0], o[1]};
                    } else {
                        if (p.accumulate) o[0] += *dst;
                        *dst = o[0];
                    }
                }
            }
    }
}

// Tile forms: 0 = 128 x 256 (wave 64 x 128), 1 = 256 x 128 (wave 128 x 64), 2 = 128 x 128 (wave 64 x 64), 3 = 64 x 128 (wave 32 x 64),
// 4 = 128 x 64 (wave 64 x 32), 5 = 64 x 64 (wave 32 x 32)
static const int kGemmBM[6] = {128, 256, 128, 64, 128, 64}, kGemmBN[6] = {256, 128, 128, 128, 64, 64};

int gemm_pw_tile_rows(int form) { return form >= 0 && form < 6 ? kGemmBM[form] : 0; }
int gemm_pw_tile_cols(int form) { return form >= 0 && form < 6 ? kGemmBN[form] : 0; }

int launch_gemm_pw(const float* a, int64_t lda, const float* b, int64_t ldb, const float* bias, float* c, int64_t ldc, int64_t M, int N, int K,
                   int accumulate, int form, int xcd_remap, hipStream_t st, int b_transposed)
{
    if (form < 0 || form >= 6) return fail(PP_ERR_BAD_ARG, "gemm_pw: tile form %d", form);
    GemmParams p{a, b, bias, c, lda, ldb, ldc, M, N, K, 0, xcd_remap, accumulate};
    const int bm = kGemmBM[form], bn = kGemmBN[form];
    p.n_tiles = (N + bn - 1) / bn;
    const dim3 grid((unsigned)(((M + bm - 1) / bm) * p.n_tiles)), block(256);
    if (b_transposed) {
        switch (form) {
            case 0: hipLaunchKernelGGL((gemm_pw_kernel<2, 4, 2, 2, 32, true>), grid, block, 0, st, p); break;
            case 1: hipLaunchKernelGGL((gemm_pw_kernel<4, 2, 2, 2, 32, true>), grid, block, 0, st, p); break;
            case 2: hipLaunchKernelGGL((gemm_pw_kernel<2, 2, 2, 2, 32, true>), grid, block, 0, st, p); break;
            case 3: hipLaunchKernelGGL((gemm_pw_kernel<1, 2, 2, 2, 64, true>), grid, block, 0, st, p); break;
            case 4: hipLaunchKernelGGL((gemm_pw_kernel<2, 1, 2, 2, 64, true>), grid, block, 0, st, p); break;
            default: hipLaunchKernelGGL((gemm_pw_kernel<1, 1, 2, 2, 64, true>), grid, block, 0, st, p); break;
        }
    } else {
        switch (form) {
            case 0: hipLaunchKernelGGL((gemm_pw_kernel<2, 4, 2, 2, 32>), grid, block, 0, st, p); break;
            case 1: hipLaunchKernelGGL((gemm_pw_kernel<4, 2, 2, 2, 32>), grid, block, 0, st, p); break;
            case 2: hipLaunchKernelGGL((gemm_pw_kernel<2, 2, 2, 2, 32>), grid, block, 0, st, p); break;
            case 3: hipLaunchKernelGGL((gemm_pw_kernel<1, 2, 2, 2, 64>), grid, block, 0, st, p); break;
            case 4: hipLaunchKernelGGL((gemm_pw_kernel<2, 1, 2, 2, 64>), grid, block, 0, st, p); break;
            default: hipLaunchKernelGGL((gemm_pw_kernel<1, 1, 2, 2, 64>), grid, block, 0, st, p); break;
        }
    }
    return hipGetLastError() == hipSuccess ? PP_OK : fail(PP_ERR_LAUNCH, "gemm_pw_kernel launch failed");
}

// element offsets are 32-bit, loads are 16-byte
int gemm_pw_supported(const float* a, int64_t lda, const float* b, int64_t ldb, const float* c, int64_t ldc, int64_t M, int N, int K)
{
    return lda % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 && N % 4 == 0 && K % 4 == 0 && ((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0 &&
           ((uintptr_t)c & 15) == 0;
}

}  // namespace pp
