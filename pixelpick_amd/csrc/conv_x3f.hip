// conv_x3f.hip — the bf16x3 convolution (conv_igemm.hip: conv_x3_kernel, six bf16 MFMAs per fp32 product on operands split exactly into
// three bf16 planes) with the ACTIVATION operand split INSIDE the kernel.
//
// Replaces, for the layers it takes, the pair x3_split_kernel + conv_x3_kernel behind the dense nn.Conv2d calls of the reference's
// ResNet50 models and heads: networks/backbones/resnet_models.py:58-94 (Bottleneck 1x1 / 3x3), networks/decoders.py:25-77,107-114,
// networks/aspp.py:49-58, and their backward-data (model.py:121).
//
// Why: the split launch reads the fp32 activations once and writes 6 bytes per element of planes, which the convolution then reads
// back - 40 us for the 2048-channel Bottleneck input, more than the convolution gains on every layer below ~16 GFLOP (round 3-5
// records: profiles/r04_graded_1x1.txt "the planes would have to come from the producer").  Here a lane loads the 8 fp32 channels
// of its (row, half) slot straight from the NHWC tensor two K steps ahead (global_load_dwordx4 x 2, in registers), splits them with
// v_cvt_pk_bf16_f32 (round to nearest even: the planes are bit-identical to x3_split_kernel's) behind the MFMA groups of a later step
// and writes the three 16-byte plane pieces into the LDS slots the DMA of conv_x3_kernel would have filled.  ~45 VALU operations per
// lane and K step against 24 MFMAs of 32 cycles.  The weights keep coming as pre-split planes over the LDS-DMA path (they are split
// once per optimiser step on the second queue, engine._prefetch_weight_planes).  Results are bit-identical to conv_x3_kernel.
//
// Pipeline (one barrier per K step, ring of NSTAGE stages, PRE = NSTAGE):
//   step k, top   : wait for my B pieces of step k+1, barrier
//   group 0, 1    : MFMAs of plane pairs; convert A registers of step k+PRE (loaded in step k-2) -> ds_write into ring slot k % NSTAGE
//   group 2       : issue the A loads of step k+PRE+2 into the (accumulation) registers just freed
//   group 3       : B planes of step k+PRE by LDS-DMA (waves 0 .. BN/32-1)
//   groups 0-5    : fragment reads of step k+1 behind each group
//
// STATUS (round 6, measured: profiles/r06_x3f_in_kernel_split.txt): correct - bit-identical to conv_x3_kernel on every shape both can run
// (tests/test_conv_x3f_gpu.py) - and SLOWER than what it was meant to replace: 10-25 % behind conv_x3_kernel with the planes held and
// behind the fp32-MFMA kernels on the mid-size layers, whatever the form of the waits (three forms measured).  The bf16x3 kernels run at
// the chip's power limit (profiles/r05_conv_x3_power.txt); the ~80 VALU operations per lane and step of the split lower the clock under
// the MFMA stream by about what they cost.  It is therefore compiled into the TEST BUILD only (-DPP_DEBUG_KNOBS, off unless
// pp_debug_set_x3f(1)); the product library does not contain it.
#ifdef PP_DEBUG_KNOBS
#include <type_traits>

#include "conv_types.h"

namespace pp {

typedef float f32x4 __attribute__((ext_vector_type(4)));     // (a native vector: HIP's float4 is a struct, which inline asm cannot tie to registers)

__device__ __attribute__((aligned(16))) float g_x3f_zero[4] = {0.f, 0.f, 0.f, 0.f};

// the three planes of one B piece behind ONE M0 save / restore (as conv_igemm.hip: x3_glds16x3)
__device__ __forceinline__ void x3f_glds16x3(uint32_t voff, const void* s0, const void* s1, const void* s2, uint32_t lds_dst_in, uint32_t stride_in)
{
    const uint32_t d0 = __builtin_amdgcn_readfirstlane(lds_dst_in);
    const uint32_t st = __builtin_amdgcn_readfirstlane(stride_in);
    auto uni = [](const void* q) -> const void* {
        const uint64_t v = reinterpret_cast<uint64_t>(q);
        return reinterpret_cast<const void*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) |
                                             (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v));
    };
    const void* b0 = uni(s0); const void* b1 = uni(s1); const void* b2 = uni(s2);
    unsigned keep, t;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
                 "s_add_u32 %1, %6, %7\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4\n\t"
                 "s_add_u32 %1, %1, %7\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep), "=&s"(t) : "v"(voff), "s"(b0), "s"(b1), "s"(b2), "s"(d0), "s"(st) : "memory", "scc");
}

template <int BM, int BN>
__global__ __launch_bounds__(BM * 2, 2) void conv_x3f_kernel(ConvParams p, X3Operands o)
{
    constexpr int TM = 2, TN = BN / 64, WN = 2, NWAVE = BM / 32;
    constexpr int NSTAGE = (BM == 256 && BN == 128) ? 4 : (BM == 256 ? 5 : 3);
    constexpr int PRE = NSTAGE;
    constexpr int A_BYTES = BM * 32, B_BYTES = BN * 32;                    // one plane of one stage
    constexpr int STAGE_BYTES = 3 * (A_BYTES + B_BYTES);
    constexpr int NBW = BN / 32;                                          // waves that DMA B rows (32 rows each)
    static_assert(NBW <= NWAVE, "B rows are loaded by the first BN/32 waves");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NSTAGE * STAGE_BYTES];
    struct Frags { bf16x8_t a[3][TM], b[3][TN]; };

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
    const int n0 = o.col_base + nt * BN;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);

    // ---- per-lane addressing: lane -> (row = lane >> 1 of this wave's 32 rows, 16-byte half swizzled by bit 3 of the row)
    const int lr = lane >> 1;
    const int lhalf = (lane & 1) ^ ((lr >> 3) & 1);
    uint32_t a_f0 = 0;                                   // BYTE offset of (pixel row, channel lhalf * 8) in the fp32 NHWC tensor, tap (0,0), chunk 0
    unsigned a_vm = 0u;
    {
        const int64_t m = m0 + wave * 32 + lr;
        if (m < p.M) {
            const unsigned mu = (unsigned)m;
            const unsigned t = mu / (unsigned)p.Wo;
            const int ow = (int)(mu - t * (unsigned)p.Wo);
            const unsigned bb = t / (unsigned)p.Ho;
            const int oh = (int)(t - bb * (unsigned)p.Ho);
            const int ih0 = oh * p.stride, iw0 = ow * p.stride;
            a_f0 = ((uint32_t)(((int)bb * p.H + ih0) * p.W + iw0) * (uint32_t)p.ldx + (uint32_t)(lhalf * 8)) * 4u;
            for (int t2 = 0; t2 < p.taps.n; ++t2) {
                const int ih = ih0 + p.taps.dh[t2], iw = iw0 + p.taps.dw[t2];
                a_vm |= ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) ? (1u << t2) : 0u;
            }
        }
    }
    const bool loads_b = wave < NBW;
    const int b_row = n0 + wave * 32 + lr;
    const bool b_ok = loads_b && b_row < p.Cn;
    const uint32_t b_e0 = (uint32_t)(b_row * 16 + lhalf * 8) * 2u;
    const uint32_t b_zero = o.b_zero + (uint32_t)(lhalf * 16);
    const uint16_t* b0 = o.b; const uint16_t* b1 = o.b + o.b_plane; const uint16_t* b2 = o.b + 2 * o.b_plane;
    const char* xbase = reinterpret_cast<const char*>(p.x);
    const char* zero = reinterpret_cast<const char*>(g_x3f_zero);
    asm volatile("" : "+v"(zero));

    __shared__ int s_tap[32][2];
    if (tid < p.taps.n) {
        s_tap[tid][0] = (p.taps.dh[tid] * p.W + p.taps.dw[tid]) * (int)p.ldx * 4;    // byte shift of the A row in the fp32 tensor
        s_tap[tid][1] = p.taps.widx[tid] * o.n_rows * 32;                            // byte offset of the tap's B rows
    }
    __syncthreads();

    // (wave-uniform values the compiler keeps in VGPRs once they meet a vector compare - the tap count comes out of the 1.3-KiB argument
    // block through a vector load: readfirstlane puts the loop control and the step addressing on the scalar unit)
    const int ntaps = __builtin_amdgcn_readfirstlane(p.taps.n);
    const int nchunk = __builtin_amdgcn_readfirstlane(o.Kp / 16);
    const int n = __builtin_amdgcn_readfirstlane(ntaps * nchunk);      // K steps (no split-K on this path)
    const int Ck = __builtin_amdgcn_readfirstlane(p.Ck);
    int ib_ch = 0, ib_ti = 0;                            // (chunk, tap) of the next B step to issue: chunk outer, tap inner
    int ia_ch = 0, ia_ti = 0;                            // ... of the next A step to LOAD
    // the tap-table entries of those steps, read from LDS one step BEFORE they are needed (a read in front of its use would sit behind
    // the step's fragment reads and drain them: lgkmcnt(0))
    int ta_next = s_tap[0][0], tb_next = s_tap[0][1];
    const int chan0 = lhalf * 8;

    // ---- A operand: registers -> planes -> LDS ----
    // The loads of the steady state land in ACCUMULATION registers a[0:7] (even steps) / a[8:15] (odd steps) that only the inline asm
    // below names: the compiler keeps its MFMA accumulators in VGPRs here and allocates no AGPR of its own, so nothing it generates
    // (copies at control-flow joins, spills) can touch a load in flight, and - the loads being invisible to it - it adds no waits of
    // its own: the counted s_waitcnt in a_take is the only one.  (Two earlier forms: inline-asm loads into VGPRs - the compiler copied
    // the destination registers at the loop's joins before the data had arrived: garbage; plain C++ loads - correct, but its waits
    // count only the loads it knows and so drained the B pieces of the previous step at every step: slower than the fp32 kernels.)
    auto a_addr = [&](const char*& q0, const char*& q1) {  // addresses of the next A step (8 channels of this lane's row); advances (ia_ti, ia_ch)
        const bool okr = ((a_vm >> ia_ti) & 1u) != 0u;
        const int c = ia_ch * 16 + chan0;
        const uint32_t off = a_f0 + (uint32_t)(__builtin_amdgcn_readfirstlane(ta_next) + ia_ch * 64);
        q0 = (okr && c < Ck) ? xbase + off : zero;
        q1 = (okr && c + 4 < Ck) ? xbase + off + 16 : zero;
        if (++ia_ti == ntaps) { ia_ti = 0; ++ia_ch; }
        ta_next = s_tap[ia_ti][0];
    };
    auto a_load_acc = [&](auto slot_tag) {                // (steady state and tail) into the slot's AGPRs
        const char* q0; const char* q1;
        a_addr(q0, q1);
        if constexpr (decltype(slot_tag)::value == 0)
            asm volatile("global_load_dwordx4 a[0:3], %0, off\n\tglobal_load_dwordx4 a[4:7], %1, off" :: "v"(q0), "v"(q1)
                         : "memory", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
        else
            asm volatile("global_load_dwordx4 a[8:11], %0, off\n\tglobal_load_dwordx4 a[12:15], %1, off" :: "v"(q0), "v"(q1)
                         : "memory", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
    };
    // wait (vmcnt(N): the caller's count of operations issued behind the slot's loads) and move the slot into ordinary registers
#define PP_X3F_TAKE(N, A0, A1, A2, A3, A4, A5, A6, A7)                                                                                      \
    asm volatile("s_waitcnt vmcnt(" #N ")\n\tv_accvgpr_read_b32 %0, " #A0 "\n\tv_accvgpr_read_b32 %1, " #A1 "\n\tv_accvgpr_read_b32 %2, " #A2   \
                 "\n\tv_accvgpr_read_b32 %3, " #A3 "\n\tv_accvgpr_read_b32 %4, " #A4 "\n\tv_accvgpr_read_b32 %5, " #A5                         \
                 "\n\tv_accvgpr_read_b32 %6, " #A6 "\n\tv_accvgpr_read_b32 %7, " #A7                                                           \
                 : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3), "=v"(f4), "=v"(f5), "=v"(f6), "=v"(f7) :: "memory")
    auto a_take = [&](auto slot_tag, auto steady_tag, f32x4 (&r)[2]) {
        float f0, f1, f2, f3, f4, f5, f6, f7;
        constexpr bool ST = decltype(steady_tag)::value;
        if constexpr (decltype(slot_tag)::value == 0) {
            if (!ST)          PP_X3F_TAKE(0, a0, a1, a2, a3, a4, a5, a6, a7);
            else if (loads_b) PP_X3F_TAKE(8, a0, a1, a2, a3, a4, a5, a6, a7);
            else              PP_X3F_TAKE(2, a0, a1, a2, a3, a4, a5, a6, a7);
        } else {
            if (!ST)          PP_X3F_TAKE(0, a8, a9, a10, a11, a12, a13, a14, a15);
            else if (loads_b) PP_X3F_TAKE(8, a8, a9, a10, a11, a12, a13, a14, a15);
            else              PP_X3F_TAKE(2, a8, a9, a10, a11, a12, a13, a14, a15);
        }
        r[0] = f32x4{f0, f1, f2, f3};
        r[1] = f32x4{f4, f5, f6, f7};
    };
#undef PP_X3F_TAKE
    auto a_load = [&](f32x4 (&r)[2]) {                    // (prologue) plain loads the compiler tracks and waits for itself
        const char* q0; const char* q1;
        a_addr(q0, q1);
        typedef const f32x4 __attribute__((address_space(1))) * gp4;     // (a generic pointer would make these flat loads)
        r[0] = *(gp4)(q0);
        r[1] = *(gp4)(q1);
    };
    struct Planes { uint32_t hi[4], mid[4], lo[4]; };
    auto split4 = [&](const f32x4& v, uint32_t* hi, uint32_t* mid, uint32_t* lo) {      // four fp32 -> two packed words per plane
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bf2 hh, mm, ll;
            float r1[2], r2[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                hh[e] = (__bf16)f[2 * j + e];
                r1[e] = f[2 * j + e] - (float)hh[e];
                mm[e] = (__bf16)r1[e];
                r2[e] = r1[e] - (float)mm[e];
                ll[e] = (__bf16)r2[e];
            }
            hi[j] = __builtin_bit_cast(uint32_t, hh);
            mid[j] = __builtin_bit_cast(uint32_t, mm);
            lo[j] = __builtin_bit_cast(uint32_t, ll);
        }
    };
    auto a_store = [&](const Planes& P, int stage) {
        uint4* S = reinterpret_cast<uint4*>(smem + stage * STAGE_BYTES + wave * 1024) + lane;
        S[0] = make_uint4(P.hi[0], P.hi[1], P.hi[2], P.hi[3]);
        S[A_BYTES / 16] = make_uint4(P.mid[0], P.mid[1], P.mid[2], P.mid[3]);
        S[2 * (A_BYTES / 16)] = make_uint4(P.lo[0], P.lo[1], P.lo[2], P.lo[3]);
    };

    // ---- B operand: LDS-DMA of the pre-split weight planes ----
    auto b_issue = [&](int stage) {
        if (loads_b) {
            const int tb = __builtin_amdgcn_readfirstlane(tb_next);
            const uint32_t vb = b_ok ? b_e0 + (uint32_t)tb + (uint32_t)ib_ch * o.b_chunk : b_zero;
            x3f_glds16x3(vb, b0, b1, b2, lds0 + (uint32_t)(stage * STAGE_BYTES) + (uint32_t)(3 * A_BYTES + wave * 1024), B_BYTES);
        }
        if (++ib_ti == ntaps) { ib_ti = 0; ++ib_ch; }
        tb_next = s_tap[ib_ti][1];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // fragment slot of (row r, logical half h): 16-byte slot r*2 + (h ^ ((r >> 3) & 1))
    int a_slot[TM], b_slot[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) { const int r = (wm * TM + t) * 32 + l31; a_slot[t] = r * 2 + (h ^ ((r >> 3) & 1)); }
#pragma unroll
    for (int t = 0; t < TN; ++t) { const int r = (wn * TN + t) * 32 + l31; b_slot[t] = r * 2 + (h ^ ((r >> 3) & 1)); }
    auto read_a = [&](int stage, Frags& F, int pl) {
        const bf16x8_t* S = reinterpret_cast<const bf16x8_t*>(smem + stage * STAGE_BYTES);
#pragma unroll
        for (int t = 0; t < TM; ++t) F.a[pl][t] = S[pl * (A_BYTES / 16) + a_slot[t]];
    };
    auto read_b = [&](int stage, Frags& F, int pl) {
        const bf16x8_t* S = reinterpret_cast<const bf16x8_t*>(smem + stage * STAGE_BYTES);
#pragma unroll
        for (int t = 0; t < TN; ++t) F.b[pl][t] = S[3 * (A_BYTES / 16) + pl * (B_BYTES / 16) + b_slot[t]];
    };
    // six product terms, smallest first: (A plane, B plane) - conv_x3_kernel's order, so the sums are bit-identical
    auto mma_term = [&](const Frags& F, int term) {
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.a[TA[term]][tm], F.b[TB[term]][tn], acc[tm][tn], 0, 0, 0);
    };

    // One K step.  STEADY (compile time): the two previous steps issued their full pattern (2 A loads in group 2, 3 B pieces in group 3)
    // and this one does too, so the wait for the B pieces is counted; otherwise it is vmcnt(0) and every issue is checked against n.
    auto kstep = [&](auto steady_tag, auto slot_tag, int k, int stage, Frags& cur, Frags& nxt) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        const bool rd = STEADY || k + 1 < n, put = STEADY || k + PRE < n, ld = STEADY || k + PRE + 2 < n;
        const int sn = stage + 1 == NSTAGE ? 0 : stage + 1;
        if (rd) {
            // my B pieces of step k+1 (issued PRE - 1 steps ago in group 3) have landed; behind them this wave issued 2 + 3 operations
            // per later step, which may still fly
            if (STEADY) { if (loads_b) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(5 * (PRE - 2)) : "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                    // everyone's pieces of step k+1; ring slot `stage` (step k: fragments in registers) is free
            asm volatile("" ::: "memory");
        }
        Planes P;
        f32x4 ar[2];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            mma_term(cur, g);
            __builtin_amdgcn_sched_barrier(0);
            if (rd) { if (g < 3) read_a(sn, nxt, g); else read_b(sn, nxt, g - 3); }
            if (g == 0 && put) {
                // the slot was loaded two steps ago (group 2): behind it this wave issued 3 + 2 + 3 (B waves) / 2 operations
                a_take(slot_tag, steady_tag, ar);
                split4(ar[0], P.hi, P.mid, P.lo);
            }
            if (g == 1 && put) {
                split4(ar[1], P.hi + 2, P.mid + 2, P.lo + 2);
                a_store(P, stage);
            }
            if (g == 2 && ld) a_load_acc(slot_tag);
            if (g == 3 && put) b_issue(stage);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- prologue: steps 0 .. PRE-1 into the ring (B by DMA, A through registers), A registers of steps PRE, PRE+1 in flight
    Frags F0, F1;
#pragma unroll
    for (int s0 = 0; s0 < PRE; ++s0)
        if (s0 < n) b_issue(s0);
    {
        f32x4 tmp[PRE][2];
#pragma unroll
        for (int s0 = 0; s0 < PRE; ++s0)
            if (s0 < n) a_load(tmp[s0]);
#pragma unroll
        for (int s0 = 0; s0 < PRE; ++s0)
            if (s0 < n) {
                Planes P;
                split4(tmp[s0][0], P.hi, P.mid, P.lo);
                split4(tmp[s0][1], P.hi + 2, P.mid + 2, P.lo + 2);
                a_store(P, s0);
            }
    }
    if (n > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                     // (with its lgkmcnt(0): the plane pieces this wave wrote are in LDS)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) { read_a(0, F0, pl); read_b(0, F0, pl); }
    }
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
    if (PRE < n) a_load_acc(S0{});
    if (PRE + 1 < n) a_load_acc(S1{});

    int k = 0;
    // the first pair runs with checked issues and full waits: the counted waits assume two full steps behind them
    if (k < n) { kstep(std::false_type{}, S0{}, k, k % NSTAGE, F0, F1); ++k; }
    if (k < n) { kstep(std::false_type{}, S1{}, k, k % NSTAGE, F1, F0); ++k; }
    for (; k + PRE + 3 < n; k += 2) {
        kstep(std::true_type{}, S0{}, k, k % NSTAGE, F0, F1);
        kstep(std::true_type{}, S1{}, k + 1, (k + 1) % NSTAGE, F1, F0);
    }
    for (; k + 1 < n; k += 2) {
        kstep(std::false_type{}, S0{}, k, k % NSTAGE, F0, F1);
        kstep(std::false_type{}, S1{}, k + 1, (k + 1) % NSTAGE, F1, F0);
    }
    if (k < n) kstep(std::false_type{}, S0{}, k, k % NSTAGE, F0, F1);
    conv_epilogue<TM, TN>(p, acc, m0, n0, wm, wn);
}

int conv_x3f_supported(const ConvParams& p)
{
    // float4 loads from byte offsets that fit 32 bits; tap masks of 32 bits; no input affine, no strided backward-data
    return p.ldx % 4 == 0 && p.Ck % 4 == 0 && (reinterpret_cast<uintptr_t>(p.x) & 15) == 0 && p.taps.n >= 1 && p.taps.n <= 32 &&
           (int64_t)p.B * p.H * p.W * p.ldx * 4 < (1ll << 32) - (1ll << 20) && !p.in_scale && p.bwd_stride <= 1 && p.splits <= 1;
}

int launch_conv_x3f(const ConvParams& p, const X3Operands& o, bool m256, bool n128, unsigned blocks, hipStream_t st)
{
    const dim3 grid(blocks);
    if (m256) {
        if (n128) hipLaunchKernelGGL((conv_x3f_kernel<256, 128>), grid, dim3(512), 0, st, p, o);
        else      hipLaunchKernelGGL((conv_x3f_kernel<256, 64>), grid, dim3(512), 0, st, p, o);
    } else {
        if (n128) hipLaunchKernelGGL((conv_x3f_kernel<128, 128>), grid, dim3(256), 0, st, p, o);
        else      hipLaunchKernelGGL((conv_x3f_kernel<128, 64>), grid, dim3(256), 0, st, p, o);
    }
    return hipGetLastError() == hipSuccess ? PP_OK : fail(PP_ERR_LAUNCH, "conv_x3f_kernel launch failed");
}

}  // namespace pp
#endif  // PP_DEBUG_KNOBS
