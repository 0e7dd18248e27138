// ckks_ntt_one.h — a whole forward transform per block for logN 13 .. 15: ONE launch, one trip through HBM.
//
// A limb of 2^15 words is 256 KiB: more than the LDS of a CU (160 KiB), exactly the register file of 1024 threads at
// 64 VGPRs each.  So the limb lives in REGISTERS — T = N / 32 threads, 32 words per thread — and goes through three
// radix-32 register steps (logN - 10, 5 and 5 stages); between them the words change owner:
//   step A  distances N/2 .. 1024   element e of thread t = word (e >> SUB) * 1024 + t * 2^SUB + (e & (2^SUB - 1)),
//                                   SUB = 15 - logN: a lane loads 2^SUB adjacent words per 1024-word slice; the twiddles of
//                                   these stages are the same for every thread of the limb (scalar loads);
//   ---- exchange 1: block-wide, through LDS in natural word order, HALF the limb at a time (N / 2 words = 128 KiB at
//        logN 15).  Every thread writes the 16 of its elements that lie in the half; the waves whose step-B words lie in
//        it read all 32 of theirs (whole waves: the half is the top bit of the thread index).  3 + 1 barriers;
//   step B  distances 512 .. 32     thread t = hi * 32 + l holds words hi * 1024 + e * 32 + l
//   ---- exchange 2: the 32 x 32 transpose inside each half-wave (thread (hi, l), element e  ->  thread (hi, e), element l):
//        wave-private LDS, no block barrier, again in two halves of 16 elements, rows padded to 33 words;
//   step C  distances 16 .. 1       thread t holds the 32 consecutive words 32 t + e
//   ---- store: through the wave's own LDS span so that a store instruction writes four full 128-byte lines.
// Relaxed arithmetic only (the engine's internal transforms: ArithDpR on fp64-class limbs, Shoup products on the 60-bit
// ones); the source of the words is a template hook — here the key switch's digit extension (ks_ext1_body in ckks_ks.hip).
// The two-pass form (column kernel + 4096-word tiled pass) moves 2 x 16 N bytes per limb and needs two launches; it
// remains the choice where a launch has too few limbs to give every CU a block (see ks_forward).
#pragma once
#include "ckks_ntt_core.h"

namespace {

template <int LOGN>
struct OneGeom {
    static_assert(LOGN >= 13 && LOGN <= 15, "one-launch transform: 32 words per thread, 256 .. 1024 threads");
    static constexpr int N = 1 << LOGN;
    static constexpr int T = N / 32;         // threads per block
    static constexpr int A = LOGN - 10;      // stages of the first step
    static constexpr int SUB = 5 - A;        // log2 of adjacent words per lane and 1024-word slice in the first step
    static constexpr int WAVES = T / 64;
    static constexpr int X2 = 2 * 16 * 33;   // words per wave of the half-wave transposes (2 half-waves x 16 elements x 33)
    static constexpr int ST = 64 * 17;       // words per wave of the store transposition (64 lanes x (16 + 1 pad))
    static constexpr int LDS_WORDS = (N / 2) > (WAVES * ST) ? (N / 2) : (WAVES * ST);
};

// twiddle entries fetched per group of butterflies: 8 doubles / 4 Shoup pairs (16 VGPRs either way)
template <class A> struct TwChunk { static constexpr int n = 8; };
template <> struct TwChunk<ArithShoup> { static constexpr int n = 4; };

// forward stages 0 .. S-1 of a radix-32 register step: stage u pairs the elements e and e + (16 >> u) of every group of
// 32 >> u elements; group j of stage u takes the table entry (i0 << u) + j
template <class A, int S>
__device__ __forceinline__ void fwd_regs32(typename A::T (&x)[32], int i0, const Ctx &c) {
    constexpr int CH = TwChunk<A>::n;
#pragma unroll
    for (int u = 0; u < S; ++u) {
        const int du = 16 >> u, groups = 1 << u;
#pragma unroll
        for (int j0 = 0; j0 < groups; j0 += CH) {
            typename A::W wv[CH];
            A::tw_group(c, (i0 << u) + j0, groups < CH ? groups : CH, wv);
#pragma unroll
            for (int jj = 0; jj < CH; ++jj) {
                if (j0 + jj < groups) {
                    const int e0 = (j0 + jj) << (5 - u);
#pragma unroll
                    for (int t = 0; t < du; ++t) A::fwd(c, x[e0 + t], x[e0 + t + du], wv[jj], 0);
                }
            }
        }
    }
}

// exchange 1 (see the header): x = step-A order, on return step-B order (thread hi * 32 + l: words hi * 1024 + e * 32 + l)
template <class TT, int LOGN>
__device__ __forceinline__ void one_exchange1(TT *sm, TT (&x)[32], int t) {
    typedef OneGeom<LOGN> G;
    const int my_half = t >> (LOGN - 6);     // top bit of the thread index = top bit of its step-B words
    const int rd = (t >> 5) * 1024 + (t & 31) - my_half * (G::N / 2);
    TT y[32];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (r) lds_barrier();                // the readers of the first half are done
#pragma unroll
        for (int ee = 0; ee < 16; ++ee) {
            const int e = 16 * r + ee;       // elements whose slice hi = e >> SUB has top bit r
            sm[(e >> G::SUB) * 1024 + (t << G::SUB) + (e & ((1 << G::SUB) - 1)) - r * (G::N / 2)] = x[e];
        }
        lds_barrier();
        if (my_half == r) {
#pragma unroll
            for (int e = 0; e < 32; ++e) y[e] = sm[rd + e * 32];
        }
    }
#pragma unroll
    for (int e = 0; e < 32; ++e) x[e] = y[e];
    lds_barrier();                           // LDS is rewritten wave by wave from here on
}

// exchange 2: transpose of the 32 x 32 words a half-wave holds (lane l, element e -> lane e, element l); wave-private
template <class TT, int LOGN>
__device__ __forceinline__ void one_exchange2(TT *sm, TT (&x)[32], int t) {
    typedef OneGeom<LOGN> G;
    TT *ws = sm + (t >> 6) * G::X2 + ((t >> 5) & 1) * (16 * 33);
    const int l = t & 31;
    TT y[32];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int ee = 0; ee < 16; ++ee) ws[ee * 33 + l] = x[16 * r + ee];
        wave_lds_sync();
        if ((l >> 4) == r) {
            const TT *row = ws + (l & 15) * 33;
#pragma unroll
            for (int k = 0; k < 32; ++k) y[k] = row[k];
        }
        wave_lds_sync();                     // (a wave's LDS operations execute in order: the next writes follow these reads)
    }
#pragma unroll
    for (int e = 0; e < 32; ++e) x[e] = y[e];
}

// store: thread t holds the words 32 t + e of `row`; 8 bytes per lane, four full 128-byte lines per instruction
template <int LOGN>
__device__ __forceinline__ void one_store(i64 *sm, const i64 (&o)[32], i64 *row, int t) {
    typedef OneGeom<LOGN> G;
    i64 *ws = sm + (t >> 6) * G::ST;
    const int lane = t & 63;
    i64 *span = row + ((t >> 6) << 11);      // the wave's 2048 consecutive words
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int ee = 0; ee < 16; ++ee) ws[lane * 17 + ee] = o[16 * r + ee];
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int m = k * 64 + lane;     // index inside the half: lane m >> 4, element m & 15
            span[(m >> 4) * 32 + 16 * r + (m & 15)] = ws[(m >> 4) * 17 + (m & 15)];
        }
        wave_lds_sync();
    }
}

// steps A .. C on words already in step-A order; on return x[e] = transformed word 32 t + e (lazy / balanced, not reduced)
template <class A, int LOGN>
__device__ __forceinline__ void one_fwd_steps(typename A::T *sm, typename A::T (&x)[32], int t, const Ctx &c) {
    typedef OneGeom<LOGN> G;
    fwd_regs32<A, G::A>(x, 1, c);
    one_exchange1<typename A::T, LOGN>(sm, x, t);
    fwd_regs32<A, 5>(x, (1 << G::A) + (t >> 5), c);
    one_exchange2<typename A::T, LOGN>(sm, x, t);
    fwd_regs32<A, 5>(x, G::T + t, c);
}

}  // namespace
