// ckks_ntt_tile16.h — the contiguous 4096-word pass of a two-pass transform (logN >= 13) with 16 words per thread.
//
// 256 threads per tile, three radix-16 register steps: two block-wide LDS exchanges instead of the three of the
// 8-words-per-thread form (ckks_ntt_core.h: fwd_tile12 / inv_tile12, still used for logN = 12), the per-step index /
// twiddle-address work amortised over 32 butterflies instead of 12, and the twiddles of the step at distance >= 256
// depend on the tile only (scalar loads).  LDS holds the tile once (34 KiB): four blocks = 4 waves per SIMD, so the
// arithmetic is written for instruction-level parallelism inside a wave:
//   * exact fp64 class: per stage the 8 modular products first (independent dependency chains), ONE wave-uniform
//     test for the rare lazy-fix case (T0 < 2^22, see ArithDp) whose repair sits out of the fast path, then the 8
//     add / sub pairs;
//   * relaxed fp64 class and the integer class have no per-butterfly branch to begin with; the RELAXED integer class
//     runs Shoup products on lazy 64-bit words (ArithShoup: 26 instead of 46 instructions per butterfly).
// Measured on MI355X (tools/proto16_check.py, 25 fp64-class limbs x 128 polynomials, exact): 855 -> 737 us per launch.
// (The same stage-wise guard in the 8-words-per-thread kernel, 6 waves per SIMD, changes nothing: 1 185 vs 1 200 us.)
//
// Word order inside a step: thread w holds words p + (e << LOGDL), e = 0..15, with
//   LOGDL = 8: p = w                       (distances 2048 .. 256; first forward step / last inverse step: global memory)
//   LOGDL = 4: p = (w >> 4) << 8 | w & 15  (distances 128 .. 16)
//   LOGDL = 0: p = 16 w                    (distances 8 .. 1;   last forward step / first inverse step)
// LDS index of word L is L + (L >> 4) (one pad word per 16): the 16-consecutive-word accesses of the LOGDL = 0 step
// (thread stride 17 words = 34 banks) and the strided ones are conflict-free for ds_read/write_b64.
// A tile holding a word outside [0, 2q) (signed-lazy inputs) is detected before anything is stored
// and redone stage by stage in signed integer arithmetic, exactly as the reference would compute it.
#pragma once
#include "ckks_ntt_core.h"

#define NTT16_THREADS 256
#define PAD16(L) ((L) + ((L) >> 4))
#define NTT16_LDS_WORDS (4096 + 256)
#define NTT16_FLAG NTT16_LDS_WORDS   // flag word behind the tile: one byte per wave

namespace {

// ---- register steps ----------------------------------------------------------------------------------------------
// the twiddles of the three widest-shared stages of a step (7 entries), requested ahead of the barrier in front of it:
// forward steps use stages 0..2 (1, 2, 4 entries at i0, 2 i0, 4 i0), inverse steps stages 1..3 (4, 2, 1 entries at
// 4 il, 2 il, il); the remaining 8-entry stage is loaded where it is used
template <class A>
struct Tw16Early {
    typename A::W w1[1], w2[2], w4[4];
    __device__ __forceinline__ void load(const Ctx &c, int i) {
        A::tw_group(c, i, 1, w1);
        A::tw_group(c, i << 1, 2, w2);
        A::tw_group(c, i << 2, 4, w4);
    }
    // group of `cnt` (1, 2 or 4) entries into wv
    __device__ __forceinline__ void get(int cnt, typename A::W (&wv)[8]) const {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < cnt) wv[j] = cnt == 1 ? w1[0] : (cnt == 2 ? w2[j & 1] : w4[j]);
    }
};

// forward: stage u of the step uses the 2^u table entries (i0 << u) .. (cf. fwd_step)
// the 8 twiddles of a step's last stage, held by the caller (several tiles per block, ntt_pass16_fwd_seq)
template <class A>
struct Tw16Last {
    typename A::W w8[8];
    __device__ __forceinline__ void load(const Ctx &c, int i) { A::tw_group(c, i << 3, 8, w8); }
};

// U0 = 1: the step's first stage (the one pairing word e with e + 8) has been taken by the pass in front (a column pass of one
// stage more, lf_ntt_ws): stages 1 .. 3 only, same table entries
template <class A, int U0 = 0>
__device__ __forceinline__ void fwd_regs16(typename A::T (&x)[16], int i0, const Ctx &c, const Tw16Early<A> *early = nullptr,
                                           const Tw16Last<A> *last = nullptr) {
#pragma unroll
    for (int u = U0; u < 4; ++u) {
        const int du = 1 << (3 - u);
        typename A::W wv[8];
        if (early && u < 3) early->get(1 << u, wv);
        else if (last && u == 3) {
#pragma unroll
            for (int j = 0; j < 8; ++j) wv[j] = last->w8[j];
        } else A::tw_group(c, i0 << u, 1 << u, wv);
#pragma unroll
        for (int j = 0; j < (1 << u); ++j) {
            const int e0 = j << (4 - u);
#pragma unroll
            for (int t = 0; t < du; ++t) A::fwd(c, x[e0 + t], x[e0 + t + du], wv[j], (i0 << u) + j);
        }
    }
    A::fwd_end(c, x);
}

// exact fp64 class (ArithDp semantics, bit for bit): products of a stage first, one uniform test, then add / sub
template <int U0 = 0>
__device__ __forceinline__ void fwd_regs16_exact(double (&x)[16], int i0, const Ctx &c, const Tw16Early<ArithDp> *early = nullptr,
                                                 const Tw16Last<ArithDp> *last = nullptr) {
#pragma unroll
    for (int u = U0; u < 4; ++u) {
        const int du = 1 << (3 - u);
        double wv[8];
        if (early && u < 3) early->get(1 << u, wv);
        else if (last && u == 3) {
#pragma unroll
            for (int j = 0; j < 8; ++j) wv[j] = last->w8[j];
        } else ArithDp::tw_group(c, i0 << u, 1 << u, wv);
        double V[8];
#pragma unroll
        for (int j = 0; j < (1 << u); ++j) {
            const int e0 = j << (4 - u);
#pragma unroll
            for (int t = 0; t < du; ++t) V[j * du + t] = dp_mulmod(x[e0 + t + du], wv[j], c.d);
        }
        // smallest high word of the 8 canonical products (signed: -0.0 counts as below, like dp_below_fix_limit):
        // four three-way minima and one compare instead of eight compares
        int hmin = min(min(__double2hiint(V[0]), __double2hiint(V[1])), __double2hiint(V[2]));
        hmin = min(min(hmin, __double2hiint(V[3])), __double2hiint(V[4]));
        hmin = min(min(hmin, __double2hiint(V[5])), __double2hiint(V[6]));
        hmin = min(hmin, __double2hiint(V[7]));
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(hmin < 0x41500000) != 0, 0)) {
#pragma unroll
            for (int j = 0; j < (1 << u); ++j) {
                const int e0 = j << (4 - u);
#pragma unroll
                for (int t = 0; t < du; ++t) {
                    double &v = V[j * du + t];
                    if (dp_below_fix_limit(v))
                        v = dp_lazy_fix(v, (u64)c.tw_mont[(i0 << u) + j], (u64)dp_reduce(x[e0 + t + du], c.d.q2, c.d.q2inv), c.d.q);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < (1 << u); ++j) {
            const int e0 = j << (4 - u);
#pragma unroll
            for (int t = 0; t < du; ++t) {
                const double U = x[e0 + t], Vv = V[j * du + t];
                x[e0 + t] = U + Vv;
                x[e0 + t + du] = U - Vv;
            }
        }
    }
}

// inverse: stage u of the step uses the 2^(3-u) table entries (il << (3-u)) .. (cf. inv_step); A::inv_end reduces
template <class A>
__device__ __forceinline__ void inv_regs16(typename A::T (&x)[16], int il, const Ctx &c, const Tw16Early<A> *early = nullptr) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int du = 1 << u;
        typename A::W wv[8];
        if (early && u > 0) early->get(1 << (3 - u), wv);
        else A::tw_group(c, il << (3 - u), 1 << (3 - u), wv);
#pragma unroll
        for (int h = 0; h < (1 << (3 - u)); ++h) {
            const int e0 = h << (u + 1);
#pragma unroll
            for (int t = 0; t < du; ++t) A::inv(c, x[e0 + t], x[e0 + t + du], wv[h], (il << (3 - u)) + h);
        }
    }
    A::inv_end(c, x);
}

// exact fp64 class, inverse (ArithDp::inv semantics): O = U - V, b = lazy REDC62(S * O), a = U + V
__device__ __forceinline__ void inv_regs16_exact(double (&x)[16], int il, const Ctx &c, const Tw16Early<ArithDp> *early = nullptr) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int du = 1 << u;
        double wv[8];
        if (early && u > 0) early->get(1 << (3 - u), wv);
        else ArithDp::tw_group(c, il << (3 - u), 1 << (3 - u), wv);
        double O[8], V[8];
#pragma unroll
        for (int h = 0; h < (1 << (3 - u)); ++h) {
            const int e0 = h << (u + 1);
#pragma unroll
            for (int t = 0; t < du; ++t) {
                const int k = h * du + t;
                O[k] = x[e0 + t] - x[e0 + t + du];
                V[k] = dp_mulmod(O[k], wv[h], c.d);
            }
        }
        int hmin = min(min(__double2hiint(V[0]), __double2hiint(V[1])), __double2hiint(V[2]));
        hmin = min(min(hmin, __double2hiint(V[3])), __double2hiint(V[4]));
        hmin = min(min(hmin, __double2hiint(V[5])), __double2hiint(V[6]));
        hmin = min(hmin, __double2hiint(V[7]));
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(hmin < 0x41500000) != 0, 0)) {
#pragma unroll
            for (int h = 0; h < (1 << (3 - u)); ++h) {
#pragma unroll
                for (int t = 0; t < du; ++t) {
                    const int k = h * du + t;
                    if (dp_below_fix_limit(V[k]))
                        V[k] = dp_lazy_fix(V[k], (u64)c.tw_mont[(il << (3 - u)) + h], (u64)dp_reduce(O[k], c.d.q2, c.d.q2inv), c.d.q);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < (1 << (3 - u)); ++h) {
            const int e0 = h << (u + 1);
#pragma unroll
            for (int t = 0; t < du; ++t) {
                const int k = h * du + t;
                x[e0 + t] = x[e0 + t] + x[e0 + t + du];
                x[e0 + t + du] = V[k];
            }
        }
    }
    // sums doubled four times from below 2q: back to a representative in [0, 2q) (dp_reduce takes |x| < 64 * 2q)
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = dp_reduce(x[e], c.d.q2, c.d.q2inv);
}

// ---- odd tiles: the reference's signed arithmetic, stage by stage (rare; any thread count) -------------------------
// (the tile's words are in LDS at PAD16(L), a barrier behind them — and again on return)
// (j0 .. j1 - 1 of the tile's 12 stages: a pass in front / behind may have taken the first / will take the last)
__device__ __forceinline__ void tile16_slow_compute(i64 *sm, int base, int s0, int logN, bool inverse, const Ctx &c, int j0 = 0,
                                                    int j1 = 12) {
    for (int j = j0; j < j1; ++j) {
        const int st = s0 + j;
        const int logd = inverse ? j : 11 - j;
        for (int i = threadIdx.x; i < 2048; i += NTT16_THREADS) {
            const int pa = ((i >> logd) << (logd + 1)) | (i & ((1 << logd) - 1)), pb = pa + (1 << logd);
            i64 a = sm[PAD16(pa)], b = sm[PAD16(pb)];
            if (!inverse) {
                const int idx = (1 << st) + ((base + pa) >> (logN - st));
                ArithInt<true>::fwd(c, a, b, c.tw_mont[idx], idx);
            } else {
                const int idx = (1 << (logN - st - 1)) + ((base + pa) >> (st + 1));
                ArithInt<true>::inv(c, a, b, c.tw_mont[idx], idx);
            }
            sm[PAD16(pa)] = a;
            sm[PAD16(pb)] = b;
        }
        __syncthreads();
    }
}
__device__ __forceinline__ void tile16_slow_lds(i64 *sm, i64 *dst, int base, int s0, int logN, bool inverse, const Ctx &c, int j0 = 0,
                                                int j1 = 12) {
    tile16_slow_compute(sm, base, s0, logN, inverse, c, j0, j1);
    for (int L = threadIdx.x; L < 4096; L += NTT16_THREADS) dst[L] = sm[PAD16(L)];
}
__device__ __forceinline__ void tile16_slow(i64 *sm, const i64 *src, i64 *dst, int base, int s0, int logN, bool inverse, const Ctx &c) {
    for (int L = threadIdx.x; L < 4096; L += NTT16_THREADS) sm[PAD16(L)] = src[L];
    __syncthreads();
    tile16_slow_lds(sm, dst, base, s0, logN, inverse, c);
}

// wave v owns byte v of the flag word: every wave writes its byte, so the word needs no reset
__device__ __forceinline__ void wave_flag_set16(i64 *sm, int pred, int w) {
    const bool any = __builtin_amdgcn_ballot_w64(pred != 0) != 0;
    if ((w & 63) == 0) reinterpret_cast<unsigned char *>(sm + NTT16_FLAG)[w >> 6] = any ? 1 : 0;
}
__device__ __forceinline__ bool tile_flagged16(const i64 *sm) { return reinterpret_cast<const unsigned *>(sm + NTT16_FLAG)[0] != 0; }

template <bool RLX> struct DpArith { typedef ArithDp type; };
template <> struct DpArith<true> { typedef ArithDpR type; };

// one radix-16 step of arithmetic class A in mode RLX (the exact fp64 class has its own stage-wise routine)
template <class A, bool EXACT_DP, int U0 = 0>
__device__ __forceinline__ void fwd_step16(typename A::T (&x)[16], int i0, const Ctx &c, const Tw16Early<A> *early,
                                           const Tw16Last<A> *last = nullptr) {
    if constexpr (EXACT_DP) fwd_regs16_exact<U0>(x, i0, c, early, last);
    else fwd_regs16<A, U0>(x, i0, c, early, last);
}
template <class A, bool EXACT_DP>
__device__ __forceinline__ void inv_step16(typename A::T (&x)[16], int il, const Ctx &c, const Tw16Early<A> *early) {
    if constexpr (EXACT_DP) inv_regs16_exact(x, il, c, early);
    else inv_regs16<A>(x, il, c, early);
}

// ---- forward tile: words w + 256 e in, 16 w + e out ------------------------------------------------------------------
// A = arithmetic class of the limb; T = its word type in LDS (double / i64).  The twiddles of the next step's first
// three stages are requested before the barrier that ends the current one.
template <class A, bool DP, bool RLX, int U0 = 0>
__device__ __forceinline__ bool fwd_tile16_steps(typename A::T *smt, const i64 *sm, typename A::T (&x)[16], int w, int base,
                                                 int E, int s, const Ctx &c, bool check, const Tw16Last<A> *lastC = nullptr) {
    constexpr bool EX = DP && !RLX;
    const int iA = (1 << s) + (base >> (E - s));                 // tile-uniform
    const int pB = ((w >> 4) << 8) | (w & 15);
    const int iB = (1 << (s + 4)) + ((base + pB) >> (E - s - 4));
    const int iC = (1 << (s + 8)) + ((base + 16 * w) >> (E - s - 8));
    Tw16Early<A> twB, twC;
    fwd_step16<A, EX, U0>(x, iA, c, nullptr);
    {
        typename A::T *sp = smt + PAD16(w);                      // PAD16(w + 256 e) = PAD16(w) + 272 e
#pragma unroll
        for (int e = 0; e < 16; ++e) sp[e * 272] = x[e];
    }
    twB.load(c, iB);
    lds_barrier();
    if (check && tile_flagged16(sm)) return false;
    {
        typename A::T *sp = smt + PAD16(pB);                     // PAD16(p + 16 e) = PAD16(p) + 17 e
#pragma unroll
        for (int e = 0; e < 16; ++e) x[e] = sp[e * 17];
        fwd_step16<A, EX>(x, iB, c, &twB);
#pragma unroll
        for (int e = 0; e < 16; ++e) sp[e * 17] = x[e];
    }
    twC.load(c, iC);
    lds_barrier();
    {
        const typename A::T *sp = smt + 17 * w;                  // PAD16(16 w + e) = 17 w + e
#pragma unroll
        for (int e = 0; e < 16; ++e) x[e] = sp[e];
    }
    fwd_step16<A, EX>(x, iC, c, &twC, lastC);
    return true;
}

// arithmetic class of a forward tile
template <bool DP, bool RLX> struct FwdArith { typedef ArithInt<false> type; };
template <> struct FwdArith<true, false> { typedef ArithDp type; };
template <> struct FwdArith<true, true> { typedef ArithDpR type; };
template <> struct FwdArith<false, true> { typedef ArithShoup type; };

// PLN (fp64 class, relaxed: the extended digits of a key switch between ks_ext_cols, this pass and the inner product): the row
// holds its canonical words (< 2^41) as two planes — u32 low[N] at byte 0, u16 high[N] at byte 4 N — 6 bytes per word
// instead of 8 on each of the digits' three trips.  A tile reads and writes the same 16 KiB + 8 KiB: in place as before.
template <bool DP, bool RLX, bool PLN = false>
__device__ __forceinline__ void fwd_tile16(i64 *sm, i64 *__restrict__ row, int tile, const PassGeom &g, const Ctx &c,
                                           const Tw16Last<typename FwdArith<DP, RLX>::type> *lastC = nullptr) {
    static_assert(!PLN || (DP && RLX), "planes: relaxed fp64-class rows only");
    const int w = lf_tid();
    const int base = tile << 12, E = g.logN, s = g.s0;
    constexpr bool CHECK = !RLX;   // relaxed tiles take the canonical words the library's own first pass wrote
    constexpr bool NTL = RLX ? NT_RLOAD : NT_EXACT, NTS = RLX ? NT_RSTORE : NT_EXACT;   // streaming accesses, ckks_ntt_core.h
    i64 raw[16];
    unsigned plo[PLN ? 16 : 1];
    unsigned short phi[PLN ? 16 : 1];
    if constexpr (PLN) {
        const unsigned *lo = reinterpret_cast<const unsigned *>(row) + base;
        const unsigned short *hi = reinterpret_cast<const unsigned short *>(row + ((i64)1 << (E - 1))) + base;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            plo[e] = __builtin_nontemporal_load(uniform_ptr(lo + (e << 8)) + (unsigned)w);
            phi[e] = __builtin_nontemporal_load(uniform_ptr(hi + (e << 8)) + (unsigned)w);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {   // SGPR row pointers + lane index
            if constexpr (NTL) raw[e] = __builtin_nontemporal_load(uniform_row(row + base, e << 8) + (unsigned)w);
            else raw[e] = uniform_row(row + base, e << 8)[(unsigned)w];
        }
    }
    if (CHECK) {
        int odd = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) odd |= ((u64)raw[e] >= (u64)c.m.q2);
        wave_flag_set16(sm, odd, w);
    }
    i64 o[16];
    bool ok;
    if constexpr (DP) {
        typedef typename DpArith<RLX>::type AD;
        double x[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) x[e] = PLN ? dp_from_planes(plo[e], phi[e]) : RLX ? dp_from_signed(raw[e]) : dp_from_word(raw[e]);
        ok = fwd_tile16_steps<AD, true, RLX>(reinterpret_cast<double *>(sm), sm, x, w, base, E, s, c, CHECK, lastC);
        // the pass accumulated without subtractions: back to the lazy word in [0, 2q) (relaxed: canonical residue)
        const double md = RLX ? c.d.q : c.d.q2, mi = RLX ? c.d.qinv : c.d.q2inv;
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = dp_to_word(dp_reduce(x[e], md, mi));
    } else if constexpr (RLX) {   // residues only: Shoup products on lazy words (ArithShoup), canonical on the way out
        ok = fwd_tile16_steps<ArithShoup, false, true>(sm, sm, raw, w, base, E, s, c, false, lastC);
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = ArithShoup::canon(c, raw[e]);
    } else {
        ok = fwd_tile16_steps<ArithInt<false>, false, RLX>(sm, sm, raw, w, base, E, s, c, true, lastC);
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = raw[e];
    }
    if (!ok) {   // a word outside [0, 2q): nothing has been stored yet
        __syncthreads();
        tile16_slow(sm, row + base, row + base, base, s, E, false, c);
        return;
    }
    if constexpr (PLN) {
        // the wave's own LDS span (17 * 64 words = 2 176 dwords; its last reads were this wave's own): 16 low dwords per lane
        // 20 apart, behind them 8 dwords of packed high halves per lane 12 apart (strides that keep the 16-byte accesses of 16
        // lanes on distinct banks on the way in) -> 16-byte stores, 1 KiB per instruction, on both planes
        const int lane = w & 63, wave = w >> 6;
        unsigned *sw = reinterpret_cast<unsigned *>(sm + 17 * 64 * wave);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const lf_u4_t v = {(unsigned)o[4 * k], (unsigned)o[4 * k + 1], (unsigned)o[4 * k + 2], (unsigned)o[4 * k + 3]};
            *reinterpret_cast<lf_u4_t *>(sw + lane * 20 + 4 * k) = v;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            lf_u4_t v;
#pragma unroll
            for (int m = 0; m < 4; ++m)
                v[m] = (unsigned)(o[8 * k + 2 * m] >> 32) | ((unsigned)(o[8 * k + 2 * m + 1] >> 32) << 16);
            *reinterpret_cast<lf_u4_t *>(sw + 1280 + lane * 12 + 4 * k) = v;
        }
        wave_lds_sync();
        unsigned *lo = reinterpret_cast<unsigned *>(row) + base + (wave << 10);
        unsigned *hi = reinterpret_cast<unsigned *>(row + ((i64)1 << (E - 1))) + ((base + (wave << 10)) >> 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // words 256 i + 4 lane .. + 3 of the span: thread 16 i + lane / 4, its quad lane % 4
            const lf_u4_t v = *reinterpret_cast<const lf_u4_t *>(sw + (16 * i + (lane >> 2)) * 20 + 4 * (lane & 3));
            *reinterpret_cast<lf_u4_t *>(lo + (i << 8) + 4 * lane) = v;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {   // words 512 i + 8 lane .. + 7: thread 32 i + lane / 2, its half lane % 2
            const lf_u4_t v = *reinterpret_cast<const lf_u4_t *>(sw + 1280 + (32 * i + (lane >> 1)) * 12 + 4 * (lane & 1));
            *reinterpret_cast<lf_u4_t *>(hi + (i << 8) + 4 * lane) = v;
        }
        return;
    }
    // 16 consecutive result words per thread -> the wave's own 1024-word LDS span -> 16-byte stores, 1 KiB per instruction
    {
        i64 *sp = sm + 17 * w;
#pragma unroll
        for (int e = 0; e < 16; ++e) sp[e] = o[e];
    }
    wave_lds_sync();
    const int L0 = ((w >> 6) << 10) + ((w & 63) << 1);
    const i64 *so = sm + PAD16(L0);                              // L0 even: PAD16(L0 + 128 i + 1) = PAD16(L0) + 136 i + 1
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        longlong2 v;
        v.x = so[i * 136];
        v.y = so[i * 136 + 1];
        if constexpr (NTS) nt_store2(row + base + L0 + (i << 7), v);
        else *reinterpret_cast<longlong2 *>(row + base + L0 + (i << 7)) = v;
    }
}

// ---- forward tile of an EXACT transform whose column pass wrote a workspace (lf_ntt_ws) -----------------------------------
// The two passes of a logN >= 13 transform exchange every word through HBM once.  With a workspace the column pass
// (fwd_cols_ws_body, ckks_ntt_core.h) leaves the fp64-class rows there as PLANES — u32 low[N] at byte 0, u16 mid[N] at byte 4 N:
// a lazy word is below 2^42 — and this pass reads 6 bytes per word instead of 8; the result goes to the tensor as raw words.
// Words the reference's arithmetic produces from operands outside [0, 2q) can be anything: a column wave that met such an
// operand also writes the words' top 16 bits (third plane at byte 6 N) and raises its flag byte wf[v][e] (columns 256 e + 64 v
// .. + 63); the tile's wave v reads exactly the words of those column waves, e = 0 .. 15, so it tests its own 16 flag bytes
// (two scalar loads), fetches the top plane only where one is set — and tests its words against 2q only then: the words of an
// unflagged column wave are lazy words by construction.  Integer-class rows are raw words in the workspace, flagged alike.
// returns whether one of the column waves behind this wave's words raised its flag (wave-uniform): only then can a word lie
// outside [0, 2q)
// SKIP0: the column pass took one stage more (columns are 2048 wide: word e and word e + 8 of a thread come from the same
// column wave, flag index e & 7)
template <bool DP, bool SKIP0>
__device__ __forceinline__ bool ws_load_tile(const i64 *__restrict__ srow, const unsigned char *__restrict__ wf, int base, int E,
                                             int w, i64 (&raw)[16]) {
    const int wave = __builtin_amdgcn_readfirstlane(w >> 6);
    const u64 *fp = reinterpret_cast<const u64 *>(wf + (wave << 4));
    const u64 fl = fp[0], fh = SKIP0 ? fp[0] : fp[1];
    if constexpr (DP) {
        const unsigned *lo = reinterpret_cast<const unsigned *>(srow) + base;
        const unsigned short *mid = reinterpret_cast<const unsigned short *>(srow + ((i64)1 << (E - 1))) + base;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const unsigned l = __builtin_nontemporal_load(uniform_ptr(lo + (e << 8)) + (unsigned)w);
            const unsigned m = __builtin_nontemporal_load(uniform_ptr(mid + (e << 8)) + (unsigned)w);
            raw[e] = (i64)(((u64)m << 32) | (u64)l);
        }
        if ((fl | fh) != 0) {
            const unsigned short *top = reinterpret_cast<const unsigned short *>(srow + 3 * ((i64)1 << (E - 2))) + base;
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if ((((e < 8 ? fl : fh) >> (8 * (e & 7))) & 0xffull) != 0) raw[e] |= (i64)((u64)top[(e << 8) + w] << 48);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) raw[e] = __builtin_nontemporal_load(uniform_ptr(srow + base + (e << 8)) + (unsigned)w);
    }
    return (fl | fh) != 0;
}

template <bool DP, bool SKIP0>
__device__ __forceinline__ void fwd_tile16_ws(i64 *sm, const i64 *__restrict__ srow, const unsigned char *__restrict__ wf,
                                              i64 *__restrict__ row, int tile, const PassGeom &g, const Ctx &c,
                                              const Tw16Last<typename FwdArith<DP, false>::type> *lastC = nullptr) {
    const int w = lf_tid();
    const int base = tile << 12, E = g.logN, s = g.s0;
    i64 raw[16];
    const bool flagged = ws_load_tile<DP, SKIP0>(srow, wf, base, E, w, raw);
    {
        int odd = 0;
        if (flagged) {   // (words of unflagged column waves are lazy words by construction: no test)
#pragma unroll
            for (int e = 0; e < 16; ++e) odd |= ((u64)raw[e] >= (u64)c.m.q2);
        }
        wave_flag_set16(sm, odd, w);
    }
    i64 o[16];
    bool ok;
    if constexpr (DP) {
        double x[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) x[e] = dp_from_word(raw[e]);
        ok = fwd_tile16_steps<ArithDp, true, false, SKIP0 ? 1 : 0>(reinterpret_cast<double *>(sm), sm, x, w, base, E, s, c, true, lastC);
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = dp_to_word(dp_reduce(x[e], c.d.q2, c.d.q2inv));
    } else {
        ok = fwd_tile16_steps<ArithInt<false>, false, false, SKIP0 ? 1 : 0>(sm, sm, raw, w, base, E, s, c, true, lastC);
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = raw[e];
    }
    if (!ok) {   // a word outside [0, 2q): nothing has been stored yet; the raw words once more, into LDS, then the generic tile
        __syncthreads();
        ws_load_tile<DP, SKIP0>(srow, wf, base, E, w, raw);
#pragma unroll
        for (int e = 0; e < 16; ++e) sm[PAD16(w + (e << 8))] = raw[e];
        __syncthreads();
        tile16_slow_lds(sm, row + base, base, s, E, false, c, SKIP0 ? 1 : 0);
        return;
    }
    {
        i64 *sp = sm + 17 * w;
#pragma unroll
        for (int e = 0; e < 16; ++e) sp[e] = o[e];
    }
    wave_lds_sync();
    const int L0 = ((w >> 6) << 10) + ((w & 63) << 1);
    const i64 *so = sm + PAD16(L0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        longlong2 v;
        v.x = so[i * 136];
        v.y = so[i * 136 + 1];
        if constexpr (NT_EXACT) nt_store2(row + base + L0 + (i << 7), v);
        else *reinterpret_cast<longlong2 *>(row + base + L0 + (i << 7)) = v;
    }
}

// ---- inverse tile (first pass of a two-pass inverse transform, no chain tail): 16 w + e in, w + 256 e out ------------
template <class A, bool DP, bool RLX>
__device__ __forceinline__ bool inv_tile16_steps(typename A::T *smt, const i64 *sm, typename A::T (&x)[16], int w, int base,
                                                 int logN, int s, const Ctx &c, bool check) {
    constexpr bool EX = DP && !RLX;
    const int iC = (1 << (logN - s - 4)) + ((base + 16 * w) >> (s + 4));
    const int pB = ((w >> 4) << 8) | (w & 15);
    const int iB = (1 << (logN - s - 8)) + ((base + pB) >> (s + 8));
    const int iA = (1 << (logN - s - 12)) + (base >> (s + 12));   // tile-uniform
    Tw16Early<A> twB;
    inv_step16<A, EX>(x, iC, c, nullptr);
    {
        typename A::T *sp = smt + 17 * w;
#pragma unroll
        for (int e = 0; e < 16; ++e) sp[e] = x[e];
    }
    twB.load(c, iB);
    lds_barrier();
    if (check && tile_flagged16(sm)) return false;
    {
        typename A::T *sp = smt + PAD16(pB);
#pragma unroll
        for (int e = 0; e < 16; ++e) x[e] = sp[e * 17];
        inv_step16<A, EX>(x, iB, c, &twB);
#pragma unroll
        for (int e = 0; e < 16; ++e) sp[e * 17] = x[e];
    }
    lds_barrier();
    {
        const typename A::T *sp = smt + PAD16(w);
#pragma unroll
        for (int e = 0; e < 16; ++e) x[e] = sp[e * 272];
    }
    inv_step16<A, EX>(x, iA, c, nullptr);                         // tile-uniform twiddles: scalar loads
    return true;
}

// MUL (relaxed only): the tile is the element-wise product src_row * mul_row, formed as the words come in — cc_mult's
// x1 * y1 never exists in HBM.  fp64 class: plain canonical residues in, canonical product (held as doubles from here
// on); integer class: Montgomery-form words below 2q in, the lazy REDC62 product.
// PLN (fp64 class, relaxed: the sums of a key switch between the inner product and the two inverse passes; cc_mult's operand stack
// as the product-on-load pass reads it): source AND destination rows hold canonical words as planes (fwd_tile16).  A wave's
// 1024-word span arrives as 4 x 16 bytes of the low plane + 4 x 8 bytes of the high plane per lane (words 4 lane .. + 3 of every
// 256), assembled to words on the way into the LDS span; the results leave as one 4-byte and one 2-byte store per word.
template <bool DP, bool RLX, bool MUL = false, int PLN = 0>   // PLN: PLN_IN | PLN_OUT
__device__ __forceinline__ void inv_tile16(i64 *sm, const i64 *src_row, i64 *dst_row, int tile, const PassGeom &g, const Ctx &c,
                                           const i64 *mul_row = nullptr) {
    static_assert(!MUL || RLX, "product-on-load exists for the relaxed inverse transform only");
    static_assert(!PLN || (DP && RLX), "planes: relaxed fp64-class rows only");
    const int w = lf_tid();
    const int base = tile << 12, logN = g.logN, s = g.s0;
    constexpr bool CHECK = !RLX;   // relaxed inverse transforms take lazy words in [0, 2q) (include/ckks_hip.h)
    const int L0 = ((w >> 6) << 10) + ((w & 63) << 1);
    if constexpr ((PLN & PLN_IN) != 0) {
        const int L4 = ((w >> 6) << 10) + ((w & 63) << 2);
        const unsigned *lo = reinterpret_cast<const unsigned *>(src_row) + base + L4;
        const unsigned *hi = reinterpret_cast<const unsigned *>(src_row + ((i64)1 << (logN - 1))) + ((base + L4) >> 1);
        lf_u4_t l[4], lb[4];
        lf_u2_t h[4], hb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            l[i] = __builtin_nontemporal_load(reinterpret_cast<const lf_u4_t *>(lo + (i << 8)));
            h[i] = __builtin_nontemporal_load(reinterpret_cast<const lf_u2_t *>(hi + (i << 7)));
        }
        if (MUL) {
            const unsigned *mlo = reinterpret_cast<const unsigned *>(mul_row) + base + L4;
            const unsigned *mhi = reinterpret_cast<const unsigned *>(mul_row + ((i64)1 << (logN - 1))) + ((base + L4) >> 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                lb[i] = __builtin_nontemporal_load(reinterpret_cast<const lf_u4_t *>(mlo + (i << 8)));
                hb[i] = __builtin_nontemporal_load(reinterpret_cast<const lf_u2_t *>(mhi + (i << 7)));
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            i64 *sp = sm + PAD16(L4 + (i << 8));                  // 4-aligned + m < 4: no padding step inside the quad
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const unsigned hm = (h[i][m >> 1] >> (16 * (m & 1))) & 0xffffu;
                if (MUL) {
                    const unsigned hbm = (hb[i][m >> 1] >> (16 * (m & 1))) & 0xffffu;
                    sp[m] = __double_as_longlong(dp_mulmod(dp_from_planes(l[i][m], hm), dp_from_planes(lb[i][m], hbm), c.d));
                } else {
                    sp[m] = (i64)(((u64)hm << 32) | (u64)l[i][m]);
                }
            }
        }
    } else {
        longlong2 in[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) in[i] = INV_LD2(src_row + base + L0 + (i << 7));
        if (MUL) {
            longlong2 mb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) mb[i] = INV_LD2(mul_row + base + L0 + (i << 7));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (DP) {
                    in[i].x = __double_as_longlong(dp_mulmod(dp_from_word(in[i].x), dp_from_word(mb[i].x), c.d));
                    in[i].y = __double_as_longlong(dp_mulmod(dp_from_word(in[i].y), dp_from_word(mb[i].y), c.d));
                } else {
                    in[i].x = mm62u((u64)in[i].x, (u64)mb[i].x, c.m.q, c.m.k);
                    in[i].y = mm62u((u64)in[i].y, (u64)mb[i].y, c.m.q, c.m.k);
                }
            }
        }
        int odd = 0;
        i64 *sp = sm + PAD16(L0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (CHECK) odd |= ((u64)in[i].x >= (u64)c.m.q2) | ((u64)in[i].y >= (u64)c.m.q2);
            sp[i * 136] = in[i].x;
            sp[i * 136 + 1] = in[i].y;
        }
        if (CHECK) wave_flag_set16(sm, odd, w);
    }
    wave_lds_sync();
    i64 raw[16];
    {
        const i64 *sp = sm + 17 * w;
#pragma unroll
        for (int e = 0; e < 16; ++e) raw[e] = sp[e];
    }
    i64 *out = dst_row + base;                                    // wave-uniform; the lane index is added per store
    Ctx cc = c;
    cc.inv_reduce = 1;                                            // fp64 classes: fold at the end of every radix-16 step
    bool ok;
    if (DP) {
        typedef typename DpArith<RLX>::type AD;
        double x[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) x[e] = MUL ? __longlong_as_double(raw[e]) : dp_from_word(raw[e]);
        ok = inv_tile16_steps<AD, true, RLX>(reinterpret_cast<double *>(sm), sm, x, w, base, logN, s, cc, CHECK);
        if constexpr ((PLN & PLN_OUT) != 0) {   // (relaxed: ok is always true)
            unsigned *olo = reinterpret_cast<unsigned *>(dst_row) + base;
            unsigned short *ohi = reinterpret_cast<unsigned short *>(dst_row + ((i64)1 << (logN - 1))) + base;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const i64 o = dp_to_word(dp_addmask(x[e], c.d.q));
                uniform_ptr(olo + (e << 8))[(unsigned)w] = (unsigned)o;
                uniform_ptr(ohi + (e << 8))[(unsigned)w] = (unsigned short)((u64)o >> 32);
            }
        } else if (ok) {   // exact: the lazy word in [0, 2q); relaxed: balanced residue -> canonical
#pragma unroll
            for (int e = 0; e < 16; ++e) INV_ST(uniform_row(out, e << 8) + (unsigned)w, dp_to_word(RLX ? dp_addmask(x[e], c.d.q) : x[e]));
        }
    } else if (RLX) {
        ok = inv_tile16_steps<ArithShoup, false, true>(sm, sm, raw, w, base, logN, s, cc, false);
#pragma unroll
        for (int e = 0; e < 16; ++e) INV_ST(uniform_row(out, e << 8) + (unsigned)w, ArithShoup::canon(c, raw[e]));
    } else {
        ok = inv_tile16_steps<ArithInt<false>, false, RLX>(sm, sm, raw, w, base, logN, s, cc, true);
        if (ok) {
#pragma unroll
            for (int e = 0; e < 16; ++e) INV_ST(uniform_row(out, e << 8) + (unsigned)w, raw[e]);
        }
    }
    if (!ok) {
        __syncthreads();
        tile16_slow(sm, src_row + base, dst_row + base, base, s, logN, true, c);
    }
}

// ---- inverse tile of an EXACT transform through a workspace (lf_intt_ws): tensor in, workspace out -----------------------
// The mirror image of fwd_tile16_ws: the tiled pass comes first and leaves the fp64-class rows in the workspace as planes for the
// column pass (inv_cols_compute<.., WS>).  A tile that left the fast form (an operand outside [0, 2q): the reference's signed
// arithmetic, whose words can be anything) writes all three planes and raises tflags[tile]; every other tile clears it, and
// the column pass tests the words of unflagged tiles neither for their top plane nor against 2q.
template <bool DP>
__device__ __forceinline__ void inv_tile16_ws(i64 *sm, const i64 *src_row, i64 *ws_row, unsigned char *tflags, int tile,
                                              const PassGeom &g, const Ctx &c) {
    const int w = lf_tid();
    const int base = tile << 12, logN = g.logN, s = g.s0;
    const int L0 = ((w >> 6) << 10) + ((w & 63) << 1);
    {
        longlong2 in[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) in[i] = INV_LD2(src_row + base + L0 + (i << 7));
        int odd = 0;
        i64 *sp = sm + PAD16(L0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            odd |= ((u64)in[i].x >= (u64)c.m.q2) | ((u64)in[i].y >= (u64)c.m.q2);
            sp[i * 136] = in[i].x;
            sp[i * 136 + 1] = in[i].y;
        }
        wave_flag_set16(sm, odd, w);
    }
    wave_lds_sync();
    i64 raw[16];
    {
        const i64 *sp = sm + 17 * w;
#pragma unroll
        for (int e = 0; e < 16; ++e) raw[e] = sp[e];
    }
    Ctx cc = c;
    cc.inv_reduce = 1;
    unsigned *lo = reinterpret_cast<unsigned *>(ws_row) + base;
    unsigned short *mid = reinterpret_cast<unsigned short *>(ws_row + ((i64)1 << (logN - 1))) + base;
    bool ok;
    if constexpr (DP) {
        double x[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) x[e] = dp_from_word(raw[e]);
        ok = inv_tile16_steps<ArithDp, true, false>(reinterpret_cast<double *>(sm), sm, x, w, base, logN, s, cc, true);
        if (ok) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const i64 v = dp_to_word(x[e]);
                __builtin_nontemporal_store((unsigned)v, uniform_ptr(lo + (e << 8)) + (unsigned)w);
                __builtin_nontemporal_store((unsigned short)((u64)v >> 32), uniform_ptr(mid + (e << 8)) + (unsigned)w);
            }
            if (w == 0) tflags[tile] = 0;
        }
    } else {
        ok = inv_tile16_steps<ArithInt<false>, false, false>(sm, sm, raw, w, base, logN, s, cc, true);
        if (ok) {
#pragma unroll
            for (int e = 0; e < 16; ++e) INV_ST(uniform_ptr(ws_row + base + (e << 8)) + (unsigned)w, raw[e]);
            if (w == 0) tflags[tile] = 0;
        }
    }
    if (!ok) {
        __syncthreads();
        if constexpr (DP) {
            for (int L = threadIdx.x; L < 4096; L += NTT16_THREADS) sm[PAD16(L)] = src_row[base + L];
            __syncthreads();
            tile16_slow_compute(sm, base, s, logN, true, c);
            unsigned short *top = reinterpret_cast<unsigned short *>(ws_row + 3 * ((i64)1 << (logN - 2))) + base;
            for (int L = threadIdx.x; L < 4096; L += NTT16_THREADS) {
                const u64 v = (u64)sm[PAD16(L)];
                lo[L] = (unsigned)v, mid[L] = (unsigned short)(v >> 32), top[L] = (unsigned short)(v >> 48);
            }
            if (threadIdx.x == 0) tflags[tile] = 1;
        } else {
            tile16_slow(sm, src_row + base, ws_row + base, base, s, logN, true, c);
            if (threadIdx.x == 0) tflags[tile] = 1;
        }
    }
}

template <bool DP>
__device__ __forceinline__ void pass16_inv_ws_body(i64 *sm, int b, const i64 *src, i64 *ws, unsigned char *wflags, const PassGeom &g,
                                                   const RowList &rl, const i64 *__restrict__ tw_br,
                                                   const double *__restrict__ tw_dp, const i64 *__restrict__ ql,
                                                   const i64 *__restrict__ qh, const i64 *__restrict__ kl,
                                                   const i64 *__restrict__ kh) {
    int poly, crow, tile;
    block_coords(g, rl, b, poly, crow, tile);
    poly = __builtin_amdgcn_readfirstlane(poly), crow = __builtin_amdgcn_readfirstlane(crow);
    tile = __builtin_amdgcn_readfirstlane(tile);
    Ctx c;
    c.m = load_mod(ql, qh, kl, kh, crow);
    c.tw_mont = tw_br + ((i64)crow << g.logN);
    set_aux<DP>(c, tw_dp, crow, g.logN);
    c.d = DP ? make_dp_tab(c.m, c.tw_dp) : make_dp(c.m);
    c.relaxed = 0;
    c.inv_reduce = 0;
    const i64 ri = (i64)(poly * g.rows + crow);
    inv_tile16_ws<DP>(sm, src + (ri << g.logN), ws + (ri << g.logN), wflags + (ri << 6), tile, g, c);
}

__global__ void __launch_bounds__(NTT16_THREADS, 4) ntt_pass16_inv_ws(const i64 *src, i64 *ws, unsigned char *wflags, PassGeom g,
                                                                        ClassLists cl, const i64 *__restrict__ tw_br,
                                                                        const double *__restrict__ tw_dp,
                                                                        const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                                        const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    __shared__ i64 sm[NTT16_LDS_WORDS + 1];
    const int b = blockIdx.x;
    if (b < cl.in_blocks) {
        if (b < cl.in_real) pass16_inv_ws_body<false>(sm, b, src, ws, wflags, g, cl.in, tw_br, tw_dp, ql, qh, kl, kh);
    } else {
        pass16_inv_ws_body<true>(sm, b - cl.in_blocks, src, ws, wflags, g, cl.dp, tw_br, tw_dp, ql, qh, kl, kh);
    }
}

// ---- kernels: both arithmetic classes in one launch (integer-class blocks first), see ntt_fwd_pass_mixed -----------
// second factor of a product-on-load inverse pass (MUL): polynomial `poly` of the pass reads a + poly * a_stride and
// multiplies by b + poly * b_stride (strides in words; the factors may sit anywhere, e.g. x1 and y1 of cc_mult's stack)
struct MulSrc {
    const i64 *b;
    i64 a_stride, b_stride;
};

template <bool DP, bool RLX, bool INV, bool MUL = false, int PLN = 0>   // PLN: forward: planes in place; inverse: PLN_IN | PLN_OUT
__device__ __forceinline__ void pass16_body(i64 *sm, int b, const i64 *src, i64 *dst, const PassGeom &g, const RowList &rl,
                                            const i64 *__restrict__ tw_br, const double *__restrict__ tw_dp,
                                            const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                            const i64 *__restrict__ kl, const i64 *__restrict__ kh, const MulSrc *ms = nullptr) {
    int poly, crow, tile;
    block_coords(g, rl, b, poly, crow, tile);
    // (integer divisions run on the VALU: pin the wave-uniform coordinates back into SGPRs)
    poly = __builtin_amdgcn_readfirstlane(poly), crow = __builtin_amdgcn_readfirstlane(crow);
    tile = __builtin_amdgcn_readfirstlane(tile);
    if (!INV && RLX && g.skip_own != nullptr && (int)g.skip_own[crow] == g.skip_off + poly % g.skip_mod) return;   // see PassGeom
    Ctx c;
    c.m = load_mod(ql, qh, kl, kh, crow);
    c.tw_mont = tw_br + ((i64)crow << g.logN);
    set_aux<DP>(c, tw_dp, crow, g.logN);
    c.d = DP ? make_dp_tab(c.m, c.tw_dp) : make_dp(c.m);
    c.relaxed = RLX ? 1 : 0;
    c.inv_reduce = 0;
    const i64 off = (i64)(poly * g.rows + crow) << g.logN;
    if constexpr (INV && MUL) {
        const i64 roff = (i64)crow << g.logN;
        inv_tile16<DP, RLX, true, PLN>(sm, src + (i64)poly * ms->a_stride + roff, dst + off, tile, g, c,
                                       ms->b + (i64)poly * ms->b_stride + roff);
    } else if constexpr (INV) {
        inv_tile16<DP, RLX, false, PLN>(sm, src + off, dst + off, tile, g, c);
    } else {
        fwd_tile16<DP, RLX, PLN != 0>(sm, dst + off, tile, g, c);
    }
}

template <bool RLX, bool INV>
__global__ void __launch_bounds__(NTT16_THREADS, 4) ntt_pass16_mixed(const i64 *src, i64 *dst, PassGeom g, ClassLists cl,
                                                                       const i64 *__restrict__ tw_br,
                                                                       const double *__restrict__ tw_dp,
                                                                       const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                                       const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    __shared__ i64 sm[NTT16_LDS_WORDS + 1];
    const int b = blockIdx.x;
    if (b < cl.in_blocks) {
        if (b < cl.in_real) pass16_body<false, RLX, INV>(sm, b, src, dst, g, cl.in, tw_br, tw_dp, ql, qh, kl, kh);
    } else {
        pass16_body<true, RLX, INV>(sm, b - cl.in_blocks, src, dst, g, cl.dp, tw_br, tw_dp, ql, qh, kl, kh);
    }
}

// relaxed forward pass over extended digits whose fp64-class rows are in planes format (fwd_tile16<.., PLN>), both classes
__global__ void __launch_bounds__(NTT16_THREADS, 4) ntt_pass16_fwd_planes(i64 *dst, PassGeom g, ClassLists cl,
                                                                            const i64 *__restrict__ tw_br,
                                                                            const double *__restrict__ tw_dp,
                                                                            const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                                            const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    __shared__ i64 sm[NTT16_LDS_WORDS + 1];
    const int b = blockIdx.x;
    if (b < cl.in_blocks) {
        if (b < cl.in_real) pass16_body<false, true, false>(sm, b, dst, dst, g, cl.in, tw_br, tw_dp, ql, qh, kl, kh);
    } else {
        pass16_body<true, true, false, false, PLN_IN | PLN_OUT>(sm, b - cl.in_blocks, dst, dst, g, cl.dp, tw_br, tw_dp, ql, qh, kl, kh);
    }
}

// relaxed inverse pass whose fp64-class rows are planes on both sides (inv_tile16<.., PLN>): the sums of a key switch, src -> dst
__global__ void __launch_bounds__(NTT16_THREADS, 4) ntt_pass16_inv_planes(const i64 *src, i64 *dst, PassGeom g, ClassLists cl,
                                                                            const i64 *__restrict__ tw_br,
                                                                            const double *__restrict__ tw_dp,
                                                                            const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                                            const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    __shared__ i64 sm[NTT16_LDS_WORDS + 1];
    const int b = blockIdx.x;
    if (b < cl.in_blocks) {
        if (b < cl.in_real) pass16_body<false, true, true>(sm, b, src, dst, g, cl.in, tw_br, tw_dp, ql, qh, kl, kh);
    } else {
        pass16_body<true, true, true, false, PLN_IN | PLN_OUT>(sm, b - cl.in_blocks, src, dst, g, cl.dp, tw_br, tw_dp, ql, qh, kl, kh);
    }
}

// relaxed inverse pass of a product whose factors' fp64-class rows are planes (cc_mult's operand stack); the product leaves as raw
// words (the column pass behind runs in place, and a pass that changes the format cannot)
__global__ void __launch_bounds__(NTT16_THREADS, 4) ntt_pass16_mul_planes(const i64 *src, i64 *dst, PassGeom g, ClassLists cl, MulSrc ms,
                                                                            const i64 *__restrict__ tw_br,
                                                                            const double *__restrict__ tw_dp,
                                                                            const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                                            const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    __shared__ i64 sm[NTT16_LDS_WORDS + 1];
    const int b = blockIdx.x;
    if (b < cl.in_blocks) {
        if (b < cl.in_real) pass16_body<false, true, true, true>(sm, b, src, dst, g, cl.in, tw_br, tw_dp, ql, qh, kl, kh, &ms);
    } else {
        pass16_body<true, true, true, true, PLN_IN>(sm, b - cl.in_blocks, src, dst, g, cl.dp, tw_br, tw_dp, ql, qh, kl, kh, &ms);
    }
}

// relaxed inverse pass of a product (MulSrc), both classes
__global__ void __launch_bounds__(NTT16_THREADS, 4) ntt_pass16_mul_mixed(const i64 *src, i64 *dst, PassGeom g, ClassLists cl, MulSrc ms,
                                                                           const i64 *__restrict__ tw_br,
                                                                           const double *__restrict__ tw_dp,
                                                                           const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                                           const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    __shared__ i64 sm[NTT16_LDS_WORDS + 1];
    const int b = blockIdx.x;
    if (b < cl.in_blocks) {
        if (b < cl.in_real) pass16_body<false, true, true, true>(sm, b, src, dst, g, cl.in, tw_br, tw_dp, ql, qh, kl, kh, &ms);
    } else {
        pass16_body<true, true, true, true>(sm, b - cl.in_blocks, src, dst, g, cl.dp, tw_br, tw_dp, ql, qh, kl, kh, &ms);
    }
}

// ---- several tiles per block, the last stage's twiddles kept in registers --------------------------------------------
// Consecutive virtual blocks of one XCD lane (b0, b0 + 8, ..) are the same (limb, tile) pair of consecutive polynomials:
// the 8 twiddles per thread of the transform's LAST stage — half of all the twiddle bytes a tile reads, and the only group
// that is requested where it is used — are loaded once per pair and block instead of once per tile.
struct TileAt {
    int live, poly, crow, tile;
};
// block b of class list rl (live: inside the class's real range); scalar reads only (a vector load here would put a full
// vmcnt wait — behind the previous tile's stores — at the top of every tile; sub-dword loads exist as vector loads only)
__device__ __forceinline__ TileAt tile_at(const PassGeom &g, const RowList &rl, int b, bool live) {
    TileAt t;
    t.live = 0, t.poly = 0, t.crow = 0, t.tile = 0;
    if (!live) return t;
    const int tiles = 1 << (g.logN - g.tl);
    const int pairs = rl.n * tiles;
    int pair, poly;
    if ((pairs & 7) == 0) {
        const int r = b >> 3;
        poly = r % g.batch;
        pair = (r / g.batch) * 8 + (b & 7);
    } else {
        pair = b / g.batch;
        poly = b % g.batch;
    }
    pair = __builtin_amdgcn_readfirstlane(pair);
    t.poly = __builtin_amdgcn_readfirstlane(poly);
    const int li = pair >> (g.logN - g.tl);
    t.crow = (int)((reinterpret_cast<const unsigned *>(rl.id)[li >> 1] >> ((li & 1) << 4)) & 0xffffu);
    t.tile = pair & (tiles - 1);
    t.live = 1;
    return t;
}

template <bool DP, bool RLX>
__device__ __forceinline__ void seq16_loop(i64 *sm, i64 *dst, const PassGeom &g, const RowList &rl, int b0, int bend, int tpb,
                                           const i64 *__restrict__ tw_br, const double *__restrict__ tw_dp,
                                           const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                           const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    typedef typename FwdArith<DP, RLX>::type AC;
    const int w = lf_tid();
    Tw16Last<AC> last;
    int last_crow = -1, last_tile = -1;
    for (int i = 0; i < tpb; ++i) {
        const int b = b0 + 8 * i;
        const TileAt t = tile_at(g, rl, b, b < bend);
        if (!t.live) break;
        Ctx c;
        c.m = load_mod(ql, qh, kl, kh, t.crow);
        c.tw_mont = tw_br + ((i64)t.crow << g.logN);
        set_aux<DP>(c, tw_dp, t.crow, g.logN);
        c.d = DP ? make_dp_tab(c.m, c.tw_dp) : make_dp(c.m);
        c.relaxed = RLX ? 1 : 0;
        c.inv_reduce = 0;
        if (t.crow != last_crow || t.tile != last_tile) {
            last.load(c, (1 << (g.s0 + 8)) + (((t.tile << 12) + 16 * w) >> (g.logN - g.s0 - 8)));
            last_crow = t.crow, last_tile = t.tile;
        }
        if (i) lds_barrier();   // the waves' store spans are rewritten block-wide by this tile's first exchange
        fwd_tile16<DP, RLX>(sm, dst + ((i64)(t.poly * g.rows + t.crow) << g.logN), t.tile, g, c, &last);
    }
}

// cl.in_blocks is a multiple of 8 * tpb here (launch_pass16): a block's tiles are of one class
template <bool RLX>
__global__ void __launch_bounds__(NTT16_THREADS, 4) ntt_pass16_fwd_seq(i64 *dst, PassGeom g, ClassLists cl, int total, int tpb,
                                                                         const i64 *__restrict__ tw_br,
                                                                         const double *__restrict__ tw_dp,
                                                                         const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                                         const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    __shared__ i64 sm[NTT16_LDS_WORDS + 1];
    const int vb0 = (int)(blockIdx.x & 7) + 8 * (int)(blockIdx.x >> 3) * tpb;
    if (vb0 < cl.in_blocks) seq16_loop<false, RLX>(sm, dst, g, cl.in, vb0, cl.in_real, tpb, tw_br, tw_dp, ql, qh, kl, kh);
    else seq16_loop<true, RLX>(sm, dst, g, cl.dp, vb0 - cl.in_blocks, total - cl.in_blocks, tpb, tw_br, tw_dp, ql, qh, kl, kh);
}

// ---- the same tile with its NEXT tile's words requested early (seq16_loop_ws) ------------------------------------------------
// A block walks 8 tiles; the loads of tile i + 1 are issued when tile i's results have gone to the LDS staging span (its 32
// result registers are dead from there on) and fly while the wave reads the span back, stores the tile and sets up the next
// one.  The words stay UNASSEMBLED in their registers (an OR on arrival would put the wait right behind the loads): fp64 class
// low / mid planes (16 + 16 registers), integer class raw words.  A flagged wave (third plane, rare) reloads the slow way.
template <bool DP> struct TilePf;
template <> struct TilePf<true> {
    unsigned lo[16], mid[16];
};
template <> struct TilePf<false> {
    i64 raw[16];
};

template <bool DP, bool SKIP0>
__device__ __forceinline__ bool ws_prefetch_tile(const i64 *__restrict__ srow, const unsigned char *__restrict__ wf, int base, int E, int w,
                                                 TilePf<DP> &pf) {
    const int wave = __builtin_amdgcn_readfirstlane(w >> 6);
    const u64 *fp = reinterpret_cast<const u64 *>(wf + (wave << 4));
    const u64 fl = fp[0], fh = SKIP0 ? fp[0] : fp[1];
    if constexpr (DP) {
        const unsigned *lo = reinterpret_cast<const unsigned *>(srow) + base;
        const unsigned short *mid = reinterpret_cast<const unsigned short *>(srow + ((i64)1 << (E - 1))) + base;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            pf.lo[e] = __builtin_nontemporal_load(uniform_ptr(lo + (e << 8)) + (unsigned)w);
            pf.mid[e] = __builtin_nontemporal_load(uniform_ptr(mid + (e << 8)) + (unsigned)w);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) pf.raw[e] = __builtin_nontemporal_load(uniform_ptr(srow + base + (e << 8)) + (unsigned)w);
    }
    return (fl | fh) != 0;
}

// cur / cur_flagged: this tile's words as ws_prefetch_tile left them; next_*: where the block's next tile is (ntile < 0: none)
template <bool DP, bool SKIP0>
__device__ __forceinline__ void fwd_tile16_ws_pf(i64 *sm, TilePf<DP> &pf, bool &pf_flagged, const i64 *__restrict__ srow,
                                                  const unsigned char *__restrict__ wf, i64 *__restrict__ row, int tile,
                                                  const PassGeom &g, const Ctx &c, const Tw16Last<typename FwdArith<DP, false>::type> *lastC,
                                                  const i64 *__restrict__ nsrow, const unsigned char *__restrict__ nwf, int ntile) {
    const int w = lf_tid();
    const int base = tile << 12, E = g.logN, s = g.s0;
    i64 raw[16];
    const bool flagged = pf_flagged;
    if (flagged) {
        ws_load_tile<DP, SKIP0>(srow, wf, base, E, w, raw);   // third plane behind a raised flag: the complete load
    } else if constexpr (DP) {
#pragma unroll
        for (int e = 0; e < 16; ++e) raw[e] = (i64)(((u64)pf.mid[e] << 32) | (u64)pf.lo[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) raw[e] = pf.raw[e];
    }
    {
        int odd = 0;
        if (flagged) {
#pragma unroll
            for (int e = 0; e < 16; ++e) odd |= ((u64)raw[e] >= (u64)c.m.q2);
        }
        wave_flag_set16(sm, odd, w);
    }
    i64 o[16];
    bool ok;
    if constexpr (DP) {
        double x[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) x[e] = dp_from_word(raw[e]);
        ok = fwd_tile16_steps<ArithDp, true, false, SKIP0 ? 1 : 0>(reinterpret_cast<double *>(sm), sm, x, w, base, E, s, c, true, lastC);
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = dp_to_word(dp_reduce(x[e], c.d.q2, c.d.q2inv));
    } else {
        ok = fwd_tile16_steps<ArithInt<false>, false, false, SKIP0 ? 1 : 0>(sm, sm, raw, w, base, E, s, c, true, lastC);
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = raw[e];
    }
    if (!ok) {   // a word outside [0, 2q): nothing has been stored yet; the raw words once more, into LDS, then the generic tile
        __syncthreads();
        ws_load_tile<DP, SKIP0>(srow, wf, base, E, w, raw);
#pragma unroll
        for (int e = 0; e < 16; ++e) sm[PAD16(w + (e << 8))] = raw[e];
        __syncthreads();
        tile16_slow_lds(sm, row + base, base, s, E, false, c, SKIP0 ? 1 : 0);
        if (ntile >= 0) pf_flagged = ws_prefetch_tile<DP, SKIP0>(nsrow, nwf, ntile << 12, E, w, pf);
        else pf = TilePf<DP>{};
        return;
    }

    {
        i64 *sp = sm + 17 * w;
#pragma unroll
        for (int e = 0; e < 16; ++e) sp[e] = o[e];
    }
    // the next tile's words, requested once this tile's results sit in the staging span (their 32 registers are dead from here
    // on; requested right behind the register steps instead, 25 .. 29 registers spill).  A scheduling fence on both sides: hoisted
    // into the steps — the kernel's pressure peak — the prefetch registers would spill; defined on every path: left as they
    // were, the OLD words would stay live across the steps.
    __builtin_amdgcn_sched_barrier(0);
    if (ntile >= 0) pf_flagged = ws_prefetch_tile<DP, SKIP0>(nsrow, nwf, ntile << 12, E, w, pf);
    else pf = TilePf<DP>{};
    __builtin_amdgcn_sched_barrier(0);
    wave_lds_sync();
    const int L0 = ((w >> 6) << 10) + ((w & 63) << 1);
    const i64 *so = sm + PAD16(L0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        longlong2 v;
        v.x = so[i * 136];
        v.y = so[i * 136 + 1];
        if constexpr (NT_EXACT) nt_store2(row + base + L0 + (i << 7), v);
        else *reinterpret_cast<longlong2 *>(row + base + L0 + (i << 7)) = v;
    }
}

#ifndef NTT16_WS_PREFETCH
#define NTT16_WS_PREFETCH 1
#endif

// the same two kernels for a transform through a workspace (fwd_tile16_ws): ws -> dst, wflags = 64 flag bytes per (poly, limb)
template <bool DP, bool SKIP0>
__device__ __forceinline__ void seq16_loop_ws(i64 *sm, const i64 *ws, const unsigned char *wflags, i64 *dst, const PassGeom &g,
                                              const RowList &rl, int b0, int bend, int tpb, const i64 *__restrict__ tw_br,
                                              const double *__restrict__ tw_dp, const i64 *__restrict__ ql,
                                              const i64 *__restrict__ qh, const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    typedef typename FwdArith<DP, false>::type AC;
    const int w = lf_tid();
    Tw16Last<AC> last;
    int last_crow = -1, last_tile = -1;
#if NTT16_WS_PREFETCH
    TilePf<DP> pf;
    bool pf_flagged = false;
    TileAt t = tile_at(g, rl, b0, b0 < bend);
    if (!t.live) return;
    {
        const i64 ri = (i64)(t.poly * g.rows + t.crow);
        pf_flagged = ws_prefetch_tile<DP, SKIP0>(ws + (ri << g.logN), wflags + (ri << 6), t.tile << 12, g.logN, w, pf);
    }
    for (int i = 0; i < tpb; ++i) {
        const int bn = b0 + 8 * (i + 1);
        TileAt tn = tile_at(g, rl, bn, i + 1 < tpb && bn < bend);
        Ctx c;
        c.m = load_mod(ql, qh, kl, kh, t.crow);
        c.tw_mont = tw_br + ((i64)t.crow << g.logN);
        set_aux<DP>(c, tw_dp, t.crow, g.logN);
        c.d = DP ? make_dp_tab(c.m, c.tw_dp) : make_dp(c.m);
        c.relaxed = 0;
        c.inv_reduce = 0;
        if (t.crow != last_crow || t.tile != last_tile) {
            last.load(c, (1 << (g.s0 + 8)) + (((t.tile << 12) + 16 * w) >> (g.logN - g.s0 - 8)));
            last_crow = t.crow, last_tile = t.tile;
        }
        if (i) lds_barrier();
        const i64 ri = (i64)(t.poly * g.rows + t.crow), rn = (i64)(tn.poly * g.rows + tn.crow);
        fwd_tile16_ws_pf<DP, SKIP0>(sm, pf, pf_flagged, ws + (ri << g.logN), wflags + (ri << 6), dst + (ri << g.logN), t.tile, g, c, &last,
                                    ws + (rn << g.logN), wflags + (rn << 6), tn.live ? tn.tile : -1);
        if (!tn.live) break;
        t = tn;
    }
#else
    for (int i = 0; i < tpb; ++i) {
        const int b = b0 + 8 * i;
        const TileAt t = tile_at(g, rl, b, b < bend);
        if (!t.live) break;
        Ctx c;
        c.m = load_mod(ql, qh, kl, kh, t.crow);
        c.tw_mont = tw_br + ((i64)t.crow << g.logN);
        set_aux<DP>(c, tw_dp, t.crow, g.logN);
        c.d = DP ? make_dp_tab(c.m, c.tw_dp) : make_dp(c.m);
        c.relaxed = 0;
        c.inv_reduce = 0;
        if (t.crow != last_crow || t.tile != last_tile) {
            last.load(c, (1 << (g.s0 + 8)) + (((t.tile << 12) + 16 * w) >> (g.logN - g.s0 - 8)));
            last_crow = t.crow, last_tile = t.tile;
        }
        if (i) lds_barrier();
        const i64 ri = (i64)(t.poly * g.rows + t.crow);
        fwd_tile16_ws<DP, SKIP0>(sm, ws + (ri << g.logN), wflags + (ri << 6), dst + (ri << g.logN), t.tile, g, c, &last);
    }
#endif
}

template <bool SKIP0>
__global__ void __launch_bounds__(NTT16_THREADS, 4) ntt_pass16_fwd_seq_ws(const i64 *ws, const unsigned char *wflags, i64 *dst,
                                                                            PassGeom g, ClassLists cl, int total, int tpb,
                                                                            const i64 *__restrict__ tw_br,
                                                                            const double *__restrict__ tw_dp,
                                                                            const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                                            const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    __shared__ i64 sm[NTT16_LDS_WORDS + 1];
    const int vb0 = (int)(blockIdx.x & 7) + 8 * (int)(blockIdx.x >> 3) * tpb;
    if (vb0 < cl.in_blocks) seq16_loop_ws<false, SKIP0>(sm, ws, wflags, dst, g, cl.in, vb0, cl.in_real, tpb, tw_br, tw_dp, ql, qh, kl, kh);
    else seq16_loop_ws<true, SKIP0>(sm, ws, wflags, dst, g, cl.dp, vb0 - cl.in_blocks, total - cl.in_blocks, tpb, tw_br, tw_dp, ql, qh, kl, kh);
}

template <bool DP, bool SKIP0>
__device__ __forceinline__ void pass16_ws_body(i64 *sm, int b, const i64 *ws, const unsigned char *wflags, i64 *dst,
                                               const PassGeom &g, const RowList &rl, const i64 *__restrict__ tw_br,
                                               const double *__restrict__ tw_dp, const i64 *__restrict__ ql,
                                               const i64 *__restrict__ qh, const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    int poly, crow, tile;
    block_coords(g, rl, b, poly, crow, tile);
    poly = __builtin_amdgcn_readfirstlane(poly), crow = __builtin_amdgcn_readfirstlane(crow);
    tile = __builtin_amdgcn_readfirstlane(tile);
    Ctx c;
    c.m = load_mod(ql, qh, kl, kh, crow);
    c.tw_mont = tw_br + ((i64)crow << g.logN);
    set_aux<DP>(c, tw_dp, crow, g.logN);
    c.d = DP ? make_dp_tab(c.m, c.tw_dp) : make_dp(c.m);
    c.relaxed = 0;
    c.inv_reduce = 0;
    const i64 ri = (i64)(poly * g.rows + crow);
    fwd_tile16_ws<DP, SKIP0>(sm, ws + (ri << g.logN), wflags + (ri << 6), dst + (ri << g.logN), tile, g, c);
}

template <bool SKIP0>
__global__ void __launch_bounds__(NTT16_THREADS, 4) ntt_pass16_fwd_ws(const i64 *ws, const unsigned char *wflags, i64 *dst,
                                                                        PassGeom g, ClassLists cl, const i64 *__restrict__ tw_br,
                                                                        const double *__restrict__ tw_dp,
                                                                        const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                                        const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    __shared__ i64 sm[NTT16_LDS_WORDS + 1];
    const int b = blockIdx.x;
    if (b < cl.in_blocks) {
        if (b < cl.in_real) pass16_ws_body<false, SKIP0>(sm, b, ws, wflags, dst, g, cl.in, tw_br, tw_dp, ql, qh, kl, kh);
    } else {
        pass16_ws_body<true, SKIP0>(sm, b - cl.in_blocks, ws, wflags, dst, g, cl.dp, tw_br, tw_dp, ql, qh, kl, kh);
    }
}

// one arithmetic class per launch (used when a transform has a single class)
template <bool DP, bool RLX, bool INV>
__global__ void __launch_bounds__(NTT16_THREADS, 4) ntt_pass16(const i64 *src, i64 *dst, PassGeom g, RowList rl,
                                                                const i64 *__restrict__ tw_br, const double *__restrict__ tw_dp,
                                                                const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                                const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    __shared__ i64 sm[NTT16_LDS_WORDS + 1];
    pass16_body<DP, RLX, INV>(sm, blockIdx.x, src, dst, g, rl, tw_br, tw_dp, ql, qh, kl, kh);
}

template <bool DP>
__global__ void __launch_bounds__(NTT16_THREADS, 4) ntt_pass16_mul(const i64 *src, i64 *dst, PassGeom g, RowList rl, MulSrc ms,
                                                                    const i64 *__restrict__ tw_br, const double *__restrict__ tw_dp,
                                                                    const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                                    const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    __shared__ i64 sm[NTT16_LDS_WORDS + 1];
    pass16_body<DP, true, true, true>(sm, blockIdx.x, src, dst, g, rl, tw_br, tw_dp, ql, qh, kl, kh, &ms);
}

template <bool DP>
inline void launch_pass16_class(bool inverse, int relaxed, unsigned blocks, hipStream_t st, const i64 *src, i64 *dst,
                                const PassGeom &g, const RowList &rl, const i64 *tw_br, const double *tw_dp, const i64 *ql,
                                const i64 *qh, const i64 *kl, const i64 *kh, const MulSrc *ms) {
    const dim3 grid(blocks), block(NTT16_THREADS);
    if (ms) {
        hipLaunchKernelGGL((ntt_pass16_mul<DP>), grid, block, 0, st, src, dst, g, rl, *ms, tw_br, tw_dp, ql, qh, kl, kh);
    } else if (inverse) {
        if (relaxed) hipLaunchKernelGGL((ntt_pass16<DP, true, true>), grid, block, 0, st, src, dst, g, rl, tw_br, tw_dp, ql, qh, kl, kh);
        else hipLaunchKernelGGL((ntt_pass16<DP, false, true>), grid, block, 0, st, src, dst, g, rl, tw_br, tw_dp, ql, qh, kl, kh);
    } else {
        if (relaxed) hipLaunchKernelGGL((ntt_pass16<DP, true, false>), grid, block, 0, st, src, dst, g, rl, tw_br, tw_dp, ql, qh, kl, kh);
        else hipLaunchKernelGGL((ntt_pass16<DP, false, false>), grid, block, 0, st, src, dst, g, rl, tw_br, tw_dp, ql, qh, kl, kh);
    }
}

// tiles per block of ntt_pass16_fwd_seq, and the launch size (in such blocks) from which it is taken: four rounds of the
// 1 024 co-resident blocks (below that the tail of 8-tile blocks costs more than the kept twiddles save).  In-process A/B
// on the headline step (tools/ab_inproc.py): tiled pass -1.6 .. -2.4 % at 8 and 16 tiles per block, nothing at 4 or 32;
// keeping the 4 twiddles of the stage before as well (124 VGPRs) adds nothing.
#ifndef NTT16_SEQ_TILES
#define NTT16_SEQ_TILES 8
#endif
#define NTT16_SEQ_MIN_BLOCKS 4096

// host: the contiguous 12-stage pass of `polys` polynomials (forward: in place on dst; inverse: src -> dst, no tail).
// ms (relaxed inverse only): the pass transforms the product of two stacks, see MulSrc.
inline void launch_pass16(bool inverse, int relaxed, int polys, hipStream_t st, const i64 *src, i64 *dst, const PassGeom &g,
                          const RowList &in, const RowList &dp, const i64 *tw_br, const double *tw_dp, const i64 *ql,
                          const i64 *qh, const i64 *kl, const i64 *kh, const MulSrc *ms = nullptr, bool planes = false) {
    const unsigned per_row = (unsigned)polys << (g.logN - 12);
    const bool split = !(in.n && dp.n);   // a single class: its own instantiation (no register cost of the other)
    // (launches that do not fill the chip — silver's 128 .. 1 216 tile blocks — were also tried on the 512-thread /
    // 8-words-per-thread pass, two waves per SIMD from one block per CU: silver cc_mult 163-165 us against 165-169,
    // inside the run-to-run spread; dropped)
    if (split) {   // integer class first: its few, long blocks should not be the tail
        if (in.n) launch_pass16_class<false>(inverse, relaxed, per_row * (unsigned)in.n, st, src, dst, g, in, tw_br, tw_dp, ql, qh, kl, kh, ms);
        if (dp.n) launch_pass16_class<true>(inverse, relaxed, per_row * (unsigned)dp.n, st, src, dst, g, dp, tw_br, tw_dp, ql, qh, kl, kh, ms);
        return;
    }
    const ClassLists cl = class_lists(in, dp, per_row * (unsigned)in.n);
    const dim3 grid((unsigned)cl.in_blocks + per_row * (unsigned)dp.n), block(NTT16_THREADS);
    if (planes && ms) {   // relaxed, both classes present (the caller's condition: lf_stack_planes())
        hipLaunchKernelGGL(ntt_pass16_mul_planes, grid, block, 0, st, src, dst, g, cl, *ms, tw_br, tw_dp, ql, qh, kl, kh);
    } else if (planes && inverse) {
        hipLaunchKernelGGL(ntt_pass16_inv_planes, grid, block, 0, st, src, dst, g, cl, tw_br, tw_dp, ql, qh, kl, kh);
    } else if (planes) {   // forward, in place
        hipLaunchKernelGGL(ntt_pass16_fwd_planes, grid, block, 0, st, dst, g, cl, tw_br, tw_dp, ql, qh, kl, kh);
    } else if (ms) {
        hipLaunchKernelGGL(ntt_pass16_mul_mixed, grid, block, 0, st, src, dst, g, cl, *ms, tw_br, tw_dp, ql, qh, kl, kh);
    } else if (inverse) {
        if (relaxed) hipLaunchKernelGGL((ntt_pass16_mixed<true, true>), grid, block, 0, st, src, dst, g, cl, tw_br, tw_dp, ql, qh, kl, kh);
        else hipLaunchKernelGGL((ntt_pass16_mixed<false, true>), grid, block, 0, st, src, dst, g, cl, tw_br, tw_dp, ql, qh, kl, kh);
    } else {
        const int tpb = NTT16_SEQ_TILES;
        // (exact transforms only: on the relaxed passes of batched key switches — 100 k .. 400 k tiles — it measured neutral)
        if (tpb > 1 && !relaxed && (int)grid.x >= tpb * NTT16_SEQ_MIN_BLOCKS) {
            ClassLists cm = cl;   // classes padded to whole blocks of tpb tiles per XCD lane
            const unsigned unit = 8u * (unsigned)tpb, dpb = per_row * (unsigned)dp.n;
            cm.in_blocks = (int)(((unsigned)cl.in_real + unit - 1u) / unit * unit);
            const int total = cm.in_blocks + (int)dpb;
            const dim3 mg((unsigned)cm.in_blocks / (unsigned)tpb + 8u * ((dpb + unit - 1u) / unit));
            hipLaunchKernelGGL((ntt_pass16_fwd_seq<false>), mg, block, 0, st, dst, g, cm, total, tpb, tw_br, tw_dp, ql, qh, kl, kh);
            return;
        }
        if (relaxed) hipLaunchKernelGGL((ntt_pass16_mixed<true, false>), grid, block, 0, st, src, dst, g, cl, tw_br, tw_dp, ql, qh, kl, kh);
        else hipLaunchKernelGGL((ntt_pass16_mixed<false, false>), grid, block, 0, st, src, dst, g, cl, tw_br, tw_dp, ql, qh, kl, kh);
    }
}

// host: the contiguous exact forward pass of a transform through a workspace (either class list may be empty)
// skip0: the column pass in front took the tile's first stage too
inline void launch_pass16_ws(int polys, hipStream_t st, const i64 *ws, const unsigned char *wflags, i64 *dst, const PassGeom &g,
                             const RowList &in, const RowList &dp, const i64 *tw_br, const double *tw_dp, const i64 *ql,
                             const i64 *qh, const i64 *kl, const i64 *kh, bool skip0) {
    const unsigned per_row = (unsigned)polys << (g.logN - 12);
    const ClassLists cl = class_lists(in, dp, per_row * (unsigned)in.n);
    const dim3 grid((unsigned)cl.in_blocks + per_row * (unsigned)dp.n), block(NTT16_THREADS);
    const int tpb = NTT16_SEQ_TILES;
    if (tpb > 1 && (int)grid.x >= tpb * NTT16_SEQ_MIN_BLOCKS) {
        ClassLists cm = cl;
        const unsigned unit = 8u * (unsigned)tpb, dpb = per_row * (unsigned)dp.n;
        cm.in_blocks = (int)(((unsigned)cl.in_real + unit - 1u) / unit * unit);
        const int total = cm.in_blocks + (int)dpb;
        const dim3 mg((unsigned)cm.in_blocks / (unsigned)tpb + 8u * ((dpb + unit - 1u) / unit));
        if (skip0) hipLaunchKernelGGL(ntt_pass16_fwd_seq_ws<true>, mg, block, 0, st, ws, wflags, dst, g, cm, total, tpb, tw_br, tw_dp, ql, qh, kl, kh);
        else hipLaunchKernelGGL(ntt_pass16_fwd_seq_ws<false>, mg, block, 0, st, ws, wflags, dst, g, cm, total, tpb, tw_br, tw_dp, ql, qh, kl, kh);
        return;
    }
    if (skip0) hipLaunchKernelGGL(ntt_pass16_fwd_ws<true>, grid, block, 0, st, ws, wflags, dst, g, cl, tw_br, tw_dp, ql, qh, kl, kh);
    else hipLaunchKernelGGL(ntt_pass16_fwd_ws<false>, grid, block, 0, st, ws, wflags, dst, g, cl, tw_br, tw_dp, ql, qh, kl, kh);
}

}  // namespace
