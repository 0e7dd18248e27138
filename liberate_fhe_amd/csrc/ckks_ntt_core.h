// ckks_ntt_core.h — device-side NTT machinery shared by ckks_ntt.hip (lf_ntt / lf_intt) and ckks_ks.hip
// (fused key switch): tile geometry, the two arithmetic classes, radix-8 register steps, pass kernels.
// See the header comment of ckks_ntt.hip for the design.
#pragma once
#include "ckks_common.h"
#include <stdlib.h>

#define NTT_THREADS 512
#define NTT_PASS_WAVES 6   // waves per SIMD the mixed pass kernels are compiled for (80 VGPRs = 3 blocks per CU; 4 and 8
                           // were measured slower)
#define NTT_LDS_WORDS ((1 << NTT_TILE_LOG_MAX) + (1 << (NTT_TILE_LOG_MAX - 3)))
#define TAIL_NONE (-1)
#define SMALL_PRIME_LIMIT (1ull << 41)

#define PAD(L) ((L) + ((L) >> 3))
// PAD(p + (e << LOGDL)) for LOGDL >= 3 and a p whose bits LOGDL .. LOGDL + K - 1 are clear: the element stride of the
// padded layout is a compile-time constant, so the 2^K accesses of a register step share ONE address register and
// differ in the instruction's immediate offset (the compiler does not derive this from the shift / or form:
// it spent three VALU instructions per LDS access on the address)
#define PAD_STRIDE(LOGDL) ((1 << (LOGDL)) + ((1 << (LOGDL)) >> 3))
#define NTT_FLAG_WORD NTT_LDS_WORDS          // two flag words live behind the tile in the same LDS array

namespace {

// Thread index rebuilt from the wave's index (one SGPR) and the lane id (two v_mbcnt): nothing has to stay in a
// VGPR across the out-of-line calls of the exact fp64 path, where the compiler otherwise spills it (8 bytes per
// thread and tile = +10 % HBM write traffic on the tiled pass, measured with the WRITE_SIZE counter).
__device__ __forceinline__ int lf_tid() {
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    return (wave << 6) | lane;
}

// Workgroup barrier that waits for LDS traffic only.  __syncthreads() also drains the vector-memory
// queue (s_waitcnt vmcnt(0)), which would serialise the register prefetch of the next tile behind
// every barrier of the current one.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// block-wide OR through an LDS flag word (double-buffered by `parity`, so no reset barrier is needed)
__device__ __forceinline__ bool block_or(i64 *sm, int pred, int parity) {
    int *flags = reinterpret_cast<int *>(sm + NTT_FLAG_WORD);
    if (threadIdx.x == 0) flags[parity ^ 1] = 0;
    if (__builtin_amdgcn_ballot_w64(pred != 0) != 0 && (threadIdx.x & 63) == 0) flags[parity] = 1;
    lds_barrier();
    return flags[parity] != 0;
}

struct RowList {
    int n;
    unsigned short id[MAX_LIST_ROWS];
};

struct PassGeom {
    int logN;
    int tl;        // log2 of tile size
    int strided;   // 0 contiguous, 1 strided
    int S;         // stages in this pass
    int s0;        // global index of the first stage of this pass
    int logC;      // strided: log2 of columns per tile row
    int rows;      // limbs per polynomial
    int batch;     // polynomials
    int relaxed;   // 1: result only needed modulo q
    int last;      // 1: last pass of the transform
    int plain;     // relaxed only: fp64-class limbs work in the PLAIN domain — no Montgomery entry on the way in
                   // (Rs applies to integer-class limbs only), inverse tail multiplies by N^-1 instead of N^-1 R^-1
    // key switch inside cc_mult (lf_relin_*): polynomial `poly` of the stack is digit skip_off + poly % skip_mod; the limbs
    // a digit is made of need no extension and no transform (the caller has them in the NTT domain already): their blocks exit
    const unsigned char *skip_own;   // device, [rows]: digit that owns the limb (255: none); nullptr: nothing is skipped
    int skip_mod, skip_off;
    // relaxed column passes of the fused ops: fp64-class rows of the stack are PLANES (u32 low[N] at byte 0 of the row's 8 N bytes,
    // u16 high[N] at byte 4 N: canonical residues, 6 bytes per word) on the way in (bit 0) / on the way out (bit 1).  A pass
    // that changes the format cannot run in place (a raw word covers the low plane of two others).
    int pln;
    const i64 *pln_src;   // inverse column pass with PLN_IN: the stack it reads (same geometry as the one it writes); nullptr: in place
};
#define PLN_IN 1
#define PLN_OUT 2

// tile-local index -> coefficient index of the row
__device__ __forceinline__ int tile_gaddr(const PassGeom &g, int tile, int L) {
    if (!g.strided) return (tile << g.tl) + L;
    const int r = L >> g.logC, c = L & ((1 << g.logC) - 1);
    return (r << (g.logN - g.S)) + (tile << g.logC) + c;
}

// block id -> (polynomial, limb, tile): the `batch` polynomials of one (limb, tile) pair are
// consecutive on one XCD (block b runs on XCD b % 8) so they share the pair's twiddles in that L2.
__device__ __forceinline__ void block_coords(const PassGeom &g, const RowList &rl, int b, int &poly, int &limb, int &tile) {
    const int tiles = 1 << (g.logN - g.tl);
    const int pairs = rl.n * tiles;
    int pair;
    if ((pairs & 7) == 0) {
        const int x = b & 7, r = b >> 3;
        poly = r % g.batch;
        pair = (r / g.batch) * 8 + x;
    } else {
        pair = b / g.batch;
        poly = b % g.batch;
    }
    limb = rl.id[pair / tiles];
    tile = pair % tiles;
}

#define NTT_PRE ((1 << NTT_TILE_LOG_MAX) / (NTT_THREADS * 2))   // 16-byte prefetch registers per thread

// issue the global loads of one tile into registers (16 B per lane); consumed by stash_tile()
__device__ __forceinline__ void prefetch_tile(longlong2 (&pre)[NTT_PRE], const i64 *row, const PassGeom &g, int tile) {
    const int T = 1 << g.tl;
#pragma unroll
    for (int v = 0; v < NTT_PRE; ++v) {
        const int L = (threadIdx.x + v * NTT_THREADS) * 2;
        if (L < T) {
            const i64 *src = row + tile_gaddr(g, tile, L);
            pre[v] = *reinterpret_cast<const longlong2 *>(src);
        }
    }
}

// prefetched words -> LDS (padded layout); returns whether any word lies outside [0, 2q)
__device__ __forceinline__ int stash_tile(i64 *sm, const longlong2 (&pre)[NTT_PRE], const PassGeom &g, i64 q2) {
    const int T = 1 << g.tl;
    int odd_word = 0;
#pragma unroll
    for (int v = 0; v < NTT_PRE; ++v) {
        const int L = (threadIdx.x + v * NTT_THREADS) * 2;
        if (L < T) {
            longlong2 w = pre[v];
            if (g.relaxed) {   // residues only: fold the signed-lazy words into [0, 2q)
                w.x = w.x < 0 ? w.x + q2 : w.x;
                w.y = w.y < 0 ? w.y + q2 : w.y;
            }
            odd_word |= ((u64)w.x >= (u64)q2) | ((u64)w.y >= (u64)q2);
            sm[PAD(L)] = w.x;
            sm[PAD(L + 1)] = w.y;
        }
    }
    return odd_word;
}

// ------------------------------------------------------------------------------------------------
// fp64 modular arithmetic for q < 2^41 (operands: non-negative integers held exactly in doubles)
// ------------------------------------------------------------------------------------------------
struct RowDp {
    double q, q2, qinv, q2inv;
};

// r < 0 ? r + m : r in two fp64 instructions (|r| < 2^53).
// (floor(r * 2^-64) as the indicator costs three; masking m with the sign bit four: both measured slower.)
__device__ __forceinline__ double dp_addmask(double r, double m) {
    // r is an INTEGER-valued double: r < 0 means r <= -1, so clamp(-r) to [0, 1] (the VOP3 output modifier) IS the
    // indicator [r < 0] — one instruction instead of the multiply + floor pair (-0.0 never occurs: a zero sum of
    // two opposite doubles is +0.0 under round-to-nearest)
    // (fmin(fmax(x, 0), 1) is the form the compiler folds into the clamp bit: v_max_f64 neg, -r, -r clamp)
    const double neg = __builtin_fmin(__builtin_fmax(-r, 0.0), 1.0);
    return __builtin_fma(neg, m, r);
}

// (a * w) mod q, canonical, for a < 2^52, w < q: exact via the FMA low part.
__device__ __forceinline__ double dp_mulmod(double a, double w, const RowDp &m) {
    const double hi = a * w;
    const double lo = __builtin_fma(a, w, -hi);
    const double quo = __builtin_rint(hi * m.qinv);
    const double r = __builtin_fma(-quo, m.q, hi) + lo;
    return dp_addmask(r, m.q);
}

// (a * w) mod q as a BALANCED residue in [-q/2, q/2] (no sign fix-up): 6 fp64 ops.  |a| < 2^52.
__device__ __forceinline__ double dp_mulmod_bal(double a, double w, const RowDp &m) {
    const double hi = a * w;
    const double lo = __builtin_fma(a, w, -hi);
    const double quo = __builtin_rint(hi * m.qinv);
    return __builtin_fma(-quo, m.q, hi) + lo;
}

// x mod q as a balanced residue, |x| < 2^52
__device__ __forceinline__ double dp_reduce_bal(double x, const RowDp &m) {
    return __builtin_fma(-__builtin_rint(x * m.qinv), m.q, x);
}

// x mod m in [0, m) for an integer x (either sign) with |x| < 64 m, m = q or 2q < 2^43, minv its reciprocal: three
// fp64 instructions.  floor(x * minv + 2^-44) IS floor(x / m): the fma's total error is below 2^-46 for quotients
// under 64, the bias lifts exact multiples of m over their integer, and a remainder of m - 1 still stays 2^-43 short
// of the next one.  (rint + fma + the three-instruction sign fix cost six.)
#define DP_FLOOR_BIAS 5.684341886080802e-14   // 2^-44
__device__ __forceinline__ double dp_reduce(double x, double m, double minv) {
    const double quo = __builtin_floor(__builtin_fma(x, minv, DP_FLOOR_BIAS));
    return __builtin_fma(-quo, m, x);
}

// v < 2^22 (> (2q)^2 / 2^62 for q < 2^41: below it the lazy REDC word may be T0 + q) for a canonical v >= 0, as one 32-bit compare on the high word (an fp64 compare costs twice as much);
// -0.0 counts as below, like +0.0
__device__ __forceinline__ bool dp_below_fix_limit(double v) { return __double2hiint(v) < 0x41500000; }

// The reference's lazy representative of REDC62(A*B) given its canonical value t0 (< 2^22):
// t0 + q iff t0 * 2^62 < A*B (integer operands A, B as the reference multiplies them).
__device__ __noinline__ double dp_lazy_fix(double t0, u64 A, u64 B, double q) {
    const u128 x = (u128)A * (u128)B;
    const u64 xh = (u64)(x >> 62);
    const u64 xl = (u64)x & M62;
    const u64 t = (u64)t0;
    return ((xh > t) || (xh == t && xl != 0)) ? t0 + q : t0;
}

// CNT consecutive table entries starting at idx0 (a multiple of CNT): the twiddles of one stage inside a
// radix-2^K step are neighbours in the bit-reversed table, so they come in as 16-byte loads.
template <int MAXC, class E>
__device__ __forceinline__ void load_group(const E *tab, int idx0, int cnt, E (&w)[MAXC]) {
    static_assert(sizeof(E) == 8, "8-byte table entries");
    if (cnt == 1) {
        w[0] = tab[(unsigned)idx0];
    } else {
        const longlong2 *src = reinterpret_cast<const longlong2 *>(tab + (unsigned)idx0);
#pragma unroll
        for (int i = 0; i < MAXC / 2; ++i) {
            if (2 * i >= cnt) break;
            const longlong2 v = src[i];
            w[2 * i] = *reinterpret_cast<const E *>(&v.x);
            w[2 * i + 1] = *reinterpret_cast<const E *>(&v.y);
        }
    }
}

struct Ctx {
    RowMod m;
    RowDp d;
    const i64 *tw_mont;     // compact Montgomery twiddles of this limb (always valid)
    const double *tw_dp;    // compact plain twiddles as doubles (fp64 class)
    const ShoupW *tw_sh;    // integer class, relaxed transforms: (quotient, plain twiddle) pairs — the same auxiliary row
    int relaxed;
    int inv_reduce;         // fp64 inverse steps: reduce mod 2q at the end of this step
};

// The auxiliary twiddle table (psi_dp / ipsi_dp of the C ABI, built by lf_twiddle_dp): one row of 2N 8-byte words per
// limb.  fp64 class: words [0, N) are the plain twiddles as doubles (slot 0: 1/q).  Integer class: N pairs
// (quotient, plain twiddle) for the Shoup products of the relaxed transforms.
template <bool DP>
__device__ __forceinline__ void set_aux(Ctx &c, const double *aux, int crow, int logN) {
    const double *row = aux ? aux + ((i64)crow << (logN + 1)) : nullptr;
    c.tw_dp = DP ? row : nullptr;
    c.tw_sh = DP ? nullptr : reinterpret_cast<const ShoupW *>(row);
}

// ------------------------------------------------------------------------------------------------
// Arithmetic policies
// ------------------------------------------------------------------------------------------------
template <bool SIGNED>
struct ArithInt {
    typedef i64 T;
    typedef i64 W;
    static __device__ __forceinline__ W tw(const Ctx &c, int idx) { return c.tw_mont[(unsigned)idx]; }
    template <int MAXC>
    static __device__ __forceinline__ void tw_group(const Ctx &c, int idx0, int cnt, W (&w)[MAXC]) { load_group<MAXC>(c.tw_mont, idx0, cnt, w); }
    static __device__ __forceinline__ T mul(const Ctx &c, W S, T O) {
        return SIGNED ? mm62s(S, O, c.m.q, c.m.k) : mm62u((u64)S, (u64)O, c.m.q, c.m.k);
    }
    // (unsigned form — every word in [0, 2q) — : the same values from the borrow-select helpers, 4 instructions fewer per butterfly)
    static __device__ __forceinline__ void fwd(const Ctx &c, T &a, T &b, W S, int) {
        const T U = a, V = mul(c, S, b);
        if constexpr (SIGNED) {
            a = csub(U + V, c.m.q2);
            b = csub(U + c.m.q2 - V, c.m.q2);
        } else {
            a = (T)csub_u((u64)U + (u64)V, (u64)c.m.q2);
            b = (T)sub_lazy_u((u64)U, (u64)V, (u64)c.m.q2);
        }
    }
    static __device__ __forceinline__ void inv(const Ctx &c, T &a, T &b, W S, int) {
        const T U = a, V = b;
        if constexpr (SIGNED) {
            const T O = csub(U + c.m.q2 - V, c.m.q2);
            b = mul(c, S, O);
            a = csub(U + V, c.m.q2);
        } else {
            const T O = (T)sub_lazy_u((u64)U, (u64)V, (u64)c.m.q2);
            b = mul(c, S, O);
            a = (T)csub_u((u64)U + (u64)V, (u64)c.m.q2);
        }
    }
    template <int NN>
    static __device__ __forceinline__ void fwd_end(const Ctx &, T (&)[NN]) {}
    template <int NN>
    static __device__ __forceinline__ void inv_end(const Ctx &, T (&)[NN]) {}
};

// Integer class, RELAXED: Shoup products (ckks_common.h) on lazy 64-bit words.  Invariant: every word < 8q (< 2^63).
//   forward:  a' = U + V < 12q, b' = U + 4q - V < 12q, each folded once by 8q;
//   inverse:  a' = U + V < 16q folded by 8q, b' = (U + 8q - V) * w < 4q.
// Words leave a pass through canon() as canonical residues.
struct ArithShoup {
    typedef i64 T;
    typedef ShoupW W;
    static __device__ __forceinline__ W tw(const Ctx &c, int idx) { return c.tw_sh[(unsigned)idx]; }
    template <int MAXC>
    static __device__ __forceinline__ void tw_group(const Ctx &c, int idx0, int cnt, W (&w)[MAXC]) {
#pragma unroll
        for (int i = 0; i < MAXC; ++i)
            if (i < cnt) w[i] = c.tw_sh[(unsigned)(idx0 + i)];
    }
    static __device__ __forceinline__ void fwd(const Ctx &c, T &a, T &b, W w, int) {
        const u64 U = (u64)a, V = shoup_mul((u64)b, w, c.m.q), q8 = c.m.q << 3;
        a = (T)csub_u(U + V, q8);
        b = (T)csub_u(U + (c.m.q << 2) - V, q8);
    }
    static __device__ __forceinline__ void inv(const Ctx &c, T &a, T &b, W w, int) {
        const u64 U = (u64)a, V = (u64)b, q8 = c.m.q << 3;
        a = (T)csub_u(U + V, q8);
        b = (T)shoup_mul(U + q8 - V, w, c.m.q);
    }
    template <int NN>
    static __device__ __forceinline__ void fwd_end(const Ctx &, T (&)[NN]) {}
    template <int NN>
    static __device__ __forceinline__ void inv_end(const Ctx &, T (&)[NN]) {}
    static __device__ __forceinline__ T canon(const Ctx &c, T x) { return (T)shoup_canon((u64)x, c.m.q); }
};

// fp64 class.  Words are representatives (< 2^52) of the reference's lazy values mod 2q.  Forward sums
// grow by at most 2q per stage and are reduced mod 2q once, when the pass stores its tile; inverse sums
// double per stage and are reduced at the end of every second step.
struct ArithDp {
    typedef double T;
    typedef double W;
    static __device__ __forceinline__ W tw(const Ctx &c, int idx) { return c.tw_dp[(unsigned)idx]; }
    template <int MAXC>
    static __device__ __forceinline__ void tw_group(const Ctx &c, int idx0, int cnt, W (&w)[MAXC]) { load_group<MAXC>(c.tw_dp, idx0, cnt, w); }
    // lazy REDC62(S * O) for O = o (any representative < 2^52 of the lazy word mod 2q)
    static __device__ __forceinline__ T mul(const Ctx &c, W w, T o, int idx) {
        T v = dp_mulmod(o, w, c.d);
        if (!c.relaxed && dp_below_fix_limit(v))
            v = dp_lazy_fix(v, (u64)c.tw_mont[idx], (u64)dp_reduce(o, c.d.q2, c.d.q2inv), c.d.q);
        return v;
    }
    static __device__ __forceinline__ void fwd(const Ctx &c, T &a, T &b, W w, int idx) {
        const T U = a, V = mul(c, w, b, idx);      // V in [0, 2q)
        a = U + V;
        b = U - V;                                  // representatives may be negative: only the class mod 2q counts
    }
    static __device__ __forceinline__ void inv(const Ctx &c, T &a, T &b, W w, int idx) {
        const T U = a, V = b;
        b = mul(c, w, U - V, idx);
        a = U + V;
    }
    template <int NN>
    static __device__ __forceinline__ void fwd_end(const Ctx &, T (&)[NN]) {}
    template <int NN>
    static __device__ __forceinline__ void inv_end(const Ctx &c, T (&x)[NN]) {
        if (c.inv_reduce) {
#pragma unroll
            for (int e = 0; e < NN; ++e) x[e] = dp_reduce(x[e], c.d.q2, c.d.q2inv);
        }
    }
};

// fp64 class, RELAXED: only the residue mod q matters, so products stay balanced (no sign fix-up), sums
// and differences are plain fp64 adds on signed words, and nothing is compared or subtracted per
// butterfly: 8 fp64 instructions per butterfly instead of 13.  Forward words grow by q/2 per stage;
// inverse words double per stage and are folded back to balanced residues every second step.
struct ArithDpR {
    typedef double T;
    typedef double W;
    static __device__ __forceinline__ W tw(const Ctx &c, int idx) { return c.tw_dp[(unsigned)idx]; }
    template <int MAXC>
    static __device__ __forceinline__ void tw_group(const Ctx &c, int idx0, int cnt, W (&w)[MAXC]) { load_group<MAXC>(c.tw_dp, idx0, cnt, w); }
    static __device__ __forceinline__ void fwd(const Ctx &c, T &a, T &b, W w, int) {
        const T U = a, V = dp_mulmod_bal(b, w, c.d);
        a = U + V;
        b = U - V;
    }
    static __device__ __forceinline__ void inv(const Ctx &c, T &a, T &b, W w, int) {
        const T U = a, V = b;
        b = dp_mulmod_bal(U - V, w, c.d);
        a = U + V;
    }
    template <int NN>
    static __device__ __forceinline__ void fwd_end(const Ctx &, T (&)[NN]) {}
    template <int NN>
    static __device__ __forceinline__ void inv_end(const Ctx &c, T (&x)[NN]) {
        if (c.inv_reduce) {
#pragma unroll
            for (int e = 0; e < NN; ++e) x[e] = dp_reduce_bal(x[e], c.d);
        }
    }
};

// Forward radix-2^K step over local distances (dl << (K-1)), ..., dl at stages s, s+1, ..
//   twiddle index of stage st at tile-local index L: (1 << st) + ((base + L) >> (E - st));
//   the 2^u groups of stage u each share one twiddle.
template <int K, class A, int LOGDL = -1>
__device__ __forceinline__ void fwd_step(typename A::T *sm, int T, int log_dl_rt, int s, int E, int base, const Ctx &c) {
    const int log_dl = LOGDL >= 0 ? LOGDL : log_dl_rt;   // compile-time in the hot 4096-word schedules
    const int items = T >> K;
    for (int w = threadIdx.x; w < items; w += NTT_THREADS) {
        const int p = ((w >> log_dl) << (log_dl + K)) | (w & ((1 << log_dl) - 1));
        typename A::T x[1 << K];
        const int pb = PAD(p);
#pragma unroll
        for (int e = 0; e < (1 << K); ++e) x[e] = LOGDL >= 3 ? sm[pb + e * PAD_STRIDE(LOGDL >= 3 ? LOGDL : 3)] : sm[PAD(p + (e << log_dl))];
        // stage u uses 2^u twiddles, entries (i0 << u) .. (i0 << u) + 2^u - 1 of the bit-reversed table
        const int i0 = (1 << s) + ((base + p) >> (E - s));
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const int du = 1 << (K - 1 - u);
            typename A::W wv[K > 1 ? (1 << (K - 1)) : 1];
            A::tw_group(c, i0 << u, 1 << u, wv);
#pragma unroll
            for (int j = 0; j < (1 << u); ++j) {
                const int e0 = j << (K - u);
#pragma unroll
                for (int t = 0; t < du; ++t) A::fwd(c, x[e0 + t], x[e0 + t + du], wv[j], (i0 << u) + j);
            }
        }
        A::fwd_end(c, x);
#pragma unroll
        for (int e = 0; e < (1 << K); ++e) {
            if (LOGDL >= 3) sm[pb + e * PAD_STRIDE(LOGDL >= 3 ? LOGDL : 3)] = x[e];
            else sm[PAD(p + (e << log_dl))] = x[e];
        }
    }
}

// The radix-8 step of the 4096-word schedules with its 7 twiddles passed in: the caller loads the NEXT
// step's twiddles before the barrier that ends the current one, so their L2 latency hides behind it.
template <class A, int LOGDL>
struct StepTw {
    typename A::W w0[1], w1[2], w2[4];
    int i0;
    __device__ __forceinline__ void load(const Ctx &c, int s, int E, int base, int w) {
        const int p = ((w >> LOGDL) << (LOGDL + 3)) | (w & ((1 << LOGDL) - 1));
        i0 = (1 << s) + ((base + p) >> (E - s));
        A::tw_group(c, i0, 1, w0);
        A::tw_group(c, i0 << 1, 2, w1);
        A::tw_group(c, i0 << 2, 4, w2);
    }
};

// the three stages of a forward radix-8 step on 8 words held in registers (x[e] = word at distance e << LOGDL)
template <class A, int LOGDL>
__device__ __forceinline__ void fwd_regs8(typename A::T (&x)[8], const StepTw<A, LOGDL> &tw, const Ctx &c) {
#pragma unroll
    for (int t = 0; t < 4; ++t) A::fwd(c, x[t], x[t + 4], tw.w0[0], tw.i0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t) A::fwd(c, x[4 * j + t], x[4 * j + t + 2], tw.w1[j], (tw.i0 << 1) + j);
#pragma unroll
    for (int j = 0; j < 4; ++j) A::fwd(c, x[2 * j], x[2 * j + 1], tw.w2[j], (tw.i0 << 2) + j);
    A::fwd_end(c, x);
}

template <class A, int LOGDL>
__device__ __forceinline__ void fwd_step8(typename A::T *sm, const StepTw<A, LOGDL> &tw, const Ctx &c, int w) {
    const int p = ((w >> LOGDL) << (LOGDL + 3)) | (w & ((1 << LOGDL) - 1));
    typename A::T x[8];
    typename A::T *sp = sm + PAD(p);   // LOGDL = 0: p = 8 w, the 8 words are consecutive (PAD(8 w + e) = 9 w + e)
    constexpr int st = LOGDL >= 3 ? PAD_STRIDE(LOGDL >= 3 ? LOGDL : 3) : 1;
    static_assert(LOGDL >= 3 || LOGDL == 0, "padded stride of a radix-8 step");
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = sp[e * st];
    fwd_regs8<A, LOGDL>(x, tw, c);
#pragma unroll
    for (int e = 0; e < 8; ++e) sp[e * st] = x[e];
}

// Inverse radix-2^K step over local distances dl, 2dl, .. at stages s, s+1, ..
//   twiddle index: (N >> (st+1)) + ((base + L) >> (st + 1 - adj))
template <int K, class A, int LOGDL = -1>
__device__ __forceinline__ void inv_step(typename A::T *sm, int T, int log_dl_rt, int s, int adj, int logN, int base,
                                         const Ctx &c) {
    const int log_dl = LOGDL >= 0 ? LOGDL : log_dl_rt;
    const int items = T >> K;
    for (int w = threadIdx.x; w < items; w += NTT_THREADS) {
        const int p = ((w >> log_dl) << (log_dl + K)) | (w & ((1 << log_dl) - 1));
        typename A::T x[1 << K];
        const int pb = PAD(p);
#pragma unroll
        for (int e = 0; e < (1 << K); ++e) x[e] = LOGDL >= 3 ? sm[pb + e * PAD_STRIDE(LOGDL >= 3 ? LOGDL : 3)] : sm[PAD(p + (e << log_dl))];
        // stage u uses 2^(K-1-u) twiddles, entries (il << (K-1-u)) .. of the bit-reversed table
        const int il = (1 << (logN - s - K)) + ((base + p) >> (s + K - adj));
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const int du = 1 << u;
            typename A::W wv[K > 1 ? (1 << (K - 1)) : 1];
            A::tw_group(c, il << (K - 1 - u), 1 << (K - 1 - u), wv);
#pragma unroll
            for (int h = 0; h < (1 << (K - 1 - u)); ++h) {
                const int e0 = h << (u + 1);
#pragma unroll
                for (int t = 0; t < du; ++t) A::inv(c, x[e0 + t], x[e0 + t + du], wv[h], (il << (K - 1 - u)) + h);
            }
        }
        A::inv_end(c, x);
#pragma unroll
        for (int e = 0; e < (1 << K); ++e) {
            if (LOGDL >= 3) sm[pb + e * PAD_STRIDE(LOGDL >= 3 ? LOGDL : 3)] = x[e];
            else sm[PAD(p + (e << log_dl))] = x[e];
        }
    }
}

// inverse counterpart: stage u of the radix-8 step uses 2^(2-u) twiddles, entries (il << (2-u)) + h
template <class A, int LOGDL>
struct StepTwInv {
    typename A::W w0[4], w1[2], w2[1];
    int il;
    __device__ __forceinline__ void load(const Ctx &c, int s, int adj, int logN, int base, int w) {
        const int p = ((w >> LOGDL) << (LOGDL + 3)) | (w & ((1 << LOGDL) - 1));
        il = (1 << (logN - s - 3)) + ((base + p) >> (s + 3 - adj));
        A::tw_group(c, il << 2, 4, w0);
        A::tw_group(c, il << 1, 2, w1);
        A::tw_group(c, il, 1, w2);
    }
};

// the three stages of an inverse radix-8 step on 8 words held in registers
template <class A, int LOGDL>
__device__ __forceinline__ void inv_regs8(typename A::T (&x)[8], const StepTwInv<A, LOGDL> &tw, const Ctx &c) {
#pragma unroll
    for (int h = 0; h < 4; ++h) A::inv(c, x[2 * h], x[2 * h + 1], tw.w0[h], (tw.il << 2) + h);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t = 0; t < 2; ++t) A::inv(c, x[4 * h + t], x[4 * h + t + 2], tw.w1[h], (tw.il << 1) + h);
#pragma unroll
    for (int t = 0; t < 4; ++t) A::inv(c, x[t], x[t + 4], tw.w2[0], tw.il);
    A::inv_end(c, x);
}

template <class A, int LOGDL>
__device__ __forceinline__ void inv_step8(typename A::T *sm, const StepTwInv<A, LOGDL> &tw, const Ctx &c, int w) {
    const int p = ((w >> LOGDL) << (LOGDL + 3)) | (w & ((1 << LOGDL) - 1));
    typename A::T x[8];
    typename A::T *sp = sm + PAD(p);
    constexpr int st = LOGDL >= 3 ? PAD_STRIDE(LOGDL >= 3 ? LOGDL : 3) : 1;
    static_assert(LOGDL >= 3 || LOGDL == 0, "padded stride of a radix-8 step");
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = sp[e * st];
    inv_regs8<A, LOGDL>(x, tw, c);
#pragma unroll
    for (int e = 0; e < 8; ++e) sp[e * st] = x[e];
}

template <class A, bool FAST>
__device__ __forceinline__ void run_fwd_stages(typename A::T *sm, const PassGeom &g, int tile, const Ctx &c) {
    const int T = 1 << g.tl;
    const int E = g.strided ? g.tl : g.logN;
    const int base = g.strided ? 0 : (tile << g.tl);
    int s = g.s0, left = g.S, log_d = g.tl - 1;
    if (FAST && g.tl == 12 && g.S == 12) {   // contiguous 4096-word pass: distances known at compile time
        StepTw<A, 9> t9;
        StepTw<A, 6> t6;
        StepTw<A, 3> t3;
        StepTw<A, 0> t0;
        t9.load(c, s, E, base, lf_tid());
        fwd_step8<A, 9>(sm, t9, c, lf_tid());
        t6.load(c, s + 3, E, base, lf_tid());
        lds_barrier();
        fwd_step8<A, 6>(sm, t6, c, lf_tid());
        t3.load(c, s + 6, E, base, lf_tid());
        lds_barrier();
        fwd_step8<A, 3>(sm, t3, c, lf_tid());
        t0.load(c, s + 9, E, base, lf_tid());
        lds_barrier();
        fwd_step8<A, 0>(sm, t0, c, lf_tid());
        lds_barrier();
        return;
    }
    if (FAST && g.tl == 12 && left >= 3) {   // strided pass of a 4096-word tile: first step at distance 2^11
        fwd_step<3, A, 9>(sm, T, 9, s, E, base, c); lds_barrier();
        s += 3; left -= 3; log_d -= 3;
        if (left == 1) {   // logN 16: the fourth strided stage, distance 2^8 known at compile time
            fwd_step<1, A, 8>(sm, T, 8, s, E, base, c); lds_barrier();
            return;
        }
    }
    while (left > 0) {
        if (FAST && left >= 3) {
            fwd_step<3, A>(sm, T, log_d - 2, s, E, base, c);
            s += 3; left -= 3; log_d -= 3;
        } else if (FAST && left == 2) {
            fwd_step<2, A>(sm, T, log_d - 1, s, E, base, c);
            s += 2; left -= 2; log_d -= 2;
        } else {
            fwd_step<1, A>(sm, T, log_d, s, E, base, c);
            s += 1; left -= 1; log_d -= 1;
        }
        lds_barrier();
    }
}

template <class A, bool FAST>
__device__ __forceinline__ void run_inv_stages(typename A::T *sm, const PassGeom &g, int tile, const Ctx &c) {
    const int T = 1 << g.tl;
    const int adj = g.strided ? (g.logN - g.S - g.logC) : 0;
    const int base = g.strided ? 0 : (tile << g.tl);
    int s = g.s0, left = g.S, log_d = g.strided ? g.logC : 0;
    Ctx cc = c;
    // fp64 class: words start < 2q and each step of K stages multiplies their bound by 2^K (<= 2^6 over two
    // radix-8 steps, within dp_reduce's |x| < 64 * 2q): reduce at the end of every second step and of the last one
    int nstep = 0;
    auto arm = [&](int K) {
        ++nstep;
        cc.inv_reduce = ((nstep & 1) == 0) || (left - K <= 0);
    };
    if (FAST && g.tl == 12 && g.S == 12) {   // the next step's twiddles are requested before each barrier
        StepTwInv<A, 0> t0;
        StepTwInv<A, 3> t3;
        StepTwInv<A, 6> t6;
        StepTwInv<A, 9> t9;
        t0.load(c, s, adj, g.logN, base, lf_tid());
        arm(3); inv_step8<A, 0>(sm, t0, cc, lf_tid()); left -= 3;
        t3.load(c, s + 3, adj, g.logN, base, lf_tid());
        lds_barrier();
        arm(3); inv_step8<A, 3>(sm, t3, cc, lf_tid()); left -= 3;
        t6.load(c, s + 6, adj, g.logN, base, lf_tid());
        lds_barrier();
        arm(3); inv_step8<A, 6>(sm, t6, cc, lf_tid()); left -= 3;
        t9.load(c, s + 9, adj, g.logN, base, lf_tid());
        lds_barrier();
        arm(3); inv_step8<A, 9>(sm, t9, cc, lf_tid());
        lds_barrier();
        return;
    }
    while (left > 0) {
        if (FAST && left >= 3) {
            arm(3);
            inv_step<3, A>(sm, T, log_d, s, adj, g.logN, base, cc);
            s += 3; left -= 3; log_d += 3;
        } else if (FAST && left == 2) {
            arm(2);
            inv_step<2, A>(sm, T, log_d, s, adj, g.logN, base, cc);
            s += 2; left -= 2; log_d += 2;
        } else {
            arm(1);
            inv_step<1, A>(sm, T, log_d, s, adj, g.logN, base, cc);
            s += 1; left -= 1; log_d += 1;
        }
        lds_barrier();
    }
}

// The same constants with 1 / q taken from slot 0 of the limb's fp64 twiddle table: no stage ever reads entry 0
// (forward indices start at 2^s >= 1, inverse ones at N >> (s + 1) >= 1), so lf_twiddle_dp stores the correctly
// rounded reciprocal there and the pass kernels skip the 13-instruction fp64 division per wave.
__device__ __forceinline__ RowDp make_dp_tab(const RowMod &m, const double *__restrict__ tw_dp) {
    RowDp d;
    d.q = (double)m.q;
    d.q2 = 2.0 * d.q;
    d.qinv = tw_dp[0];
    d.q2inv = 0.5 * d.qinv;
    return d;
}

__device__ __forceinline__ RowDp make_dp(const RowMod &m) {
    RowDp d;
    d.q = (double)m.q;
    d.q2 = 2.0 * d.q;
    d.qinv = 1.0 / d.q;
    d.q2inv = 0.5 * d.qinv;
    return d;
}

// LDS tile -> global, 16 B per lane
__device__ __forceinline__ void store_tile_raw(const i64 *sm, i64 *row, const PassGeom &g, int tile) {
    const int T = 1 << g.tl;
    for (int L = threadIdx.x * 2; L < T; L += NTT_THREADS * 2) {
        longlong2 v;
        v.x = sm[PAD(L)];
        v.y = sm[PAD(L + 1)];
        *reinterpret_cast<longlong2 *>(row + tile_gaddr(g, tile, L)) = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Register-fed 4096-word contiguous pass (tl = 12, S = 12): the tile never sits in LDS as raw words.
//   forward: the first radix-8 step (distance 512) takes its 8 words straight from global memory (8-byte
//            loads, 512 contiguous bytes per wave), the last one (8 consecutive words per thread) hands its
//            results to the stores through the wave's OWN 512-word LDS span — no block barrier;
//   inverse: mirror image (16-byte loads -> wave-private span -> 8 consecutive words; last step stores
//            8-byte words 512 apart straight from registers, chain tail applied in registers).
// Against the LDS-resident form this halves the LDS round trips per word (4 instead of 6-7) and leaves
// 3 block barriers per tile instead of 6-7.  A tile holding a word outside [0, 2q) is detected before
// anything is stored; the caller then runs the generic path on it (returns false, block-uniform).
// ------------------------------------------------------------------------------------------------
static_assert(NTT_THREADS == 512, "one flag byte per wave fills the 8-byte flag word");

// wave v owns byte v of the flag word: every wave writes its byte, so no reset and no reset barrier
__device__ __forceinline__ void wave_flag_set(i64 *sm, int pred) {
    const bool any = __builtin_amdgcn_ballot_w64(pred != 0) != 0;
    const int t = lf_tid();
    if ((t & 63) == 0) reinterpret_cast<unsigned char *>(sm + NTT_FLAG_WORD)[t >> 6] = any ? 1 : 0;
}

// LDS traffic of one wave is processed in order: a wave-private exchange needs no s_barrier
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// steps of the forward tile; x[e] = word (w + 512 e) on entry, word (8 w + e) on exit.
// KEEP: the thread index stays in ONE register for the whole tile (relaxed kernels: no out-of-line call that would
// spill it); otherwise it is rebuilt from the wave index and v_mbcnt at every use (see lf_tid).
// CHECK: the tile may hold a word outside [0, 2q) (flag word written by wave_flag_set): relaxed fp64 tiles never do.
#define LF_TID(KEEP, w0) ((KEEP) ? (w0) : lf_tid())
template <class A, bool KEEP, bool CHECK>
__device__ __forceinline__ bool fwd_tile12_core(typename A::T *smt, i64 *sm, typename A::T (&x)[8], int s, int E, int base,
                                                const Ctx &c, int w0) {
    StepTw<A, 9> t9;
    StepTw<A, 6> t6;
    StepTw<A, 3> t3;
    StepTw<A, 0> t0;
    t9.load(c, s, E, base, LF_TID(KEEP, w0));
    fwd_regs8<A, 9>(x, t9, c);
    {
        typename A::T *sp = smt + PAD(LF_TID(KEEP, w0));   // PAD(w + 512 e) = PAD(w) + 576 e
#pragma unroll
        for (int e = 0; e < 8; ++e) sp[e * PAD_STRIDE(9)] = x[e];
    }
    t6.load(c, s + 3, E, base, LF_TID(KEEP, w0));
    lds_barrier();
    if (CHECK && sm[NTT_FLAG_WORD] != 0) {
        lds_barrier();   // every wave has read the flag before the generic path resets it
        return false;
    }
    fwd_step8<A, 6>(smt, t6, c, LF_TID(KEEP, w0));
    t3.load(c, s + 6, E, base, LF_TID(KEEP, w0));
    lds_barrier();
    fwd_step8<A, 3>(smt, t3, c, LF_TID(KEEP, w0));
    t0.load(c, s + 9, E, base, LF_TID(KEEP, w0));
    lds_barrier();
    {
        const int w = LF_TID(KEEP, w0);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = smt[9 * w + e];
    }
    fwd_regs8<A, 0>(x, t0, c);
    return true;
}

// 8 consecutive result words per thread -> global, 16 B per lane and 1 KiB contiguous per wave instruction
__device__ __forceinline__ void store_tile12_regs(i64 *sm, const i64 (&o)[8], i64 *dst, int w) {
#pragma unroll
    for (int e = 0; e < 8; ++e) sm[9 * w + e] = o[e];
    wave_lds_sync();
    const int L0 = ((w >> 6) << 9) + ((w & 63) << 1);
    const i64 *sp = sm + PAD(L0);   // L0 even: PAD(L0 + 128 i + 1) = PAD(L0) + 144 i + 1
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        longlong2 v;
        v.x = sp[i * PAD_STRIDE(7)];
        v.y = sp[i * PAD_STRIDE(7) + 1];
        *reinterpret_cast<longlong2 *>(dst + L0 + (i << 7)) = v;
    }
}

template <bool DP, bool RLX>
__device__ __forceinline__ bool fwd_tile12(i64 *sm, i64 *__restrict__ row, int tile, const PassGeom &g, const Ctx &c,
                                           bool enter, i64 rs) {
    const int w = lf_tid();
    const int base = tile << 12;
    const i64 q2 = c.m.q2;
    // relaxed transforms: the fp64 class converts signed words directly (dp_from_signed: the balanced arithmetic is
    // sign-agnostic), the integer class folds them into [0, 2q) first
    const bool fold = RLX && !DP;
    i64 raw[8];
    {
        const i64 *src = row + base + w;
#pragma unroll
        for (int e = 0; e < 8; ++e) raw[e] = src[e << 9];
    }
    int odd = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if (fold) raw[e] = raw[e] < 0 ? raw[e] + q2 : raw[e];   // residues only: fold the signed-lazy words
        if (!(RLX && DP)) odd |= ((u64)raw[e] >= (u64)q2);      // relaxed fp64 tiles accept any non-negative representative
    }
    i64 o[8];
    if (DP) {
        double *smd = reinterpret_cast<double *>(sm);
        double x[8];
        const double r1 = enter ? (double)((1ull << 62) % c.m.q) : 0.0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            double v = RLX ? dp_from_signed(raw[e]) : dp_from_word(raw[e]);
            if (enter) {   // Montgomery entry, emulated exactly: REDC62(a * R^2)
                v = dp_mulmod(v, r1, c.d);
                if (!RLX && dp_below_fix_limit(v)) v = dp_lazy_fix(v, (u64)raw[e], (u64)rs, c.d.q);
            }
            x[e] = v;
        }
        if (!RLX) wave_flag_set(sm, odd);
        const bool ok = RLX ? fwd_tile12_core<ArithDpR, true, false>(smd, sm, x, g.s0, g.logN, base, c, w)
                            : fwd_tile12_core<ArithDp, false, true>(smd, sm, x, g.s0, g.logN, base, c, w);
        if (!ok) return false;
        // the pass accumulated without subtractions: back to the lazy word in [0, 2q) (relaxed: canonical residue)
        const double md = RLX ? c.d.q : c.d.q2, mi = RLX ? c.d.qinv : c.d.q2inv;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = dp_to_word(dp_reduce(x[e], md, mi));
    } else {
        if (enter) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                raw[e] = mm62s(raw[e], rs, c.m.q, c.m.k);
                odd |= ((u64)raw[e] >= (u64)q2);
            }
        }
        wave_flag_set(sm, odd);
        if (!fwd_tile12_core<ArithInt<false>, RLX, true>(sm, sm, raw, g.s0, g.logN, base, c, w)) return false;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = raw[e];
    }
    store_tile12_regs(sm, o, row + base, LF_TID(RLX, w));
    return true;
}

// steps of the inverse tile; x[e] = word (8 w + e) on entry, word (w + 512 e) on exit (KEEP / CHECK: see the forward core)
template <class A, bool KEEP, bool CHECK>
__device__ __forceinline__ bool inv_tile12_core(typename A::T *smt, i64 *sm, typename A::T (&x)[8], int s, int logN, int base,
                                                const Ctx &c, int w0) {
    Ctx cc = c;
    StepTwInv<A, 0> t0;
    StepTwInv<A, 3> t3;
    StepTwInv<A, 6> t6;
    StepTwInv<A, 9> t9;
    // fp64 class: words double per stage; reduced mod 2q at the end of every second step (run_inv_stages)
    t0.load(c, s, 0, logN, base, LF_TID(KEEP, w0));
    cc.inv_reduce = 0;
    inv_regs8<A, 0>(x, t0, cc);
    {
        const int w = LF_TID(KEEP, w0);
#pragma unroll
        for (int e = 0; e < 8; ++e) smt[9 * w + e] = x[e];
    }
    t3.load(c, s + 3, 0, logN, base, LF_TID(KEEP, w0));
    lds_barrier();
    if (CHECK && sm[NTT_FLAG_WORD] != 0) {
        lds_barrier();
        return false;
    }
    cc.inv_reduce = 1;
    inv_step8<A, 3>(smt, t3, cc, LF_TID(KEEP, w0));
    t6.load(c, s + 6, 0, logN, base, LF_TID(KEEP, w0));
    lds_barrier();
    cc.inv_reduce = 0;
    inv_step8<A, 6>(smt, t6, cc, LF_TID(KEEP, w0));
    t9.load(c, s + 9, 0, logN, base, LF_TID(KEEP, w0));
    lds_barrier();
    {
        const typename A::T *sp = smt + PAD(LF_TID(KEEP, w0));
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = sp[e * PAD_STRIDE(9)];
    }
    cc.inv_reduce = 1;
    inv_regs8<A, 9>(x, t9, cc);
    return true;
}

// integer chain tail of one word (K.cu:527-529, 754-902)
__device__ __forceinline__ i64 inv_tail_int(i64 t, int tail, i64 ninv, const Ctx &c) {
    if (tail == TAIL_NONE) return t;
    const i64 qq = (i64)c.m.q;
    i64 z = mm62s(t, ninv, c.m.q, c.m.k);
    if (tail >= 1) z = redc62(z, c.m.q, c.m.k);
    if (tail >= 2) z = z < qq ? z : z - qq;
    if (tail >= 3) z = z <= (qq >> 1) ? z : z - qq;
    return z;
}

template <bool DP, bool RLX>
__device__ __forceinline__ bool inv_tile12(i64 *sm, const i64 *src_row, i64 *dst_row, int tile, const PassGeom &g, const Ctx &c,
                                           const i64 *__restrict__ Ninv, int tail, int crow) {
    const int w = lf_tid();
    const int base = tile << 12;
    const i64 q2 = c.m.q2;
    // 16-byte loads, 1 KiB contiguous per wave instruction, exchanged through the wave's own LDS span.
    // Relaxed transforms take NON-NEGATIVE words (include/ckks_hip.h: every producer inside the library — tensor
    // product, key inner product — writes canonical or lazy [0, 2q) words): nothing to fold, and the fp64 class
    // accepts any such representative, so it has no out-of-range tiles either.
    const int L0 = ((w >> 6) << 9) + ((w & 63) << 1);
    int odd = 0;
    {
        longlong2 in[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) in[i] = *reinterpret_cast<const longlong2 *>(src_row + base + L0 + (i << 7));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const longlong2 v = in[i];
            if (!(RLX && DP)) odd |= ((u64)v.x >= (u64)q2) | ((u64)v.y >= (u64)q2);
            i64 *sp = sm + PAD(L0);   // L0 even: PAD(L0 + 128 i + 1) = PAD(L0) + 144 i + 1
            sp[i * PAD_STRIDE(7)] = v.x;
            sp[i * PAD_STRIDE(7) + 1] = v.y;
        }
    }
    if (!(RLX && DP)) wave_flag_set(sm, odd);
    wave_lds_sync();
    i64 raw[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) raw[e] = sm[9 * w + e];
    wave_lds_sync();   // the span is rewritten (as doubles) by the first step
    i64 *out = dst_row + base + w;
    if (DP) {
        double *smd = reinterpret_cast<double *>(sm);
        double x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = dp_from_word(raw[e]);
        const bool ok = RLX ? inv_tile12_core<ArithDpR, true, false>(smd, sm, x, g.s0, g.logN, base, c, w)
                            : inv_tile12_core<ArithDp, false, true>(smd, sm, x, g.s0, g.logN, base, c, w);
        if (!ok) return false;
        const int t_eff = tail;   // the caller passes TAIL_NONE unless this is the last pass
        // chain tail: z = REDC(t * Ninv); [redc]; [reduce]; [signed]    (K.cu:527-529, 754-902)
        const i64 qq = (i64)c.m.q;
        const double ninv_plain = c.d.q - (double)((c.m.q - 1) >> g.logN);                  // N^-1 mod q
        const double rinv = (double)(u64)((((u128)c.m.k * (u128)c.m.q) + 1) >> 62);         // R^-1 mod q
        const double c2 = g.plain ? ninv_plain : dp_mulmod(ninv_plain, rinv, c.d);          // N^-1 (R^-1) mod q
        const i64 ninv_mont = (t_eff != TAIL_NONE) ? Ninv[crow] : 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const double t = x[e];
            double z;
            if (t_eff == TAIL_NONE) {
                z = g.relaxed ? dp_addmask(t, c.d.q) : t;   // relaxed words are balanced residues
            } else if (t_eff >= 2) {
                z = dp_mulmod(t, c2, c.d);
                if (t_eff >= 3) z = z <= (double)(qq >> 1) ? z : z - c.d.q;
            } else {
                z = dp_mulmod(t, ninv_plain, c.d);
                if (dp_below_fix_limit(z)) z = dp_lazy_fix(z, (u64)t, (u64)ninv_mont, c.d.q);
                if (t_eff == 1) z = (z == c.d.q) ? c.d.q : dp_mulmod(z, rinv, c.d);   // redc(q) = q (K.cu:587-606)
            }
            out[e << 9] = (t_eff >= 3) ? (i64)z : dp_to_word(z);
        }
    } else {
        if (!inv_tile12_core<ArithInt<false>, RLX, true>(sm, sm, raw, g.s0, g.logN, base, c, w)) return false;
        const int t_eff = tail;   // the caller passes TAIL_NONE unless this is the last pass
        const i64 ninv = (t_eff != TAIL_NONE) ? Ninv[crow] : 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) out[e << 9] = inv_tail_int(raw[e], t_eff, ninv, c);
    }
    return true;
}

// ------------------------------------------------------------------------------------------------
// forward pass.  DP = true: fp64 class rows; false: integer class rows.
// ------------------------------------------------------------------------------------------------
template <bool DP, bool RLX>
__device__ __forceinline__ void fwd_pass_body(i64 *sm, int b, i64 *__restrict__ a, const PassGeom &g0, const RowList &rl,
                                              const i64 *__restrict__ psi_br, const double *__restrict__ psi_dp,
                                              const i64 *__restrict__ Rs, const i64 *__restrict__ ql,
                                              const i64 *__restrict__ qh, const i64 *__restrict__ kl,
                                              const i64 *__restrict__ kh) {
    PassGeom g = g0;
    g.relaxed = RLX ? 1 : 0;   // compile-time: each kernel carries one of the two arithmetic modes
    const int T = 1 << g.tl;
    const bool enter = (Rs != nullptr) && !(DP && g.plain);
    int poly, crow, tile;
    block_coords(g, rl, b, poly, crow, tile);
    {
        const int item = 0;
        Ctx c;
        c.m = load_mod(ql, qh, kl, kh, crow);
        c.tw_mont = psi_br + ((i64)crow << g.logN);
        set_aux<DP>(c, psi_dp, crow, g.logN);
        c.d = DP ? make_dp_tab(c.m, c.tw_dp) : make_dp(c.m);
        c.relaxed = g.relaxed;
        c.inv_reduce = 0;
        i64 *row = a + ((i64)(poly * g.rows + crow) << g.logN);
        const int cur_tile = tile;
        const i64 rs = enter ? Rs[crow] : 0;

        // contiguous 4096-word pass: register-fed form; false = the tile holds a word outside [0, 2q)
        if (g.tl == 12 && g.S == 12 && !g.strided) {
            if (fwd_tile12<DP, RLX>(sm, row, tile, g, c, enter, rs)) return;
        }

        if (threadIdx.x == 0) sm[NTT_FLAG_WORD] = 0;   // both block_or flag words
        lds_barrier();
        longlong2 pre[NTT_PRE];
        prefetch_tile(pre, row, g, tile);

        const int odd_raw = stash_tile(sm, pre, g, c.m.q2);
        // relaxed fp64 tiles accept any non-negative residue representative
        const bool odd = block_or(sm, odd_raw, item & 1) && !(DP && g.relaxed);

        if (DP && !odd) {
            double *smd = reinterpret_cast<double *>(sm);
            // words -> doubles (+ optional Montgomery entry, emulated exactly: REDC62(a * R^2))
            const double r1 = enter ? (double)((1ull << 62) % c.m.q) : 0.0;
            for (int L = threadIdx.x * 2; L < T; L += NTT_THREADS * 2) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const i64 raw = sm[PAD(L + e)];
                    double v = dp_from_word(raw);
                    if (enter) {
                        v = dp_mulmod(v, r1, c.d);
                        if (!g.relaxed && dp_below_fix_limit(v)) v = dp_lazy_fix(v, (u64)raw, (u64)rs, c.d.q);
                    }
                    smd[PAD(L + e)] = v;
                }
            }
            lds_barrier();
            if (g.relaxed) run_fwd_stages<ArithDpR, true>(smd, g, cur_tile, c);
            else run_fwd_stages<ArithDp, true>(smd, g, cur_tile, c);
            // the pass accumulated without subtractions: back to the lazy word in [0, 2q)
            // (relaxed: the canonical residue)
            const double md = g.relaxed ? c.d.q : c.d.q2, mi = g.relaxed ? c.d.qinv : c.d.q2inv;
            for (int L = threadIdx.x * 2; L < T; L += NTT_THREADS * 2) {
                longlong2 o;
                o.x = dp_to_word(dp_reduce(smd[PAD(L)], md, mi));
                o.y = dp_to_word(dp_reduce(smd[PAD(L + 1)], md, mi));
                *reinterpret_cast<longlong2 *>(row + tile_gaddr(g, cur_tile, L)) = o;
            }
        } else {
            // integer class (or a signed-lazy tile of the fp64 class)
            int odd2 = 0;
            if (enter) {
                for (int L = threadIdx.x * 2; L < T; L += NTT_THREADS * 2) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const i64 v = mm62s(sm[PAD(L + e)], rs, c.m.q, c.m.k);
                        odd2 |= ((u64)v >= (u64)c.m.q2);
                        sm[PAD(L + e)] = v;
                    }
                }
            }
            lds_barrier();
            const bool sgn = block_or(sm, odd2, item & 1) || odd;
            if (sgn || DP) run_fwd_stages<ArithInt<true>, false>(sm, g, cur_tile, c);   // rare: compact stage-by-stage loop
            else run_fwd_stages<ArithInt<false>, true>(sm, g, cur_tile, c);
            store_tile_raw(sm, row, g, cur_tile);
        }
    }
}

template <bool DP, bool RLX>
__global__ void __launch_bounds__(NTT_THREADS, DP ? 6 : 4) ntt_fwd_pass(i64 *__restrict__ a, PassGeom g, RowList rl,
                                                            const i64 *__restrict__ psi_br, const double *__restrict__ psi_dp,
                                                            const i64 *__restrict__ Rs, const i64 *__restrict__ ql,
                                                            const i64 *__restrict__ qh, const i64 *__restrict__ kl,
                                                            const i64 *__restrict__ kh) {
    __shared__ i64 sm[NTT_LDS_WORDS + 1];
    fwd_pass_body<DP, RLX>(sm, blockIdx.x, a, g, rl, psi_br, psi_dp, Rs, ql, qh, kl, kh);
}

// Both arithmetic classes of one pass in ONE launch: the first `in_blocks` blocks (a multiple of 8, so the
// XCD alignment of the rest is unchanged) work on the integer-class limbs, the others on the fp64 class.
// The few, long integer-class blocks start first; no side stream, no fork / join events, half the launches.
struct ClassLists {
    RowList in, dp;
    int in_blocks;      // padded to a multiple of 8; blocks in [in_real, in_blocks) exit
    int in_real;
};

template <bool RLX>
__global__ void __launch_bounds__(NTT_THREADS, NTT_PASS_WAVES) ntt_fwd_pass_mixed(i64 *__restrict__ a, PassGeom g, ClassLists cl,
                                                                       const i64 *__restrict__ psi_br,
                                                                       const double *__restrict__ psi_dp,
                                                                       const i64 *__restrict__ Rs, const i64 *__restrict__ ql,
                                                                       const i64 *__restrict__ qh, const i64 *__restrict__ kl,
                                                                       const i64 *__restrict__ kh) {
    __shared__ i64 sm[NTT_LDS_WORDS + 1];
    const int b = blockIdx.x;
    if (b < cl.in_blocks) {
        if (b < cl.in_real) fwd_pass_body<false, RLX>(sm, b, a, g, cl.in, psi_br, psi_dp, Rs, ql, qh, kl, kh);
    } else {
        fwd_pass_body<true, RLX>(sm, b - cl.in_blocks, a, g, cl.dp, psi_br, psi_dp, Rs, ql, qh, kl, kh);
    }
}

// ------------------------------------------------------------------------------------------------
// forward STRIDED pass as one radix-2^K register step straight from global memory (two-pass transforms
// with K = logN - 12 <= 4 leading stages).  Thread = one column: its 2^K words sit N/2^K apart, so a wave's
// loads and stores are 512 contiguous bytes per row; the 2^K - 1 twiddles of these stages depend only on
// the row index, i.e. they are the table entries 1 .. 2^K - 1 for every thread of the limb (scalar loads).
// No LDS, no barriers, no per-lane twiddle traffic: 2.4x fewer VALU instructions than staging the same
// four stages through a 4096-word LDS tile.  Arithmetic, entry and range handling are those of
// ntt_fwd_pass; the class decision (exact fp64 vs signed integer routine) is taken per lane.
// ------------------------------------------------------------------------------------------------
#ifndef NTT_COL_THREADS
#define NTT_COL_THREADS 128   // columns per block = contiguous 8-byte words per row and block (measured at the bench workload,
                              // column pass alone: 64 -> 0.72-0.73 ms, 128 -> 0.68-0.70, 256 -> 0.71-0.73, 512 -> 0.79-0.80)
#endif

// row pointer of a column step as an opaque SGPR pair: keeps the compiler from folding the lane index into 2^K
// per-lane 64-bit addresses (32 VGPRs held from the loads to the stores at K = 4)
__device__ __forceinline__ i64 *uniform_row(i64 *base, i64 off) {
    i64 *p = base + off;
    asm("" : "+s"(p));
    return p;
}

template <class T>
__device__ __forceinline__ T *uniform_ptr(T *p) {
    asm("" : "+s"(p));
    return p;
}

// Streaming accesses (the `nt` bit of global loads / stores): words a pass reads once and writes once should not displace
// the twiddle rows — as many bytes per tile as the data, re-read by every polynomial of the batch — from L1 / L2.
// In-process A/B on the headline step (tools/ab_inproc.py, a build with the four switches below false beside this one): whole step -2.8 %, tiled pass
// -2.3 %, column pass -2.0 %.  Exact transforms (the standalone lf_ntt / lf_intt of large batches) stream everything.
// Relaxed ones are the engine's internal passes, where the NEXT kernel may still find a pass's output in cache: loads of the
// tiled pass, the extension kernel's stores and the inverse passes stream (gold rotate -1 %, silver -3 %); the tiled pass's
// STORES do not (streamed, the inner product that reads them next went 102 -> 118 us at gold), nor the column pass of
// cc_mult's opening, the inner product's digit loads or its sums.
typedef long long ll2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ longlong2 nt_load2(const i64 *p) {
    const ll2_t v = __builtin_nontemporal_load(reinterpret_cast<const ll2_t *>(p));
    longlong2 r;
    r.x = v.x, r.y = v.y;
    return r;
}
__device__ __forceinline__ void nt_store2(i64 *p, const longlong2 &v) {
    ll2_t t;
    t.x = v.x, t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<ll2_t *>(p));
}
#define NT_EXACT true
#define NT_RLOAD true
#define NT_INV true
#define NT_KS_EXT true
#define NT_RSTORE false
#define NT_RCOLS false
// (nt: wave-uniform)
#define COLS_LD_ALL(dst, ptr)                                                            \
    if (nt) {                                                                            \
        _Pragma("unroll") for (int k = 0; k < R; ++k) dst[k] = __builtin_nontemporal_load(ptr); \
    } else {                                                                             \
        _Pragma("unroll") for (int k = 0; k < R; ++k) dst[k] = *(ptr);                   \
    }
#define COLS_ST_ALL(ptr, val)                                                            \
    if (nt) {                                                                            \
        _Pragma("unroll") for (int k = 0; k < R; ++k) __builtin_nontemporal_store((i64)(val), ptr); \
    } else {                                                                             \
        _Pragma("unroll") for (int k = 0; k < R; ++k) *(ptr) = (val);                    \
    }
#define INV_LD(p) (NT_INV ? __builtin_nontemporal_load(p) : *(p))
#define INV_LD2(p) (NT_INV ? nt_load2(p) : *reinterpret_cast<const longlong2 *>(p))
#define INV_ST(p, v)                                              \
    do {                                                          \
        if (NT_INV) __builtin_nontemporal_store((i64)(v), p);     \
        else *(p) = (v);                                          \
    } while (0)

template <class A, int K>
__device__ __forceinline__ void cols_fwd_stages(typename A::T (&x)[1 << K], const Ctx &c) {
#pragma unroll
    for (int u = 0; u < K; ++u) {
        const int du = 1 << (K - 1 - u);
#pragma unroll
        for (int j = 0; j < (1 << u); ++j) {
            const int e0 = j << (K - u);
            const int idx = (1 << u) + j;
            const typename A::W wv = A::tw(c, idx);
#pragma unroll
            for (int t = 0; t < du; ++t) A::fwd(c, x[e0 + t], x[e0 + t + du], wv, idx);
        }
    }
}

// Optional source of the column pass: the words are produced on the fly by the engine's rescale
// (ckks_engine.py:1029-1041: reduce_q(REDC((in - row0) * scale) + [row0 > q_l / 2])) from another tensor, so
// cc_mult's rescale costs no launch and no pass over HBM of its own.
#define LF_NTT_RS_MAX 8
struct RescaleSrc {
    const i64 *in[LF_NTT_RS_MAX];     // per polynomial: first surviving row of the source component, [rows, N]
    const i64 *row0[LF_NTT_RS_MAX];   // per polynomial: the dropped limb's row, [N]
    const i64 *scales;                // per surviving row: q_l^-1 * R mod q_row
    i64 round_at;
};

template <bool DP, int K, bool RS = false>
__device__ __forceinline__ void fwd_cols_body(int b, i64 *__restrict__ a, const PassGeom &g, const RowList &rl,
                                              const i64 *__restrict__ psi_br, const double *__restrict__ psi_dp,
                                              const i64 *__restrict__ Rs, const i64 *__restrict__ ql,
                                              const i64 *__restrict__ qh, const i64 *__restrict__ kl,
                                              const i64 *__restrict__ kh, const RescaleSrc *rsrc = nullptr) {
    constexpr int R = 1 << K;
    const int logC = g.logN - K;
    const int chunks = (1 << logC) / NTT_COL_THREADS;
    const int chunk = b % chunks, r = b / chunks;
    // (the integer divisions run on the VALU: pin their wave-uniform results back into SGPRs)
    const int poly = __builtin_amdgcn_readfirstlane(r % g.batch), crow = __builtin_amdgcn_readfirstlane((int)rl.id[r / g.batch]);
    const bool enter = (Rs != nullptr) && !(DP && g.plain);

    Ctx c;
    c.m = load_mod(ql, qh, kl, kh, crow);
    c.tw_mont = psi_br + ((i64)crow << g.logN);
    set_aux<DP>(c, psi_dp, crow, g.logN);
    c.d = DP ? make_dp_tab(c.m, c.tw_dp) : make_dp(c.m);
    c.relaxed = g.relaxed;
    c.inv_reduce = 0;
    const i64 rs = enter ? Rs[crow] : 0;
    // wave-uniform base + lane index: the 2^K row addresses stay in SGPRs (a per-lane pointer would pin 2^K 64-bit
    // addresses in VGPRs from the loads to the stores)
    i64 *colu = a + ((i64)(poly * g.rows + crow) << g.logN) + chunk * NTT_COL_THREADS;
    const unsigned lane = threadIdx.x;
    const bool nt = g.relaxed ? NT_RCOLS : NT_EXACT;

    if (RS && DP && g.relaxed) {
        // relaxed fp64 class (cc_mult's opening): only residues matter, so the rescale itself runs in fp64 — one balanced
        // product per word with the PLAIN constant q_l^-1 mod q instead of a 62-bit REDC, and no canonical fix-up
        // (|in - row0| < 2^42; the balanced arithmetic of the stages is sign-agnostic)
        const i64 j = (i64)chunk * NTT_COL_THREADS + threadIdx.x;
        const i64 *src = rsrc->in[poly] + ((i64)crow << g.logN) + j;
        const i64 *z0 = rsrc->row0[poly] + j;
        i64 sp = redc62(rsrc->scales[crow], c.m.q, c.m.k);
        sp = sp < (i64)c.m.q ? sp : sp - (i64)c.m.q;
        const double scp = (double)sp;
        const double r1 = enter ? (double)((1ull << 62) % c.m.q) : 0.0;
        double x[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const i64 z = z0[(i64)k << logC];
            double v = dp_mulmod_bal(dp_from_signed(src[(i64)k << logC] - z), scp, c.d) + (z > rsrc->round_at ? 1.0 : 0.0);
            if (enter) v = dp_mulmod_bal(v, r1, c.d);
            x[k] = v;
        }
        cols_fwd_stages<ArithDpR, K>(x, c);
        if (g.pln & PLN_OUT) {   // cc_mult's operand stack: canonical words as two planes (the tiled pass behind: fwd_tile16<.., PLN>)
            i64 *rowb = a + ((i64)(poly * g.rows + crow) << g.logN);
            const i64 col0 = (i64)chunk * NTT_COL_THREADS;
            unsigned *lo = reinterpret_cast<unsigned *>(rowb) + col0;
            unsigned short *hi = reinterpret_cast<unsigned short *>(rowb + ((i64)1 << (g.logN - 1))) + col0;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const i64 o = dp_to_word(dp_reduce(x[k], c.d.q, c.d.qinv));
                __builtin_nontemporal_store((unsigned)o, uniform_ptr(lo + ((i64)k << logC)) + lane);
                __builtin_nontemporal_store((unsigned short)((u64)o >> 32), uniform_ptr(hi + ((i64)k << logC)) + lane);
            }
            return;
        }
        COLS_ST_ALL(uniform_row(colu, (i64)k << logC) + lane, dp_to_word(dp_reduce(x[k], c.d.q, c.d.qinv)))
        return;
    }
    i64 w[R];
    if (RS) {
        const i64 j = (i64)chunk * NTT_COL_THREADS + threadIdx.x;
        const i64 *src = rsrc->in[poly] + ((i64)crow << g.logN) + j;
        const i64 *z0 = rsrc->row0[poly] + j;
        const i64 sc = rsrc->scales[crow], qq = (i64)c.m.q;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const i64 z = z0[(i64)k << logC];
            const i64 v = mm62s(src[(i64)k << logC] - z, sc, c.m.q, c.m.k) + (i64)(z > rsrc->round_at);
            w[k] = v < qq ? v : v - qq;
        }
    } else {
        COLS_LD_ALL(w, uniform_row(colu, (i64)k << logC) + lane)
    }
    int odd = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        if (g.relaxed && !DP) w[k] = w[k] < 0 ? w[k] + c.m.q2 : w[k];   // residues only: fold signed-lazy words
        odd |= ((u64)w[k] >= (u64)c.m.q2);
    }
    if (DP && (g.relaxed || !odd)) {
        double x[R];
        const double r1 = enter ? (double)((1ull << 62) % c.m.q) : 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            // relaxed: signed words convert directly (the balanced arithmetic is sign-agnostic)
            double v = g.relaxed ? dp_from_signed(w[k]) : dp_from_word(w[k]);
            if (enter) {   // Montgomery entry, emulated exactly: REDC62(a * R^2)
                v = dp_mulmod(v, r1, c.d);
                if (!g.relaxed && dp_below_fix_limit(v)) v = dp_lazy_fix(v, (u64)w[k], (u64)rs, c.d.q);
            }
            x[k] = v;
        }
        if (g.relaxed) cols_fwd_stages<ArithDpR, K>(x, c);
        else cols_fwd_stages<ArithDp, K>(x, c);
        const double md = g.relaxed ? c.d.q : c.d.q2, mi = g.relaxed ? c.d.qinv : c.d.q2inv;
        COLS_ST_ALL(uniform_row(colu, (i64)k << logC) + lane, dp_to_word(dp_reduce(x[k], md, mi)))
    } else {
        // integer class, or a lane of the fp64 class holding signed-lazy words
        if (enter) {
#pragma unroll
            for (int k = 0; k < R; ++k) {
                w[k] = mm62s(w[k], rs, c.m.q, c.m.k);
                odd |= ((u64)w[k] >= (u64)c.m.q2);
            }
        }
        if (!DP && g.relaxed) {
            // residues only: Shoup products on lazy words (the fold / the entry left them in [0, 2q)), canonical out
            cols_fwd_stages<ArithShoup, K>(w, c);
            COLS_ST_ALL(uniform_row(colu, (i64)k << logC) + lane, ArithShoup::canon(c, w[k]))
            return;
        }
        if (odd || DP) cols_fwd_stages<ArithInt<true>, K>(w, c);
        else cols_fwd_stages<ArithInt<false>, K>(w, c);
        COLS_ST_ALL(uniform_row(colu, (i64)k << logC) + lane, w[k])
    }
}

// The column pass of an EXACT transform through a workspace (lf_ntt_ws; the tiled pass: fwd_tile16_ws, ckks_ntt_tile16.h):
// reads the tensor, writes the workspace — fp64-class rows as planes (u32 low[N] | u16 mid[N] at byte 4 N | u16 top[N] at
// byte 6 N, the last only from a wave that met an operand outside [0, 2q): flag byte wflags[row][v][e], set by the wave holding the
// columns 256 e + 64 v .. + 63), integer-class
// rows as raw words.  The arithmetic is fwd_cols_body's exact branch, statement by statement.
template <bool DP, int K>
__device__ __forceinline__ void fwd_cols_ws_body(int b, const i64 *__restrict__ a, i64 *__restrict__ ws,
                                                 unsigned char *__restrict__ wflags, const PassGeom &g, const RowList &rl,
                                                 const i64 *__restrict__ psi_br, const double *__restrict__ psi_dp,
                                                 const i64 *__restrict__ Rs, const i64 *__restrict__ ql,
                                                 const i64 *__restrict__ qh, const i64 *__restrict__ kl,
                                                 const i64 *__restrict__ kh) {
    constexpr int R = 1 << K;
    const int logC = g.logN - K;                                  // 12: the tiled pass behind is the 4096-word one
    const int chunks = (1 << logC) / NTT_COL_THREADS;
    const int chunk = b % chunks, r = b / chunks;
    const int poly = __builtin_amdgcn_readfirstlane(r % g.batch), crow = __builtin_amdgcn_readfirstlane((int)rl.id[r / g.batch]);
    const bool enter = Rs != nullptr;
    Ctx c;
    c.m = load_mod(ql, qh, kl, kh, crow);
    c.tw_mont = psi_br + ((i64)crow << g.logN);
    set_aux<DP>(c, psi_dp, crow, g.logN);
    c.d = DP ? make_dp_tab(c.m, c.tw_dp) : make_dp(c.m);
    c.relaxed = 0;
    c.inv_reduce = 0;
    const i64 rs = enter ? Rs[crow] : 0;
    const i64 ri = (i64)(poly * g.rows + crow);
    const i64 col0 = (i64)chunk * NTT_COL_THREADS;
    const i64 *colu = a + (ri << g.logN) + col0;
    const unsigned lane = threadIdx.x;
    i64 w[R];
#pragma unroll
    for (int k = 0; k < R; ++k) w[k] = __builtin_nontemporal_load(uniform_ptr(colu + ((i64)k << logC)) + lane);
    int odd = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) odd |= ((u64)w[k] >= (u64)c.m.q2);
    if (DP && !odd) {
        double x[R];
        const double r1 = enter ? (double)((1ull << 62) % c.m.q) : 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            double v = dp_from_word(w[k]);
            if (enter) {   // Montgomery entry, emulated exactly: REDC62(a * R^2)
                v = dp_mulmod(v, r1, c.d);
                if (dp_below_fix_limit(v)) v = dp_lazy_fix(v, (u64)w[k], (u64)rs, c.d.q);
            }
            x[k] = v;
        }
        cols_fwd_stages<ArithDp, K>(x, c);
#pragma unroll
        for (int k = 0; k < R; ++k) w[k] = dp_to_word(dp_reduce(x[k], c.d.q2, c.d.q2inv));
    } else {
        if (enter) {
#pragma unroll
            for (int k = 0; k < R; ++k) {
                w[k] = mm62s(w[k], rs, c.m.q, c.m.k);
                odd |= ((u64)w[k] >= (u64)c.m.q2);
            }
        }
        if (odd || DP) cols_fwd_stages<ArithInt<true>, K>(w, c);
        else cols_fwd_stages<ArithInt<false>, K>(w, c);
    }
    i64 *orow = ws + (ri << g.logN);
    // the wave's flag: one of its lanes met an operand outside [0, 2q) — only then can a word it writes lie outside [0, 2q)
    // (the tiled pass tests its words for that only behind a raised flag) or, on an fp64-class row, need more than 48 bits
    const bool wide = __builtin_amdgcn_ballot_w64(odd != 0) != 0;
    if ((lane & 63u) == 0) {   // this wave's 64 columns c0 .. c0 + 63 are words 256 e + 64 v + (0 .. 63) of every tile: wave v, index e
        const unsigned c0 = (unsigned)col0 + lane;
        wflags[(ri << 6) + (((c0 >> 6) & 3u) << 4) + (c0 >> 8)] = wide ? 1 : 0;
    }
    if constexpr (!DP) {
#pragma unroll
        for (int k = 0; k < R; ++k) __builtin_nontemporal_store(w[k], uniform_ptr(orow + col0 + ((i64)k << logC)) + lane);
    } else {
        unsigned *lo = reinterpret_cast<unsigned *>(orow) + col0;
        unsigned short *mid = reinterpret_cast<unsigned short *>(orow + ((i64)1 << (g.logN - 1))) + col0;
#pragma unroll
        for (int k = 0; k < R; ++k) __builtin_nontemporal_store((unsigned)w[k], uniform_ptr(lo + ((i64)k << logC)) + lane);
#pragma unroll
        for (int k = 0; k < R; ++k)
            __builtin_nontemporal_store((unsigned short)((u64)w[k] >> 32), uniform_ptr(mid + ((i64)k << logC)) + lane);
        // (pairs of lanes trading halves through DPP so that each stores four bytes — half the store instructions of this plane —
        // measured neutral: 1.5518 against 1.5540 ms per step, tools/ab_ntt_ws.py with a variant build)
        if (wide) {
            unsigned short *top = reinterpret_cast<unsigned short *>(orow + 3 * ((i64)1 << (g.logN - 2))) + col0;
#pragma unroll
            for (int k = 0; k < R; ++k) top[((i64)k << logC) + lane] = (unsigned short)((u64)w[k] >> 48);
        }
    }
}

template <int K>
__global__ void __launch_bounds__(NTT_COL_THREADS) ntt_fwd_cols_ws(const i64 *__restrict__ a, i64 *__restrict__ ws,
                                                                  unsigned char *__restrict__ wflags, PassGeom g, ClassLists cl,
                                                                  const i64 *__restrict__ psi_br,
                                                                  const double *__restrict__ psi_dp,
                                                                  const i64 *__restrict__ Rs, const i64 *__restrict__ ql,
                                                                  const i64 *__restrict__ qh, const i64 *__restrict__ kl,
                                                                  const i64 *__restrict__ kh) {
    const int b = blockIdx.x;
    if (b < cl.in_blocks) {
        if (b < cl.in_real) fwd_cols_ws_body<false, K>(b, a, ws, wflags, g, cl.in, psi_br, psi_dp, Rs, ql, qh, kl, kh);
    } else {
        fwd_cols_ws_body<true, K>(b - cl.in_blocks, a, ws, wflags, g, cl.dp, psi_br, psi_dp, Rs, ql, qh, kl, kh);
    }
}

template <bool DP, int K>
__global__ void __launch_bounds__(NTT_COL_THREADS) ntt_fwd_cols(i64 *__restrict__ a, PassGeom g, RowList rl,
                                                               const i64 *__restrict__ psi_br,
                                                               const double *__restrict__ psi_dp, const i64 *__restrict__ Rs,
                                                               const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                               const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    fwd_cols_body<DP, K>(blockIdx.x, a, g, rl, psi_br, psi_dp, Rs, ql, qh, kl, kh);
}

template <int K>
__global__ void __launch_bounds__(NTT_COL_THREADS) ntt_fwd_cols_mixed(i64 *__restrict__ a, PassGeom g, ClassLists cl,
                                                                     const i64 *__restrict__ psi_br,
                                                                     const double *__restrict__ psi_dp,
                                                                     const i64 *__restrict__ Rs, const i64 *__restrict__ ql,
                                                                     const i64 *__restrict__ qh, const i64 *__restrict__ kl,
                                                                     const i64 *__restrict__ kh) {
    const int b = blockIdx.x;
    if (b < cl.in_blocks) {
        if (b < cl.in_real) fwd_cols_body<false, K>(b, a, g, cl.in, psi_br, psi_dp, Rs, ql, qh, kl, kh);
    } else {
        fwd_cols_body<true, K>(b - cl.in_blocks, a, g, cl.dp, psi_br, psi_dp, Rs, ql, qh, kl, kh);
    }
}

template <int K>
__global__ void __launch_bounds__(NTT_COL_THREADS) ntt_fwd_cols_mixed_rs(i64 *__restrict__ a, PassGeom g, ClassLists cl,
                                                                        RescaleSrc rsrc, const i64 *__restrict__ psi_br,
                                                                        const double *__restrict__ psi_dp,
                                                                        const i64 *__restrict__ Rs, const i64 *__restrict__ ql,
                                                                        const i64 *__restrict__ qh, const i64 *__restrict__ kl,
                                                                        const i64 *__restrict__ kh) {
    const int b = blockIdx.x;
    if (b < cl.in_blocks) {
        if (b < cl.in_real) fwd_cols_body<false, K, true>(b, a, g, cl.in, psi_br, psi_dp, Rs, ql, qh, kl, kh, &rsrc);
    } else {
        fwd_cols_body<true, K, true>(b - cl.in_blocks, a, g, cl.dp, psi_br, psi_dp, Rs, ql, qh, kl, kh, &rsrc);
    }
}

// ------------------------------------------------------------------------------------------------
// inverse pass (+ fused chain tail on the last pass)
// ------------------------------------------------------------------------------------------------
template <bool DP, bool RLX>
__device__ __forceinline__ void inv_pass_body(i64 *sm, int b, const i64 *src, i64 *dst, const PassGeom &g0, const RowList &rl,
                                              const i64 *__restrict__ ipsi_br, const double *__restrict__ ipsi_dp,
                                              const i64 *__restrict__ Ninv, int tail, const i64 *__restrict__ ql,
                                              const i64 *__restrict__ qh, const i64 *__restrict__ kl,
                                              const i64 *__restrict__ kh) {
    PassGeom g = g0;
    g.relaxed = RLX ? 1 : 0;
    const int T = 1 << g.tl;
    int poly, crow, tile;
    block_coords(g, rl, b, poly, crow, tile);
    {
        const int item = 0;
        Ctx c;
        c.m = load_mod(ql, qh, kl, kh, crow);
        c.tw_mont = ipsi_br + ((i64)crow << g.logN);
        set_aux<DP>(c, ipsi_dp, crow, g.logN);
        c.d = DP ? make_dp_tab(c.m, c.tw_dp) : make_dp(c.m);
        c.relaxed = g.relaxed;
        c.inv_reduce = 0;
        i64 *row = dst + ((i64)(poly * g.rows + crow) << g.logN);
        const int cur_tile = tile, cur_row = crow;
        const i64 qq = (i64)c.m.q;

        // contiguous 4096-word pass: register-fed form; false = the tile holds a word outside [0, 2q)
        if (g.tl == 12 && g.S == 12 && !g.strided) {
            if (inv_tile12<DP, RLX>(sm, src + ((i64)(poly * g.rows + crow) << g.logN), row, tile, g, c, Ninv, tail, crow)) return;
        }

        if (threadIdx.x == 0) sm[NTT_FLAG_WORD] = 0;   // both block_or flag words
        lds_barrier();
        longlong2 pre[NTT_PRE];
        prefetch_tile(pre, src + ((i64)(poly * g.rows + crow) << g.logN), g, tile);

        const int odd_raw = stash_tile(sm, pre, g, c.m.q2);
        const bool odd = block_or(sm, odd_raw, item & 1);

        if (DP && !odd) {
            double *smd = reinterpret_cast<double *>(sm);
            for (int L = threadIdx.x * 2; L < T; L += NTT_THREADS * 2) {
                smd[PAD(L)] = dp_from_word(sm[PAD(L)]);
                smd[PAD(L + 1)] = dp_from_word(sm[PAD(L + 1)]);
            }
            lds_barrier();
            if (g.relaxed) run_inv_stages<ArithDpR, true>(smd, g, cur_tile, c);
            else run_inv_stages<ArithDp, true>(smd, g, cur_tile, c);

            // chain tail: z = REDC(t * Ninv); [redc]; [reduce]; [signed]    (K.cu:527-529, 754-902)
            const double ninv_plain = c.d.q - (double)((c.m.q - 1) >> g.logN);                  // N^-1 mod q
            const double rinv = (double)(u64)((((u128)c.m.k * (u128)c.m.q) + 1) >> 62);         // R^-1 mod q
            const double c2 = g.plain ? ninv_plain : dp_mulmod(ninv_plain, rinv, c.d);          // N^-1 (R^-1) mod q
            const i64 ninv_mont = (tail != TAIL_NONE) ? Ninv[cur_row] : 0;
            for (int L = threadIdx.x * 2; L < T; L += NTT_THREADS * 2) {
                i64 o[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const double t = smd[PAD(L + e)];
                    double z;
                    if (tail == TAIL_NONE) {
                        z = g.relaxed ? dp_addmask(t, c.d.q) : t;   // relaxed words are balanced residues
                    } else if (tail >= 2) {
                        // redc then reduce yield the canonical residue of t * N^-1 * R^-1 whatever lazy
                        // representatives the intermediate steps took (redc(q) = q reduces to 0)
                        z = dp_mulmod(t, c2, c.d);
                        if (tail >= 3) z = z <= (double)(qq >> 1) ? z : z - c.d.q;
                    } else {
                        z = dp_mulmod(t, ninv_plain, c.d);
                        if (dp_below_fix_limit(z)) z = dp_lazy_fix(z, (u64)t, (u64)ninv_mont, c.d.q);
                        if (tail == 1) z = (z == c.d.q) ? c.d.q : dp_mulmod(z, rinv, c.d);   // redc(q) = q (K.cu:587-606)
                    }
                    o[e] = (tail >= 3) ? (i64)z : dp_to_word(z);
                }
                longlong2 ov;
                ov.x = o[0];
                ov.y = o[1];
                *reinterpret_cast<longlong2 *>(row + tile_gaddr(g, cur_tile, L)) = ov;
            }
        } else {
            if (odd || DP) run_inv_stages<ArithInt<true>, false>(sm, g, cur_tile, c);   // rare: compact stage-by-stage loop
            else run_inv_stages<ArithInt<false>, true>(sm, g, cur_tile, c);
            const i64 ninv = (tail != TAIL_NONE) ? Ninv[cur_row] : 0;
            for (int L = threadIdx.x * 2; L < T; L += NTT_THREADS * 2) {
                i64 t[2] = {sm[PAD(L)], sm[PAD(L + 1)]};
                if (tail != TAIL_NONE) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        i64 z = mm62s(t[e], ninv, c.m.q, c.m.k);
                        if (tail >= 1) z = redc62(z, c.m.q, c.m.k);
                        if (tail >= 2) z = z < qq ? z : z - qq;
                        if (tail >= 3) z = z <= (qq >> 1) ? z : z - qq;
                        t[e] = z;
                    }
                }
                longlong2 v;
                v.x = t[0];
                v.y = t[1];
                *reinterpret_cast<longlong2 *>(row + tile_gaddr(g, cur_tile, L)) = v;
            }
        }
    }
}

template <bool DP, bool RLX>
__global__ void __launch_bounds__(NTT_THREADS, DP ? 6 : 4) ntt_inv_pass_io(const i64 *src, i64 *dst, PassGeom g, RowList rl,
                                                            const i64 *__restrict__ ipsi_br,
                                                            const double *__restrict__ ipsi_dp, const i64 *__restrict__ Ninv,
                                                            int tail, const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                            const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    __shared__ i64 sm[NTT_LDS_WORDS + 1];
    inv_pass_body<DP, RLX>(sm, blockIdx.x, src, dst, g, rl, ipsi_br, ipsi_dp, Ninv, tail, ql, qh, kl, kh);
}

template <bool RLX>
__global__ void __launch_bounds__(NTT_THREADS, NTT_PASS_WAVES) ntt_inv_pass_mixed(const i64 *src, i64 *dst, PassGeom g, ClassLists cl,
                                                                       const i64 *__restrict__ ipsi_br,
                                                                       const double *__restrict__ ipsi_dp,
                                                                       const i64 *__restrict__ Ninv, int tail,
                                                                       const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                                       const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    __shared__ i64 sm[NTT_LDS_WORDS + 1];
    const int b = blockIdx.x;
    if (b < cl.in_blocks) {
        if (b < cl.in_real) inv_pass_body<false, RLX>(sm, b, src, dst, g, cl.in, ipsi_br, ipsi_dp, Ninv, tail, ql, qh, kl, kh);
    } else {
        inv_pass_body<true, RLX>(sm, b - cl.in_blocks, src, dst, g, cl.dp, ipsi_br, ipsi_dp, Ninv, tail, ql, qh, kl, kh);
    }
}

// ------------------------------------------------------------------------------------------------
// inverse STRIDED (last) pass as one radix-2^K register step per column, chain tail included: the mirror
// image of ntt_fwd_cols.  Stage u of the step uses the table entries 2^(K-1-u) .. 2^(K-u) - 1.
// ------------------------------------------------------------------------------------------------
template <class A, int K>
__device__ __forceinline__ void cols_inv_stages(typename A::T (&x)[1 << K], const Ctx &c) {
#pragma unroll
    for (int u = 0; u < K; ++u) {
        const int du = 1 << u;
#pragma unroll
        for (int h = 0; h < (1 << (K - 1 - u)); ++h) {
            const int e0 = h << (u + 1);
            const int idx = (1 << (K - 1 - u)) + h;
            const typename A::W wv = A::tw(c, idx);
#pragma unroll
            for (int t = 0; t < du; ++t) A::inv(c, x[e0 + t], x[e0 + t + du], wv, idx);
        }
    }
}

// the column of lane `lane` of chunk `chunk` of limb (poly, crow): its 2^K words through the trailing stages and the chain
// tail, returned in out[] (word k belongs at coefficient (k << logC) + chunk * NTT_COL_THREADS + lane)
// WS (lf_intt_ws): `a` is the workspace the tiled pass wrote — fp64-class rows as planes, tflags = the row's flag bytes, one per
// tile (= per held word k), set where the tile left the fast form and shipped a third plane (inv_tile16_ws, ckks_ntt_tile16.h)
template <bool DP, int K, bool WS = false>
__device__ __forceinline__ void inv_cols_compute(int poly, int crow, int chunk, const i64 *__restrict__ a, const PassGeom &g,
                                                 const i64 *__restrict__ ipsi_br, const double *__restrict__ ipsi_dp,
                                                 const i64 *__restrict__ Ninv, int tail, const i64 *__restrict__ ql,
                                                 const i64 *__restrict__ qh, const i64 *__restrict__ kl,
                                                 const i64 *__restrict__ kh, i64 (&out)[1 << K],
                                                 const unsigned char *__restrict__ tflags = nullptr) {
    constexpr int R = 1 << K;
    const int logC = g.logN - K;
    Ctx c;
    c.m = load_mod(ql, qh, kl, kh, crow);
    c.tw_mont = ipsi_br + ((i64)crow << g.logN);
    set_aux<DP>(c, ipsi_dp, crow, g.logN);
    c.d = DP ? make_dp_tab(c.m, c.tw_dp) : make_dp(c.m);
    c.relaxed = g.relaxed;
    c.inv_reduce = 0;
    const i64 ninv_mont = (tail != TAIL_NONE) ? Ninv[crow] : 0;
    // wave-uniform base + lane index: the 2^K row addresses stay in SGPRs (a per-lane pointer would pin 2^K 64-bit
    // addresses in VGPRs from the loads to the stores)
    i64 *colu = const_cast<i64 *>(a) + ((i64)(poly * g.rows + crow) << g.logN) + chunk * NTT_COL_THREADS;
    const unsigned lane = threadIdx.x;

    i64 w[R];
    constexpr int FW = (R + 7) / 8;                          // flag bytes of this row's 2^K tiles, eight per scalar load
    u64 fw[FW];
    u64 fany = 0;
    if constexpr (WS) {
        const u64 *fp = reinterpret_cast<const u64 *>(tflags);
#pragma unroll
        for (int i = 0; i < FW; ++i) fw[i] = fp[i];
        if (R < 8) fw[0] &= (1ull << (8 * (R & 7))) - 1;
#pragma unroll
        for (int i = 0; i < FW; ++i) fany |= fw[i];
    }
    if constexpr (WS && DP) {
        const i64 *rowb = a + ((i64)(poly * g.rows + crow) << g.logN);
        const i64 col0 = (i64)chunk * NTT_COL_THREADS;
        const unsigned *lo = reinterpret_cast<const unsigned *>(rowb) + col0;
        const unsigned short *mid = reinterpret_cast<const unsigned short *>(rowb + ((i64)1 << (g.logN - 1))) + col0;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const unsigned l = __builtin_nontemporal_load(uniform_ptr(lo + ((i64)k << logC)) + lane);
            const unsigned m = __builtin_nontemporal_load(uniform_ptr(mid + ((i64)k << logC)) + lane);
            w[k] = (i64)(((u64)m << 32) | (u64)l);
        }
        if (fany != 0) {
            const unsigned short *top = reinterpret_cast<const unsigned short *>(rowb + 3 * ((i64)1 << (g.logN - 2))) + col0;
#pragma unroll
            for (int k = 0; k < R; ++k)
                if (((fw[k >> 3] >> (8 * (k & 7))) & 0xffull) != 0) w[k] |= (i64)((u64)top[((i64)k << logC) + lane] << 48);
        }
    } else if (!WS && DP && (g.pln & PLN_IN)) {   // relaxed stacks of the fused ops: canonical words as two planes
        const i64 *rowb = a + ((i64)(poly * g.rows + crow) << g.logN);
        const i64 col0 = (i64)chunk * NTT_COL_THREADS;
        const unsigned *lo = reinterpret_cast<const unsigned *>(rowb) + col0;
        const unsigned short *hi = reinterpret_cast<const unsigned short *>(rowb + ((i64)1 << (g.logN - 1))) + col0;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const unsigned l = __builtin_nontemporal_load(uniform_ptr(lo + ((i64)k << logC)) + lane);
            const unsigned m = __builtin_nontemporal_load(uniform_ptr(hi + ((i64)k << logC)) + lane);
            w[k] = (i64)(((u64)m << 32) | (u64)l);
        }
    } else {
#pragma unroll
        for (int k = 0; k < R; ++k) w[k] = INV_LD(uniform_row(colu, (i64)k << logC) + lane);
    }
    int odd = 0;
    if (!WS || fany != 0) {   // (WS: the words of unflagged tiles are lazy words by construction)
#pragma unroll
        for (int k = 0; k < R; ++k) {
            // (relaxed: the words come from this library's relaxed tiled pass — canonical / lazy [0, 2q), never negative)
            odd |= ((u64)w[k] >= (u64)c.m.q2);
        }
    }
    if (DP && !odd) {
        double x[R];
#pragma unroll
        for (int k = 0; k < R; ++k) x[k] = dp_from_word(w[k]);
        if (g.relaxed) cols_inv_stages<ArithDpR, K>(x, c);
        else cols_inv_stages<ArithDp, K>(x, c);
        const i64 qq = (i64)c.m.q;
        const double ninv_plain = c.d.q - (double)((c.m.q - 1) >> g.logN);                  // N^-1 mod q
        const double rinv = (double)(u64)((((u128)c.m.k * (u128)c.m.q) + 1) >> 62);         // R^-1 mod q
        const double c2 = g.plain ? ninv_plain : dp_mulmod(ninv_plain, rinv, c.d);          // N^-1 (R^-1) mod q
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double t = x[k];
            double z;
            if (tail == TAIL_NONE) {
                z = g.relaxed ? dp_addmask(dp_reduce_bal(t, c.d), c.d.q) : dp_reduce(t, c.d.q2, c.d.q2inv);
            } else if (tail >= 2) {
                z = dp_mulmod(t, c2, c.d);
                if (tail >= 3) z = z <= (double)(qq >> 1) ? z : z - c.d.q;
            } else {
                const double tr = dp_reduce(t, c.d.q2, c.d.q2inv);   // the reference's lazy word
                z = dp_mulmod(tr, ninv_plain, c.d);
                if (dp_below_fix_limit(z)) z = dp_lazy_fix(z, (u64)tr, (u64)ninv_mont, c.d.q);
                if (tail == 1) z = (z == c.d.q) ? c.d.q : dp_mulmod(z, rinv, c.d);   // redc(q) = q (K.cu:587-606)
            }
            out[k] = (tail >= 3) ? (i64)z : dp_to_word(z);
        }
    } else {
        if (!DP && g.relaxed) {   // residues only (the words are canonical: this library's relaxed tiled pass wrote them)
            cols_inv_stages<ArithShoup, K>(w, c);
#pragma unroll
            for (int k = 0; k < R; ++k)
                out[k] = inv_tail_int(ArithShoup::canon(c, w[k]), tail, ninv_mont, c);
            return;
        }
        if (odd || DP) cols_inv_stages<ArithInt<true>, K>(w, c);
        else cols_inv_stages<ArithInt<false>, K>(w, c);
#pragma unroll
        for (int k = 0; k < R; ++k) out[k] = inv_tail_int(w[k], tail, ninv_mont, c);
    }
}

template <bool DP, int K>
__device__ __forceinline__ void inv_cols_body(int b, i64 *__restrict__ a, const PassGeom &g, const RowList &rl,
                                              const i64 *__restrict__ ipsi_br, const double *__restrict__ ipsi_dp,
                                              const i64 *__restrict__ Ninv, int tail, const i64 *__restrict__ ql,
                                              const i64 *__restrict__ qh, const i64 *__restrict__ kl,
                                              const i64 *__restrict__ kh) {
    constexpr int R = 1 << K;
    const int logC = g.logN - K;
    const int chunks = (1 << logC) / NTT_COL_THREADS;
    const int chunk = b % chunks, r = b / chunks;
    // (the integer divisions run on the VALU: pin their wave-uniform results back into SGPRs)
    const int poly = __builtin_amdgcn_readfirstlane(r % g.batch), crow = __builtin_amdgcn_readfirstlane((int)rl.id[r / g.batch]);
    i64 out[R];
    inv_cols_compute<DP, K>(poly, crow, chunk, g.pln_src ? g.pln_src : a, g, ipsi_br, ipsi_dp, Ninv, tail, ql, qh, kl, kh, out);
    i64 *colu = a + ((i64)(poly * g.rows + crow) << g.logN) + chunk * NTT_COL_THREADS;
    const unsigned lane = threadIdx.x;
#pragma unroll
    for (int k = 0; k < R; ++k) INV_ST(uniform_row(colu, (i64)k << logC) + lane, out[k]);
}

// last pass of an exact inverse transform through a workspace (lf_intt_ws): reads the workspace, writes the tensor
template <bool DP, int K>
__device__ __forceinline__ void inv_cols_ws_body(int b, const i64 *__restrict__ ws, const unsigned char *__restrict__ wflags,
                                                 i64 *__restrict__ a, const PassGeom &g, const RowList &rl,
                                                 const i64 *__restrict__ ipsi_br, const double *__restrict__ ipsi_dp,
                                                 const i64 *__restrict__ Ninv, int tail, const i64 *__restrict__ ql,
                                                 const i64 *__restrict__ qh, const i64 *__restrict__ kl,
                                                 const i64 *__restrict__ kh) {
    constexpr int R = 1 << K;
    const int logC = g.logN - K;
    const int chunks = (1 << logC) / NTT_COL_THREADS;
    const int chunk = b % chunks, r = b / chunks;
    const int poly = __builtin_amdgcn_readfirstlane(r % g.batch), crow = __builtin_amdgcn_readfirstlane((int)rl.id[r / g.batch]);
    i64 out[R];
    inv_cols_compute<DP, K, true>(poly, crow, chunk, ws, g, ipsi_br, ipsi_dp, Ninv, tail, ql, qh, kl, kh, out,
                                  wflags + ((i64)(poly * g.rows + crow) << 6));
    i64 *colu = a + ((i64)(poly * g.rows + crow) << g.logN) + chunk * NTT_COL_THREADS;
    const unsigned lane = threadIdx.x;
#pragma unroll
    for (int k = 0; k < R; ++k) INV_ST(uniform_row(colu, (i64)k << logC) + lane, out[k]);
}

template <int K>
__global__ void __launch_bounds__(NTT_COL_THREADS) ntt_inv_cols_ws(const i64 *__restrict__ ws, const unsigned char *__restrict__ wflags,
                                                                  i64 *__restrict__ a, PassGeom g, ClassLists cl,
                                                                  const i64 *__restrict__ ipsi_br,
                                                                  const double *__restrict__ ipsi_dp, const i64 *__restrict__ Ninv,
                                                                  int tail, const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                                  const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    const int b = blockIdx.x;
    if (b < cl.in_blocks) {
        if (b < cl.in_real) inv_cols_ws_body<false, K>(b, ws, wflags, a, g, cl.in, ipsi_br, ipsi_dp, Ninv, tail, ql, qh, kl, kh);
    } else {
        inv_cols_ws_body<true, K>(b - cl.in_blocks, ws, wflags, a, g, cl.dp, ipsi_br, ipsi_dp, Ninv, tail, ql, qh, kl, kh);
    }
}

template <bool DP, int K>
__global__ void __launch_bounds__(NTT_COL_THREADS) ntt_inv_cols(i64 *__restrict__ a, PassGeom g, RowList rl,
                                                               const i64 *__restrict__ ipsi_br,
                                                               const double *__restrict__ ipsi_dp, const i64 *__restrict__ Ninv,
                                                               int tail, const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                               const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    inv_cols_body<DP, K>(blockIdx.x, a, g, rl, ipsi_br, ipsi_dp, Ninv, tail, ql, qh, kl, kh);
}

template <int K>
__global__ void __launch_bounds__(NTT_COL_THREADS) ntt_inv_cols_mixed(i64 *__restrict__ a, PassGeom g, ClassLists cl,
                                                                     const i64 *__restrict__ ipsi_br,
                                                                     const double *__restrict__ ipsi_dp,
                                                                     const i64 *__restrict__ Ninv, int tail,
                                                                     const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                                     const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    const int b = blockIdx.x;
    if (b < cl.in_blocks) {
        if (b < cl.in_real) inv_cols_body<false, K>(b, a, g, cl.in, ipsi_br, ipsi_dp, Ninv, tail, ql, qh, kl, kh);
    } else {
        inv_cols_body<true, K>(b - cl.in_blocks, a, g, cl.dp, ipsi_br, ipsi_dp, Ninv, tail, ql, qh, kl, kh);
    }
}

// host: launch of the inverse column pass for one arithmetic class (K = number of trailing stages, 1..4)
template <bool DP>
inline void launch_inv_cols(int K, int polys, hipStream_t st, i64 *base, const PassGeom &g, const RowList &rl,
                            const i64 *ipsi_br, const double *ipsi_dp, const i64 *Ninv, int tail, const i64 *ql,
                            const i64 *qh, const i64 *kl, const i64 *kh) {
    const unsigned blocks = (unsigned)polys * (unsigned)rl.n * ((1u << (g.logN - K)) / NTT_COL_THREADS);
#define LF_ICOLS_CASE(KK)                                                                                             \
    case KK:                                                                                                          \
        hipLaunchKernelGGL((ntt_inv_cols<DP, KK>), dim3(blocks), dim3(NTT_COL_THREADS), 0, st, base, g, rl, ipsi_br,  \
                           ipsi_dp, Ninv, tail, ql, qh, kl, kh);                                                      \
        break;
    switch (K) {
        LF_ICOLS_CASE(1)
        LF_ICOLS_CASE(2)
        LF_ICOLS_CASE(3)
        LF_ICOLS_CASE(4)
    }
#undef LF_ICOLS_CASE
}

// (both arithmetic classes of a pass go into one launch whenever both are present: the split form — integer class on
// a side stream — was the round-1 A/B loser; single-class launches only serve transforms that have a single class)
inline ClassLists class_lists(const RowList &in, const RowList &dp, unsigned in_blocks) {
    ClassLists cl;
    cl.in = in;
    cl.dp = dp;
    cl.in_real = (int)in_blocks;
    cl.in_blocks = (int)((in_blocks + 7u) & ~7u);
    return cl;
}

template <int K>
inline void launch_inv_cols_mixed_k(unsigned blocks, hipStream_t st, i64 *base, const PassGeom &g, const ClassLists &cl,
                                    const i64 *ipsi_br, const double *ipsi_dp, const i64 *Ninv, int tail, const i64 *ql,
                                    const i64 *qh, const i64 *kl, const i64 *kh) {
    hipLaunchKernelGGL((ntt_inv_cols_mixed<K>), dim3(blocks), dim3(NTT_COL_THREADS), 0, st, base, g, cl, ipsi_br, ipsi_dp, Ninv,
                       tail, ql, qh, kl, kh);
}

// host: inverse column pass, both classes in one launch
inline void launch_inv_cols_mixed(int K, int polys, hipStream_t st, i64 *base, const PassGeom &g, const RowList &in,
                                  const RowList &dp, const i64 *ipsi_br, const double *ipsi_dp, const i64 *Ninv, int tail,
                                  const i64 *ql, const i64 *qh, const i64 *kl, const i64 *kh) {
    const unsigned per_limb = (unsigned)polys * ((1u << (g.logN - K)) / NTT_COL_THREADS);
    const ClassLists cl = class_lists(in, dp, per_limb * (unsigned)in.n);
    const unsigned blocks = (unsigned)cl.in_blocks + per_limb * (unsigned)dp.n;
    switch (K) {
        case 1: launch_inv_cols_mixed_k<1>(blocks, st, base, g, cl, ipsi_br, ipsi_dp, Ninv, tail, ql, qh, kl, kh); break;
        case 2: launch_inv_cols_mixed_k<2>(blocks, st, base, g, cl, ipsi_br, ipsi_dp, Ninv, tail, ql, qh, kl, kh); break;
        case 3: launch_inv_cols_mixed_k<3>(blocks, st, base, g, cl, ipsi_br, ipsi_dp, Ninv, tail, ql, qh, kl, kh); break;
        case 4: launch_inv_cols_mixed_k<4>(blocks, st, base, g, cl, ipsi_br, ipsi_dp, Ninv, tail, ql, qh, kl, kh); break;
        case 5: launch_inv_cols_mixed_k<5>(blocks, st, base, g, cl, ipsi_br, ipsi_dp, Ninv, tail, ql, qh, kl, kh); break;
    }
}

// (the strided pass runs as a column kernel whenever it has at most 4 stages, logN <= 16; beyond that the LDS-tiled
// strided pass takes over)
// the tiled-pass kernels carry their arithmetic mode (exact lazy words / relaxed residues) as a template parameter
#define LF_LAUNCH_MIXED(KERN, relaxed, ...)                                   \
    do {                                                                      \
        if (relaxed) hipLaunchKernelGGL((KERN<true>), __VA_ARGS__);           \
        else hipLaunchKernelGGL((KERN<false>), __VA_ARGS__);                  \
    } while (0)
#define LF_LAUNCH_CLASS(KERN, DP, relaxed, ...)                               \
    do {                                                                      \
        if (relaxed) hipLaunchKernelGGL((KERN<DP, true>), __VA_ARGS__);       \
        else hipLaunchKernelGGL((KERN<DP, false>), __VA_ARGS__);              \
    } while (0)

// plain canonical twiddles as doubles from the Montgomery table: w = reduce(redc(S))
__global__ void __launch_bounds__(256) twiddle_dp_kernel(const i64 *__restrict__ mont, double *__restrict__ out, i64 N,
                                                         const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                         const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    const int r = blockIdx.y;
    const i64 j = (i64)blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const RowMod m = load_mod(ql, qh, kl, kh, r);
    i64 v = redc62(mont[(i64)r * N + j], m.q, m.k);
    v = v < (i64)m.q ? v : v - (i64)m.q;
    double *row = out + 2 * (i64)r * N;                              // auxiliary rows are 2N words (set_aux)
    if (m.q < SMALL_PRIME_LIMIT) {
        row[j] = j == 0 ? 1.0 / (double)m.q : (double)v;            // slot 0 (= psi^0, never used): 1 / q, see make_dp_tab
        return;
    }
    // integer class: (floor(w * 2^64 / q), w) by 64 steps of binary long division (w < q < 2^60; runs once per table)
    u64 rem = (u64)v, quo = 0;
    for (int i = 0; i < 64; ++i) {
        rem <<= 1;
        quo <<= 1;
        if (rem >= m.q) rem -= m.q, quo |= 1;
    }
    ShoupW *pairs = reinterpret_cast<ShoupW *>(row);
    pairs[j].wq = quo;
    pairs[j].w = (u64)v;
}

}  // namespace
