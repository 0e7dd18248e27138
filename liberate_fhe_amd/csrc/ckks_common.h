// ckks_common.h — device-side scalar arithmetic shared by the gfx950 kernels.
//
// Word model = the reference's 62-bit mode: int64 words, Montgomery radix R = 2^62, lazy values in
// [0, 2q).  The reference builds REDC62 from 31-bit half-words (K.cu:12-59, K.cu =
// src/liberate/ntt/ntt_cuda_kernel.cu); here the same function is evaluated in closed form with
// native 64x64->128 products:
//     mm(a,b) = hi62(a*b) + floor(4*s*q / 2^64) + [lo62(a*b) != 0],   s = lo62(a*b) * k mod 2^62
// which equals (a*b + s*q) / 2^62 exactly for all |a|,|b| < 2^62, i.e. bit-identical outputs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef long long i64;
typedef unsigned long long u64;
typedef unsigned __int128 u128;
typedef __int128 i128;

#define M62 ((1ull << 62) - 1ull)

// capacities compiled into the kernels (reported by lf_limits)
#define KS_MAX_ALPHA 8       // limbs per key-switch digit (= number of special primes, <= 6 in the presets)
#define KS_MAX_K 8           // special primes
#define MAX_LIST_ROWS 250    // limb rows per call
#define LF_BATCH_MAX 8       // operand sets per batched call
#define NTT_TILE_LOG_MAX 12  // log2 of the largest LDS tile; transforms go up to logN = 2 * NTT_TILE_LOG_MAX

// ------------------------------------------------------------------------------------------------
// Scalar arithmetic
// ------------------------------------------------------------------------------------------------

// REDC62 of a signed product (reference K.cu:12-59), any |a|,|b| < 2^62.
static __device__ __forceinline__ i64 mm62s(i64 a, i64 b, u64 q, u64 k) {
    const i128 x = (i128)a * (i128)b;
    const u64 lo = (u64)x;
    const i64 hi = (i64)(x >> 64);
    const u64 xl = lo & M62;
    const i64 xh = (i64)(((u64)hi << 2) | (lo >> 62));
    const u64 s = (xl * k) & M62;
    return xh + (i64)__umul64hi(s << 2, q) + (i64)(xl != 0);
}

// Same for operands known to be non-negative and below 2^61 (butterflies, key products: both in [0, 2q), q < 2^60).
// One operand enters pre-shifted by 2 (4a < 2^63): the 128-bit product 4ab then has (ab >> 62) as its high word and
// 4 * (ab mod 2^62) as its low word, and low * k mod 2^64 IS 4 * s — no masks, no 62-bit realignment.
// (The same function on explicit 32-bit limbs, every multiply-add chain carrying its sum in the 64-bit addend, is 4
// instructions shorter in isolation and measured no faster inside the kernels: 320 vs 311 us for the integer-class
// tiled pass)
static __device__ __forceinline__ i64 mm62u(u64 a, u64 b, u64 q, u64 k) {
    const u128 x = (u128)(a << 2) * (u128)b;
    const u64 lo = (u64)x;              // 4 * xl
    const u64 xh = (u64)(x >> 64);      // (a * b) >> 62
    const u64 s4 = lo * k;              // 4 * ((xl * k) mod 2^62), exactly (mod 2^64)
    return (i64)(xh + __umul64hi(s4, q) + (u64)(lo != 0));
}

// REDC62 of a signed 128-bit value x, |x| < 2^124: (x + ((x*k) mod R) * q) / R — the tail of mm62s without the product.
// The fused key switch accumulates sum_i y_i * c_i over a digit's limbs in 128 bits and reduces ONCE (only the
// residue matters there), instead of one REDC and one conditional subtraction per limb.
static __device__ __forceinline__ i64 redc62_wide(i128 x, u64 q, u64 k) {
    const u64 lo = (u64)x;
    const i64 hi = (i64)(x >> 64);
    const u64 xl = lo & M62;
    const i64 xh = (i64)(((u64)hi << 2) | (lo >> 62));
    const u64 s = (xl * k) & M62;
    return xh + (i64)__umul64hi(s << 2, q) + (i64)(xl != 0);
}

// mont_redc body (K.cu:587-606): (x + ((x*k) mod R) * q) / R for signed x, |x| < 2^62.
static __device__ __forceinline__ i64 redc62(i64 x, u64 q, u64 k) {
    const u64 xl = (u64)x & M62;
    const i64 xh = x >> 62;
    const u64 s = (xl * k) & M62;
    return xh + (i64)__umul64hi(s << 2, q) + (i64)(xl != 0);
}

static __device__ __forceinline__ i64 csub(i64 v, i64 m) { return v < m ? v : v - m; }

// ------------------------------------------------------------------------------------------------
// Shoup multiplication: the integer class of the RELAXED transforms (key switching), where only residues count.
// A twiddle w (PLAIN residue) comes with its quotient wq = floor(w * 2^64 / q); for ANY 64-bit y
//     y * w - floor(y * wq / 2^64) * q   lies in [0, 2q)                      (Shoup / Harvey)
// and the quotient taken from three 32-bit multiplies (high product + the high halves of the two cross products,
// the low product dropped) is short by at most 2, so the value below lies in [0, 4q): 64-bit wrap-around arithmetic,
// 18 instructions against the 46 of the REDC62 butterfly the exact ops must use.  q < 2^60 leaves room for lazy words
// up to 16q; conditional subtractions (4 instructions on the carry flag) keep them below 8q.
// ------------------------------------------------------------------------------------------------
struct ShoupW {
    u64 wq, w;   // quotient, plain twiddle: one 16-byte table entry
};

// y * w mod q as a lazy word in [0, 4q); y any 64-bit value
static __device__ __forceinline__ u64 shoup_mul(u64 y, const ShoupW t, u64 q) {
    const unsigned yh = (unsigned)(y >> 32), yl = (unsigned)y, wh = (unsigned)(t.wq >> 32), wl = (unsigned)t.wq;
    const u64 mid = (u64)__umulhi(yh, wl) + (u64)__umulhi(yl, wh);
    const u64 quo = (u64)yh * (u64)wh + mid;
    return y * t.w - quo * q;
}

// x >= m ? x - m : x for a wave-uniform m: subtract, then select on the borrow (the compiler spends a 64-bit compare
// and a move on top)
static __device__ __forceinline__ u64 csub_u(u64 x, u64 m) {
    const unsigned xl = (unsigned)x, xh = (unsigned)(x >> 32), ml = (unsigned)m, mh = (unsigned)(m >> 32);
    unsigned tl, th;
    // (the borrow-in already occupies the constant bus of the second subtraction: its modulus half sits in a VGPR)
    asm("v_subrev_co_u32 %0, vcc, %4, %2\n\tv_subb_co_u32 %1, vcc, %3, %5, vcc\n\t"
        "v_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %3, vcc"
        : "=&v"(tl), "=&v"(th)
        : "v"(xl), "v"(xh), "s"(ml), "v"(mh)
        : "vcc");
    return ((u64)th << 32) | tl;
}

// U - V + (U < V ? m : 0) for 0 <= U, V < m: subtract, select the modulus on the borrow, add — the value of
// csub(U + m - V, m) in 6 instructions instead of 8
static __device__ __forceinline__ u64 sub_lazy_u(u64 U, u64 V, u64 m) {
    const unsigned ul = (unsigned)U, uh = (unsigned)(U >> 32), vl = (unsigned)V, vh = (unsigned)(V >> 32), ml = (unsigned)m, mh = (unsigned)(m >> 32);
    unsigned tl, th, sl, sh;
    asm("v_sub_co_u32 %0, vcc, %4, %6\n\tv_subb_co_u32 %1, vcc, %5, %7, vcc\n\t"
        "v_cndmask_b32 %2, 0, %8, vcc\n\tv_cndmask_b32 %3, 0, %9, vcc\n\t"
        "v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc"
        : "=&v"(tl), "=&v"(th), "=&v"(sl), "=&v"(sh)
        : "v"(ul), "v"(uh), "v"(vl), "v"(vh), "v"(ml), "v"(mh)
        : "vcc");
    return ((u64)th << 32) | tl;
}

// lazy word below 8q -> canonical residue
static __device__ __forceinline__ u64 shoup_canon(u64 x, u64 q) { return csub_u(csub_u(csub_u(x, q << 2), q << 1), q); }

struct RowMod {
    u64 q, k;
    i64 q2;
};

static __device__ __forceinline__ RowMod load_mod(const i64 *ql, const i64 *qh, const i64 *kl, const i64 *kh, int i) {
    RowMod m;
    m.q = ((u64)qh[i] << 31) | (u64)ql[i];
    m.k = ((u64)kh[i] << 31) | (u64)kl[i];
    m.q2 = (i64)(m.q << 1);
    return m;
}


// ------------------------------------------------------------------------------------------------
// int64 <-> fp64 without the generic 64-bit conversions (the fp64-class kernels hold integers below 2^52)
// ------------------------------------------------------------------------------------------------
// exact int <-> double for 0 <= x < 2^52 with one OR/AND on the high word and one fp64 add
// (the compiler's generic 64-bit conversions cost 4-10 instructions each)
#define DP_MAGIC 4503599627370496.0   // 2^52
static __device__ __forceinline__ double dp_from_word(i64 x) {
    return __longlong_as_double(x | 0x4330000000000000ll) - DP_MAGIC;
}
static __device__ __forceinline__ i64 dp_to_word(double d) {
    return __double_as_longlong(d + DP_MAGIC) & 0x000FFFFFFFFFFFFFll;
}

// a word below 2^48 kept as two PLANES (low 32 bits | high 16 bits: 6 bytes instead of 8, lf_key_planes and the extended
// digits of a key switch): the double is assembled in registers — exponent | high half in the upper dword, the low word
// below, minus 2^52 — for the price of the raw word's conversion
static __device__ __forceinline__ double dp_from_planes(unsigned lo, unsigned hi16) {
    return __longlong_as_double((i64)(((u64)(0x43300000u | hi16) << 32) | (u64)lo)) - DP_MAGIC;
}
typedef unsigned lf_u4_t __attribute__((ext_vector_type(4)));
typedef unsigned lf_u2_t __attribute__((ext_vector_type(2)));

// exact signed int -> double for |x| < 2^51: one 64-bit integer add and one fp64 add (the generic signed 64-bit
// conversion costs ~9 instructions)
static __device__ __forceinline__ double dp_from_signed(i64 x) {
    return __longlong_as_double(x + 0x4338000000000000ll) - 6755399441055744.0;   // 2^52 + 2^51
}


// ------------------------------------------------------------------------------------------------
// Host side: the format of library-internal scratch that crosses NATIVE CALLS (ckks_hip.hip)
// ------------------------------------------------------------------------------------------------
// Where a kernel hands fp64-class rows to the next one as 6-byte planes (the extended digits of a key switch, cc_mult's operand
// stack, the workspace of a transform), producer and consumer may be launched by DIFFERENT native calls (lf_ks_fwd -> lf_ks_tail,
// lf_cc_mult_evk_pre -> _post, lf_ntt_pass_ws 1 -> 2) and both decide the format from lf_tune's process-wide knobs.  The
// producer notes the format it wrote over [p, p + bytes); the consumer states the format it is about to read and gets
// LF_ERR_STATE — nothing launched — when a noted range it overlaps was written in another one (a knob flipped between the
// halves: only notes made under an EARLIER knob setting count — a mismatching note of the current setting is a stale one on memory
// the caller's allocator recycled).  Ranges nobody noted are taken on trust (a caller may fill scratch by hand).  Mutex-protected,
// 256 ranges, oldest out.
#define LF_FMT_RAW 0
#define LF_FMT_PLANES 1
#define LF_FMT_WS_SPLIT0 2      // lf_ntt_ws workspace written by a column pass without / with the extra stage
#define LF_FMT_WS_SPLIT1 3
void lf_fmt_note(const void *p, size_t bytes, int fmt);
void lf_fmt_epoch_bump();   // lf_tune changed a format knob
int lf_fmt_expect(const void *p, size_t bytes, int fmt);   // 0, or LF_ERR_STATE

static inline int lf_set_device(int device) {
    if (device >= 0) {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}
