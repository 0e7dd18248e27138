// ckks_ks.hip — fused hybrid key-switch core: lf_ks_core.
//
// Replaces, for ring degrees with a two-pass NTT (logN >= 13), the chain
//     lf_ks_extend -> lf_ntt(batch = parts) -> lf_ks_inner -> lf_intt(batch = 2, tail 2)
// (reference: ckks_engine.py extend 707-743, ntt 919, mont_mult x2 931-934, sum over parts 832-840,
// intt_exit_reduce 847-848) by a pipeline that never materialises the extended digits in the
// coefficient domain and does every 40-bit-limb product with one fp64 modular multiplication:
//
//   K2  ks_ext_pass1     : every (digit p, target limb r, column tile): the tile's coefficients are
//                          computed on the fly from the digit's Garner words (extend), taken through the
//                          strided NTT pass and written once                [parts x rows x N words out]
//   P2  ntt_fwd_pass     : the stock contiguous pass, in place
//   K3  ks_inner2_kernel : per coefficient, the digits' products with the two key polynomials accumulated
//                          in registers; the key is streamed exactly once
//   K4  ntt_inv_pass_io  : the stock inverse passes (relaxed, tail 2) -> canonical coefficients
// (Variants that fuse P2 with K3 — transform a tile, multiply, accumulate over the digits in registers — were
// measured twice: round 1 on the 8-words-per-thread tile 1.6x SLOWER; round 2 on the 16-words tile, with the contiguous
// inverse pass fused in as well (one block = one (ciphertext, limb, tile), 154 / 168 VGPRs, 3 blocks per CU): gold 231 us
// against 224 us for the three launches it replaces, silver 152 against 95 us (170 blocks for 256 CUs), batches of 16
// and 64 ciphertexts +1-2 %.  Too few, too long blocks for one ciphertext, and no gain from the 490 MB of HBM traffic it
// saves once the grid is full: these passes are bound by their LDS exchanges and barriers, not by HBM; see DESIGN.md.)
//
// All of it is "relaxed" arithmetic: only residues matter because the consumer (mod-down) needs the
// canonical coefficients, which the inverse chain's tail produces.  For limbs with a prime below 2^41
// the extension and the inner product are done in the PLAIN domain with fp64 FMAs — ext = sum y_i L_{i-1}
// (no Montgomery factor), NTT with plain twiddles, times the key word k*R gives (ext*k)*R, exactly the
// Montgomery-form residue the reference's REDC(ext*R * k*R) yields — so each product is ONE fp64
// modular multiplication.  60-bit limbs keep the reference's Montgomery integer arithmetic.
#include "../../include/ckks_hip.h"
#include "ckks_ntt_core.h"
#include "ckks_ntt_tile16.h"

#define KS_WORDS ((1 << NTT_TILE_LOG_MAX) / NTT_THREADS)   // tile words owned by one thread (8)

// streaming stores of the extension kernels (ckks_ntt_core.h: "Streaming accesses")
#define KS_ST(p, v)                                                  \
    do {                                                             \
        if (NT_KS_EXT) __builtin_nontemporal_store((i64)(v), p);     \
        else *(p) = (v);                                             \
    } while (0)

// the same word into a row in planes format (u32 low[N] at byte 0, u16 high[N] at byte 4 N); rowb = the row's first word
#define KS_ST_PL(rowb, N, idx, v)                                                                          \
    do {                                                                                                   \
        const i64 v_ = (i64)(v);                                                                           \
        unsigned *lo_ = uniform_ptr(reinterpret_cast<unsigned *>(rowb) + (idx));                           \
        unsigned short *hi_ = uniform_ptr(reinterpret_cast<unsigned short *>((rowb) + ((N) >> 1)) + (idx)); \
        if (NT_KS_EXT) {                                                                                   \
            __builtin_nontemporal_store((unsigned)v_, lo_ + lane);                                         \
            __builtin_nontemporal_store((unsigned short)(v_ >> 32), hi_ + lane);                           \
        } else {                                                                                           \
            lo_[lane] = (unsigned)v_;                                                                      \
            hi_[lane] = (unsigned short)(v_ >> 32);                                                        \
        }                                                                                                  \
    } while (0)

namespace {

struct KsGeom {
    int logN, tl, S1;
    int rows;        // target limbs on this device (with special)
    int nparts;      // digits
    i64 N;
    int nct;         // ciphertexts switched under the same key in this call (lf_ks_core_batch)
    i64 state_stride;   // words between their digit states
    // relinearisation inside cc_mult: the limbs a digit is made of are not extended (the inner product takes them from
    // the NTT-domain operands, RelinFold); own[r] = digit (numbered from the first digit of the key switch, this call
    // starting at p0) that limb r belongs to, 255 for the special limbs; nullptr: every (digit, limb) pair is extended
    const unsigned char *own;
    int p0;
    int planes;      // fp64-class rows of tmp in planes format (digit_planes(), ckks_ntt_tile16.h fwd_tile16<.., PLN>)
};

// ---- K2: extend + strided NTT pass -----------------------------------------------------------------
// desc[p] = {row_start, alpha, e_off}; E (Montgomery consts, int class) / Ed (plain consts as doubles,
// fp64 class) hold, at [e_off + i*rows + r], L_{i-1} R^2 mod q_r resp. L_{i-1} mod q_r (i = 0: R^2, 1).
template <bool DP>
__device__ __forceinline__ void ks_ext_body(i64 *sm, int b, const i64 *__restrict__ state, i64 *__restrict__ tmp,
                                            const KsGeom &kg, const RowList &rl, const i64 *__restrict__ desc,
                                            const i64 *__restrict__ E, const double *__restrict__ Ed,
                                            const i64 *__restrict__ psi_br, const double *__restrict__ psi_dp,
                                            const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                            const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    const int T = 1 << kg.tl;
    const int tiles = 1 << (kg.logN - kg.tl);
    // ciphertexts of a batch are the outermost index: each one is the single-ciphertext grid over its own state
    const int per_ct = tiles * kg.nparts * rl.n;
    const int ct = b / per_ct;
    b -= ct * per_ct;
    state += (i64)ct * kg.state_stride;
    tmp += ((i64)ct * kg.nparts * kg.rows) << kg.logN;
    // the blocks of one (digit, tile) pair differ in the target limb and re-read the same digit columns: they are
    // placed on ONE XCD (blocks b, b + 8, .. share an XCD), so those columns are fetched into one L2, once
    int ri, pt;
    if (((tiles * kg.nparts) & 7) == 0) {
        const int x = b & 7, r = b >> 3;
        ri = r % rl.n;
        pt = (r / rl.n) * 8 + x;
    } else {
        ri = b % rl.n;
        pt = b / rl.n;
    }
    const int tile = pt % tiles, p = pt / tiles;
    const int crow = rl.id[ri];
    if (kg.own != nullptr && (int)kg.own[crow] == kg.p0 + p) return;   // the digit's own limb: nothing to extend
    const PassGeom g{kg.logN, kg.tl, 1, kg.S1, 0, kg.tl - kg.S1, kg.rows, kg.nparts, 1, 0, 0, nullptr, 0, 0};

    Ctx c;
    c.m = load_mod(ql, qh, kl, kh, crow);
    c.tw_mont = psi_br + ((i64)crow << kg.logN);
    set_aux<DP>(c, psi_dp, crow, kg.logN);
    c.d = DP ? make_dp_tab(c.m, c.tw_dp) : make_dp(c.m);
    c.relaxed = 1;
    c.inv_reduce = 0;
    // desc[p] = {row_start, alpha | wide << 8, e_off}; wide = the digit's words exceed 53 bits (a digit made of
    // 60-bit primes, i.e. the base-prime digit): they are split into 31-bit halves before entering fp64
    const int row_start = (int)desc[p * 3 + 0], alpha = (int)desc[p * 3 + 1] & 0xff;
    const bool wide = ((int)desc[p * 3 + 1] >> 8) & 1;
    const i64 e_off = desc[p * 3 + 2] + crow;
    i64 *row = tmp + ((i64)(p * kg.rows + crow) << kg.logN);

    if (DP) {
        double *smd = reinterpret_cast<double *>(sm);
        double cst[KS_MAX_ALPHA], cst31[KS_MAX_ALPHA];
#pragma unroll
        for (int i = 0; i < KS_MAX_ALPHA; ++i) {
            cst[i] = i < alpha ? Ed[e_off + (i64)i * kg.rows] : 0.0;
            cst31[i] = wide ? dp_mulmod(cst[i], 2147483648.0, c.d) : 0.0;
        }
        for (int L = threadIdx.x * 2; L < T; L += NTT_THREADS * 2) {
            const i64 j = tile_gaddr(g, tile, L);
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int i = 0; i < KS_MAX_ALPHA; ++i) {
                if (i < alpha) {
                    const longlong2 y = *reinterpret_cast<const longlong2 *>(state + (i64)(row_start + i) * kg.N + j);
                    if (!wide) {
                        a0 += dp_mulmod_bal(dp_from_signed(y.x), cst[i], c.d);    // signed digit words (|y| < 2^43): the formula is sign-agnostic
                        a1 += dp_mulmod_bal(dp_from_signed(y.y), cst[i], c.d);
                    } else {
                        // 31-bit halves through the native 32-bit conversions
                        a0 += dp_mulmod_bal((double)(int)(y.x >> 31), cst31[i], c.d) + dp_mulmod_bal((double)(unsigned)(y.x & 0x7fffffffll), cst[i], c.d);
                        a1 += dp_mulmod_bal((double)(int)(y.y >> 31), cst31[i], c.d) + dp_mulmod_bal((double)(unsigned)(y.y & 0x7fffffffll), cst[i], c.d);
                    }
                }
            }
            smd[PAD(L)] = a0;        // |.| < alpha * q (balanced terms)
            smd[PAD(L + 1)] = a1;
        }
        lds_barrier();
        run_fwd_stages<ArithDpR, true>(smd, g, tile, c);
        for (int L = threadIdx.x * 2; L < T; L += NTT_THREADS * 2) {
            longlong2 o;
            o.x = dp_to_word(dp_reduce(smd[PAD(L)], c.d.q, c.d.qinv));
            o.y = dp_to_word(dp_reduce(smd[PAD(L + 1)], c.d.q, c.d.qinv));
            const i64 j = tile_gaddr(g, tile, L);   // even: words j, j + 1 are neighbours
            if (kg.planes) {   // 8 + 4 bytes for the pair (digit_planes(): fwd_tile16<.., PLN> and the inner product read planes)
                const lf_u2_t lo = {(unsigned)o.x, (unsigned)o.y};
                *reinterpret_cast<lf_u2_t *>(reinterpret_cast<unsigned *>(row) + j) = lo;
                *reinterpret_cast<unsigned *>(reinterpret_cast<unsigned short *>(row + (kg.N >> 1)) + j) =
                    (unsigned)((u64)o.x >> 32) | ((unsigned)((u64)o.y >> 32) << 16);
            } else {
                *reinterpret_cast<longlong2 *>(row + j) = o;
            }
        }
    } else {
        i64 cst[KS_MAX_ALPHA];
#pragma unroll
        for (int i = 0; i < KS_MAX_ALPHA; ++i) cst[i] = i < alpha ? E[e_off + (i64)i * kg.rows] : 0;
        for (int L = threadIdx.x * 2; L < T; L += NTT_THREADS * 2) {
            const i64 j = tile_gaddr(g, tile, L);
            i64 a0, a1;
            if (wide && alpha > 1) {
                // several 60-bit limbs in one digit (no preset has that): term by term, as the reference extends
                a0 = a1 = 0;
#pragma unroll
                for (int i = 0; i < KS_MAX_ALPHA; ++i) {
                    if (i < alpha) {
                        const longlong2 y = *reinterpret_cast<const longlong2 *>(state + (i64)(row_start + i) * kg.N + j);
                        const i64 t0 = mm62s(y.x, cst[i], c.m.q, c.m.k), t1 = mm62s(y.y, cst[i], c.m.q, c.m.k);
                        a0 = i == 0 ? t0 : csub(a0 + t0, c.m.q2);
                        a1 = i == 0 ? t1 : csub(a1 + t1, c.m.q2);
                    }
                }
            } else {
                // sum_i y_i * (L_{i-1} R^2 mod q) in 128 bits, ONE REDC: |y_i| < 2^44 for digits of 40-bit limbs
                // (<= 8 terms, constants < 2^60: |sum| < 2^107), or a single term |y| < 2^61 — the result lies in
                // (-2^59, q + 2^59), inside the (-2q, 2q) the fold below expects
                i128 x0 = 0, x1 = 0;
#pragma unroll
                for (int i = 0; i < KS_MAX_ALPHA; ++i) {
                    if (i < alpha) {
                        const longlong2 y = *reinterpret_cast<const longlong2 *>(state + (i64)(row_start + i) * kg.N + j);
                        x0 += (i128)y.x * (i128)cst[i];
                        x1 += (i128)y.y * (i128)cst[i];
                    }
                }
                a0 = redc62_wide(x0, c.m.q, c.m.k);
                a1 = redc62_wide(x1, c.m.q, c.m.k);
            }
            sm[PAD(L)] = a0 < 0 ? a0 + c.m.q2 : a0;       // residues only: fold into [0, 2q)
            sm[PAD(L + 1)] = a1 < 0 ? a1 + c.m.q2 : a1;
        }
        lds_barrier();
        run_fwd_stages<ArithShoup, true>(sm, g, tile, c);            // residues only: Shoup products, lazy words < 8q
        for (int L = threadIdx.x * 2; L < T; L += NTT_THREADS * 2) {
            longlong2 o;
            o.x = ArithShoup::canon(c, sm[PAD(L)]);
            o.y = ArithShoup::canon(c, sm[PAD(L + 1)]);
            *reinterpret_cast<longlong2 *>(row + tile_gaddr(g, tile, L)) = o;
        }
    }
}

template <bool DP>
__global__ void __launch_bounds__(NTT_THREADS, DP ? 6 : 4) ks_ext_pass1(const i64 *__restrict__ state, i64 *__restrict__ tmp,
                                                                          KsGeom kg, RowList rl, const i64 *__restrict__ desc,
                                                                          const i64 *__restrict__ E, const double *__restrict__ Ed,
                                                                          const i64 *__restrict__ psi_br,
                                                                          const double *__restrict__ psi_dp,
                                                                          const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                                          const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    __shared__ i64 sm[NTT_LDS_WORDS + 1];
    ks_ext_body<DP>(sm, blockIdx.x, state, tmp, kg, rl, desc, E, Ed, psi_br, psi_dp, ql, qh, kl, kh);
}

// both arithmetic classes in one launch (integer-class blocks first), see ntt_fwd_pass_mixed
#define KS_EXT_WAVES 8   // waves per SIMD the kernel is compiled for: 63 VGPRs, no spill, four blocks per CU (LDS) instead of
                         // the three of the 6-wave build (80 VGPRs); measured at gold: rotate 409-414 -> 402 us, cc_mult 557 -> 540-545 us
__global__ void __launch_bounds__(NTT_THREADS, KS_EXT_WAVES) ks_ext_pass1_mixed(const i64 *__restrict__ state, i64 *__restrict__ tmp,
                                                                       KsGeom kg, ClassLists cl, const i64 *__restrict__ desc,
                                                                       const i64 *__restrict__ E, const double *__restrict__ Ed,
                                                                       const i64 *__restrict__ psi_br,
                                                                       const double *__restrict__ psi_dp,
                                                                       const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                                       const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    __shared__ i64 sm[NTT_LDS_WORDS + 1];
    const int b = blockIdx.x;
    if (b < cl.in_blocks) {
        if (b < cl.in_real) ks_ext_body<false>(sm, b, state, tmp, kg, cl.in, desc, E, Ed, psi_br, psi_dp, ql, qh, kl, kh);
    } else {
        ks_ext_body<true>(sm, b - cl.in_blocks, state, tmp, kg, cl.dp, desc, E, Ed, psi_br, psi_dp, ql, qh, kl, kh);
    }
}

// ---- K2, column form (used for S1 = logN - 12 <= 3): extend + the S1 leading stages as ONE register step per column ---------
// Thread = one column of one (digit, target limb): its 2^S1 words sit N / 2^S1 apart and every one of them is the
// extension sum over the digit's alpha limbs at that coefficient, so the step needs no LDS, no barrier and no
// per-lane twiddle (the 2^S1 - 1 twiddles are the table entries 1 .. 2^S1 - 1, scalar loads) — the shape of
// ntt_fwd_cols with the rescale-free extension as its source.  A wave reads / writes 512 contiguous bytes per row.
// The blocks of one (digit, column chunk) pair differ in the target limb and re-read the same digit words: they sit
// on one XCD (as in ks_ext_body).
template <bool DP, int K>
__device__ __forceinline__ void ks_ext_cols_body(int b, const i64 *__restrict__ state, i64 *__restrict__ tmp, const KsGeom &kg,
                                                 const RowList &rl, const i64 *__restrict__ desc, const i64 *__restrict__ E,
                                                 const double *__restrict__ Ed, const i64 *__restrict__ psi_br,
                                                 const double *__restrict__ psi_dp, const i64 *__restrict__ ql,
                                                 const i64 *__restrict__ qh, const i64 *__restrict__ kl,
                                                 const i64 *__restrict__ kh) {
    constexpr int R = 1 << K;
    const int logC = kg.logN - K;
    const int chunks = (1 << logC) / NTT_COL_THREADS;
    const int per_ct = chunks * kg.nparts * rl.n;
    const int ct = b / per_ct;
    b -= ct * per_ct;
    int ri, pc;
    if (((chunks * kg.nparts) & 7) == 0) {
        const int x = b & 7, r = b >> 3;
        ri = r % rl.n;
        pc = (r / rl.n) * 8 + x;
    } else {
        ri = b % rl.n;
        pc = b / rl.n;
    }
    // (integer divisions run on the VALU: pin the wave-uniform coordinates back into SGPRs)
    const int chunk = __builtin_amdgcn_readfirstlane(pc % chunks), p = __builtin_amdgcn_readfirstlane(pc / chunks);
    const int crow = __builtin_amdgcn_readfirstlane((int)rl.id[ri]);
    const int ctu = __builtin_amdgcn_readfirstlane(ct);
    if (kg.own != nullptr && (int)kg.own[crow] == kg.p0 + p) return;   // the digit's own limb: nothing to extend

    Ctx c;
    c.m = load_mod(ql, qh, kl, kh, crow);
    c.tw_mont = psi_br + ((i64)crow << kg.logN);
    set_aux<DP>(c, psi_dp, crow, kg.logN);
    c.d = DP ? make_dp_tab(c.m, c.tw_dp) : make_dp(c.m);
    c.relaxed = 1;
    c.inv_reduce = 0;
    // (pinned to SGPRs: the digit loop below runs on the scalar unit and feeds SGPR row pointers)
    const int row_start = __builtin_amdgcn_readfirstlane((int)desc[p * 3 + 0]);
    const int alpha = __builtin_amdgcn_readfirstlane((int)desc[p * 3 + 1] & 0xff);
    const bool wide = __builtin_amdgcn_readfirstlane(((int)desc[p * 3 + 1] >> 8) & 1) != 0;
    const i64 e_off = desc[p * 3 + 2] + crow;
    const unsigned lane = threadIdx.x;
    const i64 *src = state + (i64)ctu * kg.state_stride + (i64)row_start * kg.N + chunk * NTT_COL_THREADS;
    i64 *rowb = tmp + ((((i64)ctu * kg.nparts + p) * kg.rows + crow) << kg.logN);
    const i64 col0 = (i64)chunk * NTT_COL_THREADS;
    i64 *dst = rowb + col0;

    if (DP) {
        // the digit loop is a RUNTIME loop with wave-uniform constants (scalar loads): at most R loads in flight beside the
        // R accumulated words — the unrolled form kept 8 x R loads alive (90 VGPRs at R = 16, measured slower at logN 16)
        double x[R];
        // Horner form (desc[p][1] >> 16 = offset of the table of the digit's OWN primes m_i mod q_r behind the L table, 0: none):
        //     y_0 + L_0 y_1 + L_1 y_2 + ..  =  y_0 + m_0 (y_1 + m_1 (y_2 + ..)),   L_i = m_0 .. m_i
        // alpha - 1 modular products per word instead of alpha (the first constant of the sum form is 1: a wasted product)
        const int hoff = __builtin_amdgcn_readfirstlane((int)(desc[p * 3 + 1] >> 16));
        if (!wide && hoff != 0) {
            {
                const i64 *rowl = src + ((i64)(alpha - 1) << kg.logN);
#pragma unroll
                for (int k = 0; k < R; ++k) x[k] = dp_from_signed(rowl[((i64)k << logC) + lane]);
            }
            for (int i = alpha - 2; i >= 0; --i) {
                const double mi = Ed[hoff + e_off + (i64)i * kg.rows];
                const i64 *rowi = src + ((i64)i << kg.logN);   // wave-uniform (scalar base + lane offset in the loads below)
#pragma unroll
                for (int k = 0; k < R; ++k)   // |x| < 2^44 throughout: balanced product (< q / 2) + a signed digit word (< 2^43)
                    x[k] = dp_from_signed(rowi[((i64)k << logC) + lane]) + dp_mulmod_bal(x[k], mi, c.d);
            }
            cols_fwd_stages<ArithDpR, K>(x, c);
            if (kg.planes) {
#pragma unroll
                for (int k = 0; k < R; ++k) KS_ST_PL(rowb, kg.N, col0 + ((i64)k << logC), dp_to_word(dp_reduce(x[k], c.d.q, c.d.qinv)));
                return;
            }
#pragma unroll
            for (int k = 0; k < R; ++k) KS_ST(uniform_row(dst, (i64)k << logC) + lane, dp_to_word(dp_reduce(x[k], c.d.q, c.d.qinv)));
            return;
        }
#pragma unroll
        for (int k = 0; k < R; ++k) x[k] = 0.0;
        for (int i = 0; i < alpha; ++i) {
            const double cst = Ed[e_off + (i64)i * kg.rows];
            const i64 *rowi = src + ((i64)i << kg.logN);   // wave-uniform (scalar base + lane offset in the loads below)
            if (!wide) {
#pragma unroll
                for (int k = 0; k < R; ++k)   // signed digit words (|y| < 2^43): the formula is sign-agnostic
                    x[k] += dp_mulmod_bal(dp_from_signed(rowi[((i64)k << logC) + lane]), cst, c.d);
            } else {
                const double cst31 = dp_mulmod(cst, 2147483648.0, c.d);
#pragma unroll
                for (int k = 0; k < R; ++k) {   // 60-bit digit words: 31-bit halves through the native 32-bit conversions
                    const i64 y = rowi[((i64)k << logC) + lane];
                    x[k] += dp_mulmod_bal((double)(int)(y >> 31), cst31, c.d) + dp_mulmod_bal((double)(unsigned)(y & 0x7fffffffll), cst, c.d);
                }
            }
        }
        cols_fwd_stages<ArithDpR, K>(x, c);          // |x| < alpha * q on the way in (balanced terms)
        if (kg.planes) {
#pragma unroll
            for (int k = 0; k < R; ++k) KS_ST_PL(rowb, kg.N, col0 + ((i64)k << logC), dp_to_word(dp_reduce(x[k], c.d.q, c.d.qinv)));
            return;
        }
#pragma unroll
        for (int k = 0; k < R; ++k) KS_ST(uniform_row(dst, (i64)k << logC) + lane, dp_to_word(dp_reduce(x[k], c.d.q, c.d.qinv)));
    } else {
        i64 w[R];
        if (wide && alpha > 1) {   // several 60-bit limbs in one digit (no preset has that): term by term
#pragma unroll
            for (int k = 0; k < R; ++k) w[k] = 0;
            for (int i = 0; i < alpha; ++i) {
                const i64 cst = E[e_off + (i64)i * kg.rows];
                const i64 *rowi = src + ((i64)i << kg.logN);   // wave-uniform (scalar base + lane offset in the loads below)
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const i64 t = mm62s(rowi[((i64)k << logC) + lane], cst, c.m.q, c.m.k);
                    w[k] = i == 0 ? t : csub(w[k] + t, c.m.q2);
                }
            }
        } else {                   // sum_i y_i * (L_{i-1} R^2 mod q) in 128 bits, ONE REDC (see ks_ext_body); 8 words at a time
#pragma unroll
            for (int g0 = 0; g0 < R; g0 += 8) {
                constexpr int G = R < 8 ? R : 8;
                i128 acc[G];
#pragma unroll
                for (int k = 0; k < G; ++k) acc[k] = 0;
                for (int i = 0; i < alpha; ++i) {
                    const i64 cst = E[e_off + (i64)i * kg.rows];
                    const i64 *rowi = src + ((i64)i << kg.logN);   // wave-uniform (scalar base + lane offset in the loads below)
#pragma unroll
                    for (int k = 0; k < G; ++k)
                        acc[k] += (i128)rowi[((i64)(g0 + k) << logC) + lane] * (i128)cst;
                }
#pragma unroll
                for (int k = 0; k < G; ++k) w[g0 + k] = redc62_wide(acc[k], c.m.q, c.m.k);
            }
        }
#pragma unroll
        for (int k = 0; k < R; ++k) w[k] = w[k] < 0 ? w[k] + c.m.q2 : w[k];   // residues only: fold into [0, 2q)
        cols_fwd_stages<ArithShoup, K>(w, c);
#pragma unroll
        for (int k = 0; k < R; ++k) KS_ST(dst + ((i64)k << logC) + lane, ArithShoup::canon(c, w[k]));
    }
}

#ifndef KS_EXT_COLS_WAVES
#define KS_EXT_COLS_WAVES 4
#endif
template <int K>
__global__ void __launch_bounds__(NTT_COL_THREADS) __attribute__((amdgpu_waves_per_eu(K == 5 ? 2 : KS_EXT_COLS_WAVES))) ks_ext_cols_mixed(const i64 *__restrict__ state, i64 *__restrict__ tmp,
                                                                     KsGeom kg, ClassLists cl, const i64 *__restrict__ desc,
                                                                     const i64 *__restrict__ E, const double *__restrict__ Ed,
                                                                     const i64 *__restrict__ psi_br,
                                                                     const double *__restrict__ psi_dp,
                                                                     const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                                     const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    const int b = blockIdx.x;
    if (b < cl.in_blocks) {
        if (b < cl.in_real) ks_ext_cols_body<false, K>(b, state, tmp, kg, cl.in, desc, E, Ed, psi_br, psi_dp, ql, qh, kl, kh);
    } else {
        ks_ext_cols_body<true, K>(b - cl.in_blocks, state, tmp, kg, cl.dp, desc, E, Ed, psi_br, psi_dp, ql, qh, kl, kh);
    }
}

// ---- the key in PLANES format (lf_key_planes): the HBM-bound launch reads fewer key bytes ------------------------------
// ks_inner2_kernel streams the key at the rate HBM delivers (gold cc_mult: 687 MB in 112 us, 450 MB of them key words), so
// only fewer BYTES make it shorter.  A word of an fp64-class key row is a residue below 2^41.  lf_key_planes stores the two
// components (b, a) of such a row, once per key, as
//     slot of component 0 (8 N bytes):         N / 2 groups of 16 bytes  { lo32 b[j], lo32 b[j+1], lo32 a[j], lo32 a[j+1] }
//     slot of component 1, first 4 N bytes:    N / 2 groups of  8 bytes  { hi16 b[j], hi16 b[j+1], hi16 a[j], hi16 a[j+1] }
// of the CANONICAL residues — 12 N bytes per row pair instead of 16 N; integer-class rows stay raw words in their slots.
// A thread of the inner product (two coefficients) then issues, per digit, ONE 16-byte and ONE 8-byte key load for both
// components where the raw layout needs two 16-byte loads: fewer bytes AND no more load instructions, at the register
// count of the raw kernel (a first version with four coefficients per thread and per-component planes read 25 % fewer key
// bytes at 96 .. 256 VGPRs and was no faster).  The double is assembled in registers — exponent | high word in the upper
// dword, the low word below, minus 2^52 — for the price of the raw word's conversion.  Same sums modulo q: same outputs.
// lf_key_planes: one row pair per blockIdx.y, two coefficients per thread
__global__ void __launch_bounds__(256) key_planes_kernel(const i64 *__restrict__ src_b, const i64 *__restrict__ src_a,
                                                         i64 *__restrict__ dst_b, i64 *__restrict__ dst_a, i64 N,
                                                         const i64 *__restrict__ ql, const i64 *__restrict__ qh) {
    const int r = blockIdx.y;
    const i64 j = ((i64)blockIdx.x * 256 + threadIdx.x) * 2;
    if (j >= N) return;
    const i64 q = (qh[r] << 31) | ql[r];
    const longlong2 b = *reinterpret_cast<const longlong2 *>(src_b + (i64)r * N + j);
    const longlong2 a = *reinterpret_cast<const longlong2 *>(src_a + (i64)r * N + j);
    if ((u64)q >= SMALL_PRIME_LIMIT) {
        *reinterpret_cast<longlong2 *>(dst_b + (i64)r * N + j) = b;
        *reinterpret_cast<longlong2 *>(dst_a + (i64)r * N + j) = a;
        return;
    }
    const i64 w[4] = {b.x, b.y, a.x, a.y};
    unsigned lo[4], hi[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        i64 c = w[v] % q;            // any word (lazy, signed-lazy): its canonical residue; once per key
        c = c < 0 ? c + q : c;
        lo[v] = (unsigned)c;
        hi[v] = (unsigned)(c >> 32);
    }
    const lf_u4_t l = {lo[0], lo[1], lo[2], lo[3]};
    const lf_u2_t h = {hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16)};
    *reinterpret_cast<lf_u4_t *>(reinterpret_cast<unsigned *>(dst_b + (i64)r * N) + 2 * j) = l;
    *reinterpret_cast<lf_u2_t *>(reinterpret_cast<unsigned *>(dst_a + (i64)r * N) + j) = h;
}

// The key (gold: 429 MB per key switch) is read exactly once: nontemporal loads (global_load_dwordx4 .. nt) keep it
// from displacing the digits, which the forward pass has just written, out of L2 / Infinity Cache
// (measured at gold: 124.8 -> 96.2 us together with one 16-byte column per thread instead of two).
#define KI_COLS 1
#ifndef KI_UNROLL
#define KI_UNROLL 2   // digits of the fp64-class inner product in flight per thread (measured: profiles / LAB_NOTES round 4)
#endif
__device__ __forceinline__ longlong2 ld_nt(const i64 *p) {
    longlong2 v;
    v.x = __builtin_nontemporal_load(p);
    v.y = __builtin_nontemporal_load(p + 1);
    return v;
}

// ---- K3: inner product with the key, summed over digits, on the relaxed NTT-domain words --------------
// fp64 rows: words are plain canonical residues x; x * (k R) mod q is the Montgomery-form product the
// reference's REDC(xR * kR) yields, one fp64 modular multiplication each.  Integer rows: REDC as the reference.
// grid = (N / 1024, rows); the digits' products are accumulated in registers, the key is read exactly once.
// NCT ciphertexts switched under the same key share every key word: it is read once for all of them.
// FOLD (relinearisation inside cc_mult, lf_relin_*): the first two components of the tensor product never get an inverse
// transform of their own.  Dividing by P is linear and P * d vanishes modulo every special prime, so
//     moddown(s) + d  ==  moddown(s + P * d  on the ordinary rows)
// and the addends enter HERE, in the NTT domain, from the four transformed operand polynomials:
//     s0 += P (x0 y0),   s1 += P (x0 y1 + x1 y0)          (ckks_engine.py:1095-1101, 1135-1140)
// in the representation of the accumulators (Montgomery-form residues): fp64 rows hold plain residues, so the factor is
// the number PR = P * R mod q as a double; integer rows hold Montgomery-form words and take REDC(d * PR).
struct RelinFold {
    const i64 *x;       // [nct][4][ell][N] = x0, x1, y0, y1 per ciphertext pair, as lf_rescale_ntt(RELAXED | PLAIN) leaves them
    i64 ct_stride;      // words between the stacks of consecutive pairs
    const i64 *PR;      // [ell]  P * R mod q_r
    int ell;            // ordinary rows: the first `ell` of the `rows` limbs
    // own[r] = the digit limb r belongs to (nullptr: none skipped): that digit's extension to limb r IS the third tensor
    // component x1 * y1 in the NTT domain — the extension (mod q_r) of a digit's mixed-radix form to one of its own primes
    // is the residue it was built from — so it is formed here from the operands instead of being read from `ext`
    const unsigned char *own;
    int xpl;            // 1: fp64-class rows of x are planes (lf_rescale_ntt with LF_NTT_PLANES; key_format | LF_STACK_PLANES)
};

// the two words at coefficients j0, j0 + 1 of an fp64-class row: raw 16 bytes, or 8 + 4 bytes of its planes
static __device__ __forceinline__ void ld_pair_dp(const i64 *row, i64 j0, i64 N, int planes, double &a, double &b) {
    if (planes) {
        const lf_u2_t l = *reinterpret_cast<const lf_u2_t *>(reinterpret_cast<const unsigned *>(row) + j0);
        const unsigned h = *reinterpret_cast<const unsigned *>(reinterpret_cast<const unsigned short *>(row + (N >> 1)) + j0);
        a = dp_from_planes(l.x, h & 0xffffu), b = dp_from_planes(l.y, h >> 16);
    } else {
        const longlong2 v = *reinterpret_cast<const longlong2 *>(row + j0);
        a = dp_from_word(v.x), b = dp_from_word(v.y);
    }
}

template <int NCT, bool FOLD, bool PLANES, bool DPL>   // DPL: fp64-class rows of `ext` in planes format (digit_planes())
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NCT == 4 ? ((DPL && !FOLD) ? 5 : 4) : 1))) ks_inner2_kernel(const i64 *__restrict__ ext, const i64 *__restrict__ ksk,
                                                        i64 part_stride, i64 comp_stride, i64 row_off, i64 *__restrict__ s,
                                                        int nparts, int rows, i64 N, RelinFold fold, int spl,
                                                        const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                        const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    // spl: the sums of fp64-class rows leave as planes (the inverse passes behind read them so: ks_tail)
    // each thread owns KI_V 16-byte column pairs 4 KiB apart: every block streams KI_V x 4 KiB contiguous runs
    // from 3 x nparts arrays, enough bytes in flight to keep HBM busy
    constexpr int KI_V = KI_COLS;
    const int r = blockIdx.y;
    const i64 j0 = (i64)blockIdx.x * (512 * KI_V) + threadIdx.x * 2;
    if (j0 >= N) return;
    const RowMod m = load_mod(ql, qh, kl, kh, r);
    const RowDp d = make_dp(m);
    const i64 *e = ext + (i64)r * N + j0;
    const i64 *k = ksk + (row_off + r) * N + j0;
    const i64 ct_ext = (i64)nparts * rows * N;   // words between the ciphertexts' extended digits
    const i64 ct_s = 2 * (i64)rows * N;          // .. and between their output pairs
    if (m.q < SMALL_PRIME_LIMIT) {
        double acc[NCT][2][2];
#pragma unroll
        for (int t = 0; t < NCT; ++t) acc[t][0][0] = acc[t][0][1] = acc[t][1][0] = acc[t][1][1] = 0.0;
        const int p_own = (FOLD && fold.own != nullptr && r < fold.ell) ? (int)fold.own[r] : -1;
        const unsigned bo_lo = (unsigned)j0 * 4u, bo_hi = (unsigned)j0 * 2u;   // DPL: byte offsets of the thread's pair in the planes
        // the own digit's words: x1 * y1, plain canonical.  SCALAR arrays, selected by value below: a choice between a 16-byte
        // struct in registers and one in global memory is compiled to a load through select(private address, global address),
        // and an array whose address is taken that way lives in scratch memory (48 .. 128 bytes per lane in the batched kernels
        // until round 6; profiles/r06_kernel_resources.txt)
        i64 xo_x[NCT], xo_y[NCT];
        if (p_own >= 0) {
#pragma unroll
            for (int t = 0; t < NCT; ++t) {
                const i64 *xs = fold.x + t * fold.ct_stride + (i64)r * N + (i64)fold.ell * N;
                double x1a, x1b, y1a, y1b;
                ld_pair_dp(xs, j0, N, fold.xpl, x1a, x1b);
                ld_pair_dp(xs + 2 * (i64)fold.ell * N, j0, N, fold.xpl, y1a, y1b);
                xo_x[t] = dp_to_word(dp_mulmod(x1a, y1a, d));
                xo_y[t] = dp_to_word(dp_mulmod(x1b, y1b, d));
                if constexpr (DPL) {   // in the register form of a pair read from the planes: one conversion for every digit
                    const u64 a = (u64)xo_x[t], b = (u64)xo_y[t];
                    xo_x[t] = (i64)((a & 0xffffffffull) | (b << 32));
                    xo_y[t] = (i64)((a >> 32) | ((b >> 32) << 16));
                }
            }
        }
#pragma unroll KI_UNROLL
        for (int p = 0; p < nparts; ++p) {
            longlong2 x[NCT];   // DPL: .x = the two low words, low half of .y = the two high halves (8 + 4 bytes, fwd_tile16<.., PLN>)
            if constexpr (DPL) {   // SGPR row base + one per-thread byte offset per plane
#pragma unroll
                for (int t = 0; t < NCT; ++t) {
                    const char *er = reinterpret_cast<const char *>(uniform_ptr(ext + (((i64)t * nparts + p) * rows + r) * N));
                    if (p == p_own) {
                        x[t].x = xo_x[t], x[t].y = xo_y[t];
                    } else {
                        x[t].x = *reinterpret_cast<const i64 *>(er + bo_lo);
                        x[t].y = (i64)*reinterpret_cast<const unsigned *>(er + 4 * N + bo_hi);
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < NCT; ++t) {
                    if (p == p_own) {
                        x[t].x = xo_x[t], x[t].y = xo_y[t];
                    } else {
                        const longlong2 v = *reinterpret_cast<const longlong2 *>(e + t * ct_ext + (i64)p * rows * N);
                        x[t].x = v.x, x[t].y = v.y;
                    }
                }
            }
            double k0x, k0y, k1x, k1y;
            if (PLANES) {   // 16 + 8 bytes for both components (see lf_key_planes)
                const i64 *kr = k - j0 + (i64)p * part_stride;
                const lf_u4_t l = __builtin_nontemporal_load(reinterpret_cast<const lf_u4_t *>(reinterpret_cast<const unsigned *>(kr) + 2 * j0));
                const lf_u2_t h = __builtin_nontemporal_load(reinterpret_cast<const lf_u2_t *>(reinterpret_cast<const unsigned *>(kr + comp_stride) + j0));
                k0x = dp_from_planes(l.x, h.x & 0xffffu), k0y = dp_from_planes(l.y, h.x >> 16);
                k1x = dp_from_planes(l.z, h.y & 0xffffu), k1y = dp_from_planes(l.w, h.y >> 16);
            } else {
                const longlong2 k0 = ld_nt(k + (i64)p * part_stride);
                const longlong2 k1 = ld_nt(k + (i64)p * part_stride + comp_stride);
                k0x = dp_from_word(k0.x), k0y = dp_from_word(k0.y), k1x = dp_from_word(k1.x), k1y = dp_from_word(k1.y);
            }
#pragma unroll
            for (int t = 0; t < NCT; ++t) {
                double x0, x1;
                if constexpr (DPL) {
                    const unsigned h = (unsigned)x[t].y;
                    x0 = dp_from_planes((unsigned)x[t].x, h & 0xffffu);
                    x1 = dp_from_planes((unsigned)((u64)x[t].x >> 32), h >> 16);
                } else {
                    x0 = dp_from_word(x[t].x), x1 = dp_from_word(x[t].y);
                }
                acc[t][0][0] += dp_mulmod_bal(x0, k0x, d);
                acc[t][0][1] += dp_mulmod_bal(x1, k0y, d);
                acc[t][1][0] += dp_mulmod_bal(x0, k1x, d);
                acc[t][1][1] += dp_mulmod_bal(x1, k1y, d);
            }
        }
        if (FOLD && r < fold.ell) {
            const double pr = dp_from_word(fold.PR[r]);
            const i64 pstride = (i64)fold.ell * N;
#pragma unroll
            for (int t = 0; t < NCT; ++t) {
                const i64 *xs = fold.x + t * fold.ct_stride + (i64)r * N;
                double x0[2], x1[2], y0[2], y1[2];
                ld_pair_dp(xs, j0, N, fold.xpl, x0[0], x0[1]);
                ld_pair_dp(xs + pstride, j0, N, fold.xpl, x1[0], x1[1]);
                ld_pair_dp(xs + 2 * pstride, j0, N, fold.xpl, y0[0], y0[1]);
                ld_pair_dp(xs + 3 * pstride, j0, N, fold.xpl, y1[0], y1[1]);
#pragma unroll
                for (int e = 0; e < 2; ++e) {   // balanced terms: |d0| <= q / 2, |d1| <= q
                    const double d0 = dp_mulmod_bal(x0[e], y0[e], d);
                    const double d1 = dp_mulmod_bal(x0[e], y1[e], d) + dp_mulmod_bal(x1[e], y0[e], d);
                    acc[t][0][e] += dp_mulmod_bal(d0, pr, d);
                    acc[t][1][e] += dp_mulmod_bal(d1, pr, d);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NCT; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                longlong2 o;
                o.x = dp_to_word(dp_reduce(acc[t][c][0], d.q, d.qinv));
                o.y = dp_to_word(dp_reduce(acc[t][c][1], d.q, d.qinv));
                i64 *srow = s + t * ct_s + ((i64)c * rows + r) * N;
                if (spl) {
                    const lf_u2_t l = {(unsigned)o.x, (unsigned)o.y};
                    *reinterpret_cast<lf_u2_t *>(reinterpret_cast<unsigned *>(srow) + j0) = l;
                    *reinterpret_cast<unsigned *>(reinterpret_cast<unsigned short *>(srow + (N >> 1)) + j0) =
                        (unsigned)((u64)o.x >> 32) | ((unsigned)((u64)o.y >> 32) << 16);
                } else {
                    *reinterpret_cast<longlong2 *>(srow + j0) = o;
                }
            }
    } else {
        i64 acc[NCT][2][2];
#pragma unroll
        for (int t = 0; t < NCT; ++t) acc[t][0][0] = acc[t][0][1] = acc[t][1][0] = acc[t][1][1] = 0;
        const int p_own = (FOLD && fold.own != nullptr && r < fold.ell) ? (int)fold.own[r] : -1;
        i64 xo_x[NCT], xo_y[NCT];   // the own digit's words: REDC62(x1 * y1), Montgomery form below 2q (scalars: see above)
        if (p_own >= 0) {
#pragma unroll
            for (int t = 0; t < NCT; ++t) {
                const i64 *xs = fold.x + t * fold.ct_stride + (i64)r * N + j0 + (i64)fold.ell * N;
                const longlong2 X1 = *reinterpret_cast<const longlong2 *>(xs), Y1 = *reinterpret_cast<const longlong2 *>(xs + 2 * (i64)fold.ell * N);
                xo_x[t] = mm62u((u64)X1.x, (u64)Y1.x, m.q, m.k);
                xo_y[t] = mm62u((u64)X1.y, (u64)Y1.y, m.q, m.k);
            }
        }
        for (int p = 0; p < nparts; ++p) {
            const longlong2 k0 = ld_nt(k + (i64)p * part_stride);
            const longlong2 k1 = ld_nt(k + (i64)p * part_stride + comp_stride);
#pragma unroll
            for (int t = 0; t < NCT; ++t) {
                longlong2 x;
                if (p == p_own) {
                    x.x = xo_x[t], x.y = xo_y[t];
                } else {
                    const longlong2 v = *reinterpret_cast<const longlong2 *>(e + t * ct_ext + (i64)p * rows * N);
                    x.x = v.x, x.y = v.y;
                }
                acc[t][0][0] = csub(acc[t][0][0] + mm62u((u64)x.x, (u64)k0.x, m.q, m.k), m.q2);
                acc[t][0][1] = csub(acc[t][0][1] + mm62u((u64)x.y, (u64)k0.y, m.q, m.k), m.q2);
                acc[t][1][0] = csub(acc[t][1][0] + mm62u((u64)x.x, (u64)k1.x, m.q, m.k), m.q2);
                acc[t][1][1] = csub(acc[t][1][1] + mm62u((u64)x.y, (u64)k1.y, m.q, m.k), m.q2);
            }
        }
        if (FOLD && r < fold.ell) {
            const u64 pr = (u64)fold.PR[r];
            const i64 pstride = (i64)fold.ell * N;
#pragma unroll
            for (int t = 0; t < NCT; ++t) {
                const i64 *xs = fold.x + t * fold.ct_stride + (i64)r * N + j0;
                const longlong2 X0 = *reinterpret_cast<const longlong2 *>(xs), X1 = *reinterpret_cast<const longlong2 *>(xs + pstride);
                const longlong2 Y0 = *reinterpret_cast<const longlong2 *>(xs + 2 * pstride), Y1 = *reinterpret_cast<const longlong2 *>(xs + 3 * pstride);
                const u64 x0[2] = {(u64)X0.x, (u64)X0.y}, x1[2] = {(u64)X1.x, (u64)X1.y};
                const u64 y0[2] = {(u64)Y0.x, (u64)Y0.y}, y1[2] = {(u64)Y1.x, (u64)Y1.y};
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const i64 d0 = mm62u(x0[e], y0[e], m.q, m.k);
                    const i64 d1 = csub(mm62u(x0[e], y1[e], m.q, m.k) + mm62u(x1[e], y0[e], m.q, m.k), m.q2);
                    acc[t][0][e] = csub(acc[t][0][e] + mm62u((u64)d0, pr, m.q, m.k), m.q2);
                    acc[t][1][e] = csub(acc[t][1][e] + mm62u((u64)d1, pr, m.q, m.k), m.q2);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NCT; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                longlong2 o;
                o.x = acc[t][c][0];
                o.y = acc[t][c][1];
                *reinterpret_cast<longlong2 *>(s + t * ct_s + ((i64)c * rows + r) * N + j0) = o;
            }
    }
}

// largest number of leading stages (logN - 12) whose extension + strided pass runs as the column kernel (lf_tune).
// With the digit loop as a runtime loop (R loads in flight, 100 VGPRs at R = 16) the column form also wins at logN 16:
// gold cc_mult 2 104-2 130 -> 2 168-2 183 ops/s, rotate 2 653-2 695 -> 2 733-2 763, 64 rotations under one key
// 3 110 -> 3 300 /s (tools/eo.py --ext-cols-max 3 | 4, one box); round 2's fully unrolled form had lost there (116 vs 95 us).
// And at logN 17 (platinum, 32 words per thread: 173 VGPRs, 2 waves per SIMD — the kernel's waves_per_eu follows K), together with
// the column form of the sums' last inverse pass: cc_mult 2 163 -> 2 054 us, rotate 1 845 -> 1 740 us, same words.
int g_ks_ext_cols_max = 5;

// The extended digits between ks_forward and ks_tail (tmp: scratch of the key switch, opaque to the caller) keep the
// fp64-class rows in planes format — 6 bytes per word on each of their three trips — at every two-pass ring degree (the column
// kernel and the LDS-tiled extension both write it) where both classes are present.  BOTH halves decide with this function: the
// knob must not change between an lf_ks_fwd and its lf_ks_tail (lf_tune is a start-up / A-B facility, see the header).
int g_digit_planes = 1;
int g_more_planes = 3;   // LF_TUNE_MORE_PLANES: bit 0 = the sums of a key switch, bit 1 = cc_mult's operand stack (with g_digit_planes)
bool digit_planes(int logN, const RowList &dp, const RowList &in) {
    return g_digit_planes && logN > NTT_TILE_LOG_MAX && dp.n && in.n;
}

}  // namespace
extern "C" int lf_stack_planes(int logN, int rows, const int64_t *q_host) {
    if (!g_digit_planes || !(g_more_planes & 2) || logN <= NTT_TILE_LOG_MAX || logN > 2 * NTT_TILE_LOG_MAX || !q_host || rows < 1) return 0;
    int small = 0, large = 0;
    for (int r = 0; r < rows; ++r) ((uint64_t)q_host[r] < SMALL_PRIME_LIMIT ? small : large)++;
    return small && large ? 1 : 0;
}
namespace {
void classify_rows(int rows, const int64_t *q_host, RowList &dp, RowList &in) {
    dp.n = in.n = 0;
    for (int r = 0; r < rows; ++r) {
        RowList &dst = (q_host && (uint64_t)q_host[r] < SMALL_PRIME_LIMIT) ? dp : in;
        dst.id[dst.n++] = (unsigned short)r;
    }
}

// K2 + P2 of `nparts` digits (descriptors desc[0 .. nparts), extended digits into tmp[nct][nparts][rows][N])
int ks_forward(const int64_t *state, int64_t state_stride, int nct, int nparts, int rows, int logN, const int64_t *desc,
               const int64_t *E, const double *Ed, int64_t *tmp, const int64_t *psi_br, const double *psi_dp,
               const int64_t *q_host, const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh,
               hipStream_t st, const unsigned char *own = nullptr, int p0 = 0) {
    if (!psi_dp) return LF_ERR_ARG;   // the relaxed arithmetic of both classes lives in the auxiliary table
    const int tl = NTT_TILE_LOG_MAX, S1 = logN - tl;
    RowList dp, in;
    classify_rows(rows, q_host, dp, in);
    const bool dplanes = digit_planes(logN, dp, in);
    lf_fmt_note(tmp, ((size_t)nct * nparts * rows << logN) * 8, dplanes ? LF_FMT_PLANES : LF_FMT_RAW);
    const KsGeom kg{logN, tl, S1, rows, nparts, (i64)1 << logN, nct, (i64)state_stride, own, p0, dplanes ? 1 : 0};
    const unsigned tiles = 1u << (logN - tl);
    const unsigned polys = (unsigned)nparts * (unsigned)nct;   // extended digits of all ciphertexts: one stack
    const bool mixed = dp.n && in.n;   // both arithmetic classes in one launch per step
    // K2: extend + strided pass — as one register step per column when the strided pass has at most 4 stages
    // (measured on MI355X, extension kernel alone: silver / logN 15 22.5 -> 20.0 us; gold / logN 16 see g_ks_ext_cols_max)
    if (S1 <= g_ks_ext_cols_max) {
        const unsigned per_limb = ((1u << tl) / NTT_COL_THREADS) * polys;   // column chunks x digits x ciphertexts
        const ClassLists cl = class_lists(in, dp, per_limb * (unsigned)in.n);   // either list may be empty
        const dim3 grid((unsigned)cl.in_blocks + per_limb * (unsigned)dp.n), block(NTT_COL_THREADS);
#define LF_EXT_COLS_CASE(KK)                                                                                            \
    case KK:                                                                                                            \
        hipLaunchKernelGGL((ks_ext_cols_mixed<KK>), grid, block, 0, st, (const i64 *)state, (i64 *)tmp, kg, cl,          \
                           (const i64 *)desc, (const i64 *)E, Ed, (const i64 *)psi_br, psi_dp, (const i64 *)ql,         \
                           (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);                                         \
        break;
        switch (S1) { LF_EXT_COLS_CASE(1) LF_EXT_COLS_CASE(2) LF_EXT_COLS_CASE(3) LF_EXT_COLS_CASE(4) LF_EXT_COLS_CASE(5) }
#undef LF_EXT_COLS_CASE
    } else if (mixed) {
        const ClassLists cl = class_lists(in, dp, tiles * in.n * polys);
        hipLaunchKernelGGL(ks_ext_pass1_mixed, dim3((unsigned)cl.in_blocks + tiles * dp.n * polys), dim3(NTT_THREADS), 0, st,
                           (const i64 *)state, (i64 *)tmp, kg, cl, (const i64 *)desc, (const i64 *)E, Ed, (const i64 *)psi_br,
                           psi_dp, (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
    } else if (dp.n)
        hipLaunchKernelGGL(ks_ext_pass1<true>, dim3(tiles * dp.n * polys), dim3(NTT_THREADS), 0, st, (const i64 *)state,
                           (i64 *)tmp, kg, dp, (const i64 *)desc, (const i64 *)E, Ed, (const i64 *)psi_br, psi_dp,
                           (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
    if (in.n && !mixed)
        hipLaunchKernelGGL(ks_ext_pass1<false>, dim3(tiles * in.n * polys), dim3(NTT_THREADS), 0, st, (const i64 *)state,
                           (i64 *)tmp, kg, in, (const i64 *)desc, (const i64 *)E, Ed, (const i64 *)psi_br, psi_dp,
                           (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
    // contiguous forward pass, in place on tmp (relaxed)
    {
        const PassGeom g{logN, tl, 0, tl, S1, 0, rows, (int)polys, 1, 1, 0, own, nparts, p0};
        launch_pass16(false, 1, (int)polys, st, (const i64 *)tmp, (i64 *)tmp, g, in, dp, (const i64 *)psi_br, psi_dp,
                      (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh, nullptr, dplanes);
    }
    return (int)hipGetLastError();
}

// K3 + K4: inner product of the nparts extended digits with the key, inverse transform to canonical coefficients
int ks_tail(int nct, int nparts, int rows, int logN, const int64_t *ksk, int64_t part_stride, int64_t comp_stride,
            int64_t row_off, int64_t *tmp, int64_t *s, const int64_t *ipsi_br, const double *ipsi_dp,
            const int64_t *Ninv, const int64_t *q_host, const int64_t *ql, const int64_t *qh, const int64_t *kl,
            const int64_t *kh, hipStream_t st, const RelinFold *fold = nullptr, int key_format = LF_KEY_RAW) {
    if (!ipsi_dp || (key_format != LF_KEY_RAW && key_format != LF_KEY_PLANES)) return LF_ERR_ARG;
    if (key_format == LF_KEY_PLANES && ((((uintptr_t)ksk | (uintptr_t)(part_stride * 8) | (uintptr_t)(comp_stride * 8)) & 15)))
        return LF_ERR_ARG;
    const int tl = NTT_TILE_LOG_MAX, S1 = logN - tl;
    RowList dp, in;
    classify_rows(rows, q_host, dp, in);
    const bool mixed = dp.n && in.n;
    const bool dplanes = digit_planes(logN, dp, in);
    const bool cols_last = S1 <= 4 || (S1 == 5 && mixed && g_ks_ext_cols_max > 4);   // column form of the last inverse pass
    // the digits in tmp must be in the format this half is about to read (lf_tune flipped between lf_ks_fwd and here: LF_ERR_STATE)
    if (int e = lf_fmt_expect(tmp, ((size_t)nct * nparts * rows << logN) * 8, dplanes ? LF_FMT_PLANES : LF_FMT_RAW)) return e;
    // The SUMS travel the same way: the inner product writes fp64-class rows as planes into s, the tiled inverse pass carries
    // them s -> tmp (the digits are spent by then: tmp is scratch, and a pass that changes the format cannot run in place),
    // the column pass reads the planes and leaves canonical words in s.  6 instead of 8 bytes per word on three of the
    // sums' four trips; needs room for 2 nct polynomials in tmp (two digits or more) and the column form of the last pass.
    const bool spl = dplanes && (g_more_planes & 1) && cols_last && nparts >= 2;
    // K3: inner product with the key, summed over the digits
    {
        const i64 N = (i64)1 << logN;
        dim3 grid((unsigned)((N + 512 * KI_COLS - 1) / (512 * KI_COLS)), (unsigned)rows);
        const RelinFold nofold{nullptr, 0, nullptr, 0, nullptr, 0};
#define LF_INNER_LAUNCH(NCT, FOLDB, PL, DPLB, FOLDV)                                                                   \
    hipLaunchKernelGGL((ks_inner2_kernel<NCT, FOLDB, PL, DPLB>), grid, dim3(256), 0, st, (const i64 *)tmp, (const i64 *)ksk, \
                       (i64)part_stride, (i64)comp_stride, (i64)row_off, (i64 *)s, nparts, rows, N, FOLDV, spl ? 1 : 0, (const i64 *)ql, \
                       (const i64 *)qh, (const i64 *)kl, (const i64 *)kh)
#define LF_INNER_DPL(NCT, FOLDB, PL, FOLDV)                                                                            \
    do {                                                                                                               \
        if (dplanes) LF_INNER_LAUNCH(NCT, FOLDB, PL, true, FOLDV);                                                     \
        else LF_INNER_LAUNCH(NCT, FOLDB, PL, false, FOLDV);                                                            \
    } while (0)
#define LF_INNER_CASE(NCT)                                                                                             \
    case NCT:                                                                                                          \
        if (fold && planes) LF_INNER_DPL(NCT, true, true, *fold);                                                      \
        else if (fold) LF_INNER_DPL(NCT, true, false, *fold);                                                          \
        else if (planes) LF_INNER_DPL(NCT, false, true, nofold);                                                       \
        else LF_INNER_DPL(NCT, false, false, nofold);                                                                  \
        break;
        const bool planes = key_format == LF_KEY_PLANES;
        switch (nct) {
            LF_INNER_CASE(1) LF_INNER_CASE(2) LF_INNER_CASE(4)
        }
#undef LF_INNER_LAUNCH
#undef LF_INNER_DPL
#undef LF_INNER_CASE
    }
    // K4: inverse transform -> canonical coefficients (relaxed, tail 2), in place on s
    const int inv_polys = 2 * nct;
    const unsigned per_row2 = (unsigned)inv_polys << (logN - tl);
    for (int pass = 0; pass < 2; ++pass) {
        PassGeom g = pass == 0 ? PassGeom{logN, tl, 0, tl, 0, 0, rows, inv_polys, 1, 0, 0}
                               : PassGeom{logN, tl, 1, S1, tl, tl - S1, rows, inv_polys, 1, 1, 0};
        if (pass == 0) {
            launch_pass16(true, 1, inv_polys, st, (const i64 *)s, spl ? (i64 *)tmp : (i64 *)s, g, in, dp, (const i64 *)ipsi_br, ipsi_dp,
                          (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh, nullptr, spl);
            continue;
        }
        if (pass == 1 && cols_last) {   // (logN 17: the column form of both ends goes with the knob)
            if (spl) {
                g.pln = PLN_IN;
                g.pln_src = (const i64 *)tmp;
            }
            if (mixed) {
                launch_inv_cols_mixed(S1, inv_polys, st, (i64 *)s, g, in, dp, (const i64 *)ipsi_br, ipsi_dp, (const i64 *)Ninv, 2,
                                      (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
                continue;
            }
            if (dp.n)
                launch_inv_cols<true>(S1, inv_polys, st, (i64 *)s, g, dp, (const i64 *)ipsi_br, ipsi_dp, (const i64 *)Ninv, 2,
                                      (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
            if (in.n)
                launch_inv_cols<false>(S1, inv_polys, st, (i64 *)s, g, in, (const i64 *)ipsi_br, ipsi_dp, (const i64 *)Ninv, 2,
                                       (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
            continue;
        }
        if (mixed) {
            const ClassLists cl = class_lists(in, dp, per_row2 * (unsigned)in.n);
            hipLaunchKernelGGL((ntt_inv_pass_mixed<true>), dim3((unsigned)cl.in_blocks + per_row2 * dp.n), dim3(NTT_THREADS), 0, st,
                               (const i64 *)s, (i64 *)s, g, cl, (const i64 *)ipsi_br, ipsi_dp, (const i64 *)Ninv,
                               pass == 1 ? 2 : TAIL_NONE, (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
            continue;
        }
        if (dp.n)
            hipLaunchKernelGGL((ntt_inv_pass_io<true, true>), dim3(per_row2 * dp.n), dim3(NTT_THREADS), 0, st, (const i64 *)s, (i64 *)s,
                               g, dp, (const i64 *)ipsi_br, ipsi_dp, (const i64 *)Ninv, pass == 1 ? 2 : TAIL_NONE,
                               (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
        if (in.n)
            hipLaunchKernelGGL((ntt_inv_pass_io<false, true>), dim3(per_row2 * in.n), dim3(NTT_THREADS), 0, st, (const i64 *)s, (i64 *)s,
                               g, in, (const i64 *)ipsi_br, ipsi_dp, (const i64 *)Ninv, pass == 1 ? 2 : TAIL_NONE,
                               (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
    }
    return (int)hipGetLastError();
}

}  // namespace

// cc_mult's product -> digits in one launch behind the tiled pass where it qualifies (ckks_ops.hip: product_digits): 0 = never
int lf_g_intt_digits = 1;
extern int lf_g_ws_extra_stage;   // ckks_ntt.hip

extern "C" {

int lf_tune(int which, int value) {
    int *knob = which == LF_TUNE_KS_EXT_COLS_MAX ? &g_ks_ext_cols_max : which == LF_TUNE_INTT_DIGITS ? &lf_g_intt_digits
                : which == LF_TUNE_DIGIT_PLANES ? &g_digit_planes : which == LF_TUNE_WS_EXTRA_STAGE ? &lf_g_ws_extra_stage
                : which == LF_TUNE_MORE_PLANES ? &g_more_planes : nullptr;
    if (!knob) return -1;
    const int old = *knob;
    if (value < 0) return old;
    if (which == LF_TUNE_KS_EXT_COLS_MAX && value > 5) return old;
    if (which == LF_TUNE_MORE_PLANES && value > 3) return old;
    if ((which == LF_TUNE_INTT_DIGITS || which == LF_TUNE_DIGIT_PLANES || which == LF_TUNE_WS_EXTRA_STAGE) && value > 1) return old;
    if (*knob != value && (which == LF_TUNE_DIGIT_PLANES || which == LF_TUNE_WS_EXTRA_STAGE || which == LF_TUNE_MORE_PLANES))
        lf_fmt_epoch_bump();   // scratch written so far was written under another format setting
    *knob = value;
    return old;
}

int lf_key_planes(const int64_t *src_b, const int64_t *src_a, int64_t *dst_b, int64_t *dst_a, int rows, int64_t N,
                  const int64_t *ql, const int64_t *qh, int device, void *stream) {
    if (rows < 0 || rows > 65535 || N < 2 || (N & 1) || !src_b || !src_a || !dst_b || !dst_a || dst_b == dst_a || src_b == dst_b ||
        src_a == dst_a || src_a == dst_b || src_b == dst_a || !ql || !qh ||
        (((uintptr_t)src_b | (uintptr_t)src_a | (uintptr_t)dst_b | (uintptr_t)dst_a) & 15))
        return LF_ERR_ARG;
    if (rows == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    hipLaunchKernelGGL(key_planes_kernel, dim3((unsigned)((N / 2 + 255) / 256), (unsigned)rows), dim3(256), 0, (hipStream_t)stream,
                       (const i64 *)src_b, (const i64 *)src_a, (i64 *)dst_b, (i64 *)dst_a, (i64)N, (const i64 *)ql, (const i64 *)qh);
    return (int)hipGetLastError();
}

int lf_ks_core_batch(const int64_t *state, int64_t state_stride, int nct, int nparts, int rows, int logN, const int64_t *desc,
                     const int64_t *E, const double *Ed, const int64_t *ksk, int64_t part_stride, int64_t comp_stride,
                     int64_t row_off, int key_format, int64_t *tmp, int64_t *s, const int64_t *psi_br, const double *psi_dp,
                     const int64_t *ipsi_br, const double *ipsi_dp, const int64_t *Ninv, const int64_t *q_host, const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh, int device,
                     void *stream) {
    if (nparts < 1 || rows < 1 || rows > MAX_LIST_ROWS || logN <= NTT_TILE_LOG_MAX || logN > 2 * NTT_TILE_LOG_MAX ||
        !q_host || !psi_dp || !ipsi_dp || !Ed || (nct != 1 && nct != 2 && nct != 4))
        return LF_ERR_ARG;
    if (int e = lf_set_device(device)) return e;
    hipStream_t st = (hipStream_t)stream;
    if (int e = ks_forward(state, state_stride, nct, nparts, rows, logN, desc, E, Ed, tmp, psi_br, psi_dp, q_host, ql, qh, kl, kh, st))
        return e;
    return ks_tail(nct, nparts, rows, logN, ksk, part_stride, comp_stride, row_off, tmp, s, ipsi_br, ipsi_dp, Ninv, q_host, ql, qh,
                   kl, kh, st, nullptr, key_format);
}

/* The two halves of lf_ks_core as separate calls, so that a limb-sharded engine can start on the digits that have
 * arrived while the others are still travelling ("gather part p+1 while extending part p"):
 *   lf_ks_fwd   extension + forward NTT of `nparts` digits, descriptors desc[0 .. nparts) (the caller offsets desc and
 *               tmp to the first digit of the group: tmp_group = tmp + first * rows * N);
 *   lf_ks_tail  once every group is done: inner product of ALL nparts digits with the key + inverse NTT. */
int lf_ks_fwd(const int64_t *state, int nparts, int rows, int logN, const int64_t *desc, const int64_t *E, const double *Ed,
              int64_t *tmp, const int64_t *psi_br, const double *psi_dp, const int64_t *q_host, const int64_t *ql,
              const int64_t *qh, const int64_t *kl, const int64_t *kh, int device, void *stream) {
    if (nparts < 0 || rows < 1 || rows > MAX_LIST_ROWS || logN <= NTT_TILE_LOG_MAX || logN > 2 * NTT_TILE_LOG_MAX ||
        !q_host || !psi_dp || !Ed)
        return LF_ERR_ARG;
    if (nparts == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    return ks_forward(state, 0, 1, nparts, rows, logN, desc, E, Ed, tmp, psi_br, psi_dp, q_host, ql, qh, kl, kh, (hipStream_t)stream);
}

int lf_ks_tail(int nparts, int rows, int logN, const int64_t *ksk, int64_t part_stride, int64_t comp_stride, int64_t row_off,
               int key_format, int64_t *tmp, int64_t *s, const int64_t *ipsi_br, const double *ipsi_dp, const int64_t *Ninv,
               const int64_t *q_host, const int64_t *ql, const int64_t *qh, const int64_t *kl,
               const int64_t *kh, int device, void *stream) {
    if (nparts < 1 || rows < 1 || rows > MAX_LIST_ROWS || logN <= NTT_TILE_LOG_MAX || logN > 2 * NTT_TILE_LOG_MAX ||
        !q_host || !ipsi_dp)
        return LF_ERR_ARG;
    if (int e = lf_set_device(device)) return e;
    return ks_tail(1, nparts, rows, logN, ksk, part_stride, comp_stride, row_off, tmp, s, ipsi_br, ipsi_dp, Ninv, q_host, ql, qh, kl,
                   kh, (hipStream_t)stream, nullptr, key_format);
}

/* Relinearisation inside cc_mult (see RelinFold): lf_ks_core_batch / lf_ks_fwd / lf_ks_tail whose sums additionally
 * receive P * (x0 y0) and P * (x0 y1 + x1 y0) on the `ell` ordinary rows, from the stack x = [nct][4][ell][N]; with `own`
 * the (digit, own limb) pairs are neither extended nor transformed: the inner product forms x1 y1 for them. */
int lf_relin_core_batch(const int64_t *state, int64_t state_stride, int nct, int nparts, int rows, int logN, const int64_t *desc,
                        const int64_t *E, const double *Ed, const int64_t *ksk, int64_t part_stride, int64_t comp_stride,
                        int64_t row_off, int key_format, int64_t *tmp, int64_t *s, const int64_t *psi_br, const double *psi_dp,
                        const int64_t *ipsi_br, const double *ipsi_dp, const int64_t *Ninv, const int64_t *x, int64_t x_ct_stride,
                        const int64_t *PR, int ell, const uint8_t *own, const int64_t *q_host,
                        const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh, int device, void *stream) {
    if (nparts < 1 || nparts > 254 || rows < 1 || rows > MAX_LIST_ROWS || logN <= NTT_TILE_LOG_MAX || logN > 2 * NTT_TILE_LOG_MAX ||
        !q_host || !psi_dp || !ipsi_dp || !Ed || (nct != 1 && nct != 2 && nct != 4) || !x || !PR || ell < 0 || ell > rows)
        return LF_ERR_ARG;
    if (int e = lf_set_device(device)) return e;
    hipStream_t st = (hipStream_t)stream;
    if (int e = ks_forward(state, state_stride, nct, nparts, rows, logN, desc, E, Ed, tmp, psi_br, psi_dp, q_host, ql, qh, kl, kh, st,
                           (const unsigned char *)own, 0))
        return e;
    const int xpl = (key_format & LF_STACK_PLANES) ? 1 : 0;
    for (int t = 0; t < nct; ++t)   // the operand stacks must be in the format the caller says (see lf_fmt_expect)
        if (int e = lf_fmt_expect(x + t * x_ct_stride, ((size_t)4 * ell << logN) * 8, xpl ? LF_FMT_PLANES : LF_FMT_RAW)) return e;
    const RelinFold fold{(const i64 *)x, (i64)x_ct_stride, (const i64 *)PR, ell, (const unsigned char *)own, xpl};
    return ks_tail(nct, nparts, rows, logN, ksk, part_stride, comp_stride, row_off, tmp, s, ipsi_br, ipsi_dp, Ninv, q_host, ql, qh,
                   kl, kh, st, &fold, key_format & ~LF_STACK_PLANES);
}

int lf_relin_fwd(const int64_t *state, int first, int nparts, int rows, int logN, const int64_t *desc, const int64_t *E,
                 const double *Ed, int64_t *tmp, const int64_t *psi_br, const double *psi_dp, const uint8_t *own,
                 const int64_t *q_host, const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh, int device,
                 void *stream) {
    if (first < 0 || nparts < 0 || first + nparts > 254 || rows < 1 || rows > MAX_LIST_ROWS || logN <= NTT_TILE_LOG_MAX ||
        logN > 2 * NTT_TILE_LOG_MAX || !q_host || !psi_dp || !Ed)
        return LF_ERR_ARG;
    if (nparts == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    return ks_forward(state, 0, 1, nparts, rows, logN, desc + 3 * (int64_t)first, E, Ed, tmp + (((int64_t)first * rows) << logN), psi_br,
                      psi_dp, q_host, ql, qh, kl, kh, (hipStream_t)stream, (const unsigned char *)own, first);
}

int lf_relin_tail(int nparts, int rows, int logN, const int64_t *ksk, int64_t part_stride, int64_t comp_stride, int64_t row_off,
                  int key_format, int64_t *tmp, int64_t *s, const int64_t *ipsi_br, const double *ipsi_dp, const int64_t *Ninv,
                  const int64_t *x, const int64_t *PR, int ell, const uint8_t *own, const int64_t *q_host, const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh, int device,
                  void *stream) {
    if (nparts < 1 || nparts > 254 || rows < 1 || rows > MAX_LIST_ROWS || logN <= NTT_TILE_LOG_MAX || logN > 2 * NTT_TILE_LOG_MAX ||
        !q_host || !ipsi_dp || !x || !PR || ell < 0 || ell > rows)
        return LF_ERR_ARG;
    if (int e = lf_set_device(device)) return e;
    const int xpl = (key_format & LF_STACK_PLANES) ? 1 : 0;
    if (int e = lf_fmt_expect(x, ((size_t)4 * ell << logN) * 8, xpl ? LF_FMT_PLANES : LF_FMT_RAW)) return e;
    const RelinFold fold{(const i64 *)x, 0, (const i64 *)PR, ell, (const unsigned char *)own, xpl};
    return ks_tail(1, nparts, rows, logN, ksk, part_stride, comp_stride, row_off, tmp, s, ipsi_br, ipsi_dp, Ninv, q_host, ql, qh, kl,
                   kh, (hipStream_t)stream, &fold, key_format & ~LF_STACK_PLANES);
}

int lf_ks_core(const int64_t *state, int nparts, int rows, int logN, const int64_t *desc, const int64_t *E,
               const double *Ed, const int64_t *ksk, int64_t part_stride, int64_t comp_stride, int64_t row_off,
               int key_format, int64_t *tmp, int64_t *s, const int64_t *psi_br, const double *psi_dp,
               const int64_t *ipsi_br, const double *ipsi_dp, const int64_t *Ninv, const int64_t *q_host, const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh, int device,
               void *stream) {
    return lf_ks_core_batch(state, 0, 1, nparts, rows, logN, desc, E, Ed, ksk, part_stride, comp_stride, row_off, key_format, tmp, s, psi_br,
                            psi_dp, ipsi_br, ipsi_dp, Ninv, q_host, ql, qh, kl, kh, device, stream);
}

}  // extern "C"
