// ckks_ntt.hip — negacyclic NTT / iNTT for gfx950: lf_ntt, lf_intt, lf_twiddle_dp.
//
// DAG and semantics: the reference's radix-2 Cooley-Tukey (forward, natural -> bit-reversed) and
// Gentleman-Sande (inverse) stages with lazy values in [0, 2q) (K.cu:236-275, 433-473; index
// tables ckks_context.py:89-142).  Execution is re-designed for CDNA4:
//
//   * at most TWO passes per transform instead of logN launches: a tile of up to 4096 coefficients of
//     one limb lives in LDS; each thread keeps 8 coefficients in registers across three consecutive
//     stages (12 butterflies per LDS round trip, 7 twiddle loads); N > 4096 splits the DAG into a
//     column-strided pass (the logN-12 largest pair distances) and a contiguous pass (12 stages);
//   * LDS words are laid out with one pad word per 8 (index L + (L >> 3)), which makes the 8-word-per-
//     thread access patterns of the small-distance steps bank-conflict free for ds_read/write_b64;
//   * twiddles come from a compact [limbs][N] table (entry x = psi^brev(x)), index = a shift of the
//     tile-local coefficient index — no per-stage tables, no gather indices;
//   * blocks are ordered so that the polynomials of a batch that share a (limb, tile) — hence a
//     twiddle tile — run back to back on the same XCD and hit its L2.  (A persistent variant with
//     register prefetch of the next tile was measured and dropped: the passes are instruction-issue
//     bound on gfx950, not latency bound, and the prefetch registers cost occupancy.);
//   * TWO arithmetic classes, one kernel instantiation each (so neither pays the other's registers):
//       integer : exact REDC62 on 64x64->128 products (ckks_common.h) — any prime < 2^60;
//       fp64    : primes < 2^41.  gfx950 issues 32-bit integer multiplies at about half rate, so a
//                 REDC62 costs ~97 issue slots; (a*w) mod q through 2 FMAs + 1 rint costs ~31.
//                 The reference's lazy arithmetic is reproduced EXACTLY: every lazy word of the
//                 reference is a residue mod 2q in [0, 2q), additions/subtractions are exact mod 2q,
//                 and REDC62(S*O) = T0 + q*[T0*2^62 < S*O] with T0 = (w*O) mod q, where the bracket can
//                 only be 1 when T0 < 2^22 (probability 2^-18) — then the integer comparison is
//                 evaluated.  So the fp64 path accumulates sums without conditional subtractions and
//                 reduces mod 2q once per LDS round trip; outputs are bit-identical to the integer path.
//     A tile containing a word outside [0, 2q) (the rare signed-lazy inputs) is run by
//     a compact fully-signed integer routine inside the same kernel, exactly as the reference would.
//   * LF_NTT_RELAXED transforms are for callers that only need the result modulo q (the fused key
//     switch): negative input words are folded, outputs are canonical residues.  Their integer class (logN 13..16)
//     multiplies by PLAIN twiddles with precomputed Shoup quotients from the auxiliary table (ckks_common.h).
#include <utility>
#include "../../include/ckks_hip.h"
#include "ckks_ntt_core.h"
#include "ckks_ntt_tile16.h"
#include <stdlib.h>

int lf_g_ws_extra_stage = 1;   // lf_ntt_ws: the column pass takes the tiles' first stage too (ntt_forward; lf_tune in ckks_ks.hip)

namespace {

// split the rows into the two arithmetic classes (host side; q_host may be NULL = all integer)
void classify(int rows, const int64_t *q_host, const void *dp_table, RowList &dp, RowList &in) {
    dp.n = in.n = 0;
    for (int r = 0; r < rows; ++r) {
        const bool small = q_host && dp_table && (uint64_t)q_host[r] < SMALL_PRIME_LIMIT;
        RowList &dst = small ? dp : in;
        dst.id[dst.n++] = (unsigned short)r;
    }
}

template <bool DP>
void launch_cols(int K, unsigned blocks, hipStream_t st, i64 *base, const PassGeom &g, const RowList &rl,
                 const int64_t *psi_br, const double *psi_dp, const i64 *rs, const int64_t *ql, const int64_t *qh,
                 const int64_t *kl, const int64_t *kh) {
#define LF_COLS_CASE(KK)                                                                                              \
    case KK:                                                                                                          \
        hipLaunchKernelGGL((ntt_fwd_cols<DP, KK>), dim3(blocks), dim3(NTT_COL_THREADS), 0, st, base, g, rl,           \
                           (const i64 *)psi_br, psi_dp, rs, (const i64 *)ql, (const i64 *)qh, (const i64 *)kl,        \
                           (const i64 *)kh);                                                                          \
        break;
    switch (K) {
        LF_COLS_CASE(1)
        LF_COLS_CASE(2)
        LF_COLS_CASE(3)
        LF_COLS_CASE(4)
    }
#undef LF_COLS_CASE
}

template <int K>
void launch_cols_mixed_k(unsigned blocks, hipStream_t st, i64 *base, const PassGeom &g, const ClassLists &cl,
                         const int64_t *psi_br, const double *psi_dp, const i64 *rs, const int64_t *ql, const int64_t *qh,
                         const int64_t *kl, const int64_t *kh) {
    hipLaunchKernelGGL((ntt_fwd_cols_mixed<K>), dim3(blocks), dim3(NTT_COL_THREADS), 0, st, base, g, cl, (const i64 *)psi_br,
                       psi_dp, rs, (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
}

void launch_cols_mixed(int K, unsigned per_limb, hipStream_t st, i64 *base, const PassGeom &g, const RowList &in,
                       const RowList &dp, const int64_t *psi_br, const double *psi_dp, const i64 *rs, const int64_t *ql,
                       const int64_t *qh, const int64_t *kl, const int64_t *kh) {
    const ClassLists cl = class_lists(in, dp, per_limb * (unsigned)in.n);
    const unsigned blocks = (unsigned)cl.in_blocks + per_limb * (unsigned)dp.n;
    switch (K) {
        case 1: launch_cols_mixed_k<1>(blocks, st, base, g, cl, psi_br, psi_dp, rs, ql, qh, kl, kh); break;
        case 2: launch_cols_mixed_k<2>(blocks, st, base, g, cl, psi_br, psi_dp, rs, ql, qh, kl, kh); break;
        case 3: launch_cols_mixed_k<3>(blocks, st, base, g, cl, psi_br, psi_dp, rs, ql, qh, kl, kh); break;
        case 4: launch_cols_mixed_k<4>(blocks, st, base, g, cl, psi_br, psi_dp, rs, ql, qh, kl, kh); break;
    }
}

template <int K>
void launch_cols_mixed_rs_k(unsigned blocks, hipStream_t st, i64 *base, const PassGeom &g, const ClassLists &cl,
                            const RescaleSrc &rsrc, const int64_t *psi_br, const double *psi_dp, const i64 *rs,
                            const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh) {
    hipLaunchKernelGGL((ntt_fwd_cols_mixed_rs<K>), dim3(blocks), dim3(NTT_COL_THREADS), 0, st, base, g, cl, rsrc,
                       (const i64 *)psi_br, psi_dp, rs, (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
}

void launch_cols_mixed_rs(int K, unsigned per_limb, hipStream_t st, i64 *base, const PassGeom &g, const RowList &in,
                          const RowList &dp, const RescaleSrc &rsrc, const int64_t *psi_br, const double *psi_dp,
                          const i64 *rs, const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh) {
    const ClassLists cl = class_lists(in, dp, per_limb * (unsigned)in.n);
    const unsigned blocks = (unsigned)cl.in_blocks + per_limb * (unsigned)dp.n;
    switch (K) {
        case 1: launch_cols_mixed_rs_k<1>(blocks, st, base, g, cl, rsrc, psi_br, psi_dp, rs, ql, qh, kl, kh); break;
        case 2: launch_cols_mixed_rs_k<2>(blocks, st, base, g, cl, rsrc, psi_br, psi_dp, rs, ql, qh, kl, kh); break;
        case 3: launch_cols_mixed_rs_k<3>(blocks, st, base, g, cl, rsrc, psi_br, psi_dp, rs, ql, qh, kl, kh); break;
        case 4: launch_cols_mixed_rs_k<4>(blocks, st, base, g, cl, rsrc, psi_br, psi_dp, rs, ql, qh, kl, kh); break;
        case 5: launch_cols_mixed_rs_k<5>(blocks, st, base, g, cl, rsrc, psi_br, psi_dp, rs, ql, qh, kl, kh); break;
    }
}

// ---- last inverse pass + chain tail + key-switch digits in ONE launch (lf_intt_mul_digits) --------------------------------------
// cc_mult's x1 * y1 leaves its inverse transform only to be cut into mixed-radix digits (ks_digits_kernel: a Garner step per
// coefficient over the alpha limbs of a digit).  Where a digit's limbs fit a thread — alpha * 2^K words, K = logN - 12 trailing
// stages: silver 2 x 8, bronze 1 x 4 — the column thread of the inverse pass takes the columns of ALL limbs of its digit, runs
// the Garner step on the canonical words it holds and stores the digit state directly: the coefficient-domain product is never
// written, one launch (of silver's eleven) and its dependency gap disappear.  desc / tab = lf_ks_digits' tables.
// coefficient k of the thread (a compile-time index: as a `#pragma unroll` loop over k the 32-word shapes — <2, 8>, <3, 4>, <4, 2> —
// were left rolled by the optimizer, x[][k] became a dynamic index and the 256 bytes of x lived in scratch memory)
template <int AMAX, int R, int k>
__device__ __forceinline__ void garner_col(const i64 (&x)[AMAX][R], i64 *out, int row_start, int alpha, const i64 *__restrict__ Y,
                                           const i64 *__restrict__ Ls, int logN, int logC, const i64 *__restrict__ ql,
                                           const i64 *__restrict__ qh, const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    i64 st[AMAX];
#pragma unroll
    for (int i = 0; i < AMAX; ++i) st[i] = x[0][k];
    int lc = 0;
#pragma unroll
    for (int i = 0; i < AMAX - 1; ++i) {
        if (i + 1 < alpha) {
            const RowMod m = load_mod(ql, qh, kl, kh, row_start + i + 1);
            const i64 y = mm62s(x[i + 1][k] - st[i + 1], Y[i], m.q, m.k);
            st[i + 1] = y;
#pragma unroll
            for (int jj = i + 2; jj < AMAX; ++jj) {
                if (jj < alpha) {
                    const RowMod mj = load_mod(ql, qh, kl, kh, row_start + jj);
                    st[jj] += mm62s(y, Ls[lc], mj.q, mj.k);
                    ++lc;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < AMAX; ++i)
        if (i < alpha) out[((i64)(row_start + i) << logN) + ((i64)k << logC)] = st[i];
}

template <int AMAX, int R, int... Ks>
__device__ __forceinline__ void garner_cols(std::integer_sequence<int, Ks...>, const i64 (&x)[AMAX][R], i64 *out, int row_start, int alpha,
                                            const i64 *__restrict__ Y, const i64 *__restrict__ Ls, int logN, int logC,
                                            const i64 *__restrict__ ql, const i64 *__restrict__ qh, const i64 *__restrict__ kl,
                                            const i64 *__restrict__ kh) {
    (garner_col<AMAX, R, Ks>(x, out, row_start, alpha, Y, Ls, logN, logC, ql, qh, kl, kh), ...);
}

// limb i of the thread's digit through the last inverse pass (compile-time i, for the same reason)
template <int K, int AMAX, int... Is>
__device__ __forceinline__ void digit_cols(std::integer_sequence<int, Is...>, i64 (&x)[AMAX][1 << K], int poly, int row_start, int alpha,
                                           int chunk, const i64 *__restrict__ src, const PassGeom &g, const i64 *__restrict__ ipsi_br,
                                           const double *__restrict__ ipsi_dp, const i64 *__restrict__ Ninv,
                                           const i64 *__restrict__ ql, const i64 *__restrict__ qh, const i64 *__restrict__ kl,
                                           const i64 *__restrict__ kh) {
    auto one = [&](auto idx) {
        constexpr int i = decltype(idx)::value;
        if (i < alpha) {
            const int crow = row_start + i;
            const u64 q = ((u64)qh[crow] << 31) | (u64)ql[crow];
            // tail 2 (intt_exit_reduce): canonical coefficients, what ks_digits_kernel reads
            if (q < SMALL_PRIME_LIMIT) inv_cols_compute<true, K>(poly, crow, chunk, src, g, ipsi_br, ipsi_dp, Ninv, 2, ql, qh, kl, kh, x[i]);
            else inv_cols_compute<false, K>(poly, crow, chunk, src, g, ipsi_br, ipsi_dp, Ninv, 2, ql, qh, kl, kh, x[i]);
        }
    };
    (one(std::integral_constant<int, Is>{}), ...);
}

template <int K, int AMAX>
__global__ void __launch_bounds__(NTT_COL_THREADS) ntt_inv_cols_digits(const i64 *__restrict__ src, i64 *__restrict__ state, PassGeom g,
                                                                      const i64 *__restrict__ desc, const i64 *__restrict__ tab,
                                                                      const i64 *__restrict__ ipsi_br,
                                                                      const double *__restrict__ ipsi_dp,
                                                                      const i64 *__restrict__ Ninv, const i64 *__restrict__ ql,
                                                                      const i64 *__restrict__ qh, const i64 *__restrict__ kl,
                                                                      const i64 *__restrict__ kh) {
    constexpr int R = 1 << K;
    const int logC = g.logN - K;
    const int chunks = (1 << logC) / NTT_COL_THREADS;
    const int chunk = (int)blockIdx.x % chunks, p = (int)blockIdx.x / chunks, poly = (int)blockIdx.y;
    const int row_start = __builtin_amdgcn_readfirstlane((int)desc[p * 4 + 0]);
    const int alpha = __builtin_amdgcn_readfirstlane((int)desc[p * 4 + 1]);
    const i64 *Y = tab + desc[p * 4 + 2];
    const i64 *Ls = tab + desc[p * 4 + 3];
    if (alpha > AMAX) __builtin_trap();   // a digit wider than the caller's max_alpha: a broken table must not pass for a result
    i64 x[AMAX][R];
    digit_cols<K, AMAX>(std::make_integer_sequence<int, AMAX>{}, x, poly, row_start, alpha, chunk, src, g, ipsi_br, ipsi_dp, Ninv, ql, qh, kl, kh);
    // the Garner step of ks_digits_kernel (ckks_fused.hip; pre_extend, ckks_engine.py:654-705), per held coefficient
    i64 *out = state + ((i64)poly * g.rows << g.logN) + chunk * NTT_COL_THREADS + threadIdx.x;
    garner_cols<AMAX, R>(std::make_integer_sequence<int, R>{}, x, out, row_start, alpha, Y, Ls, g.logN, logC, ql, qh, kl, kh);
}

// forward transform of a stack; `rsrc` (optional): the column pass takes its input from a rescale on the fly
int ntt_forward(int64_t *a, int batch, int rows, int logN, const int64_t *psi_br, const double *psi_dp,
                const int64_t *q_host, const int64_t *Rs, int flags, const int64_t *ql, const int64_t *qh,
                const int64_t *kl, const int64_t *kh, int device, void *stream, const RescaleSrc *rsrc, int only_pass = 0,
                int64_t *ws = nullptr);

template <int K>
void launch_inv_cols_ws_k(unsigned blocks, hipStream_t st, const i64 *ws, const unsigned char *wflags, i64 *base,
                                 const PassGeom &g, const ClassLists &cl, const i64 *ipsi_br, const double *ipsi_dp, const i64 *Ninv,
                                 int tail, const i64 *ql, const i64 *qh, const i64 *kl, const i64 *kh) {
    hipLaunchKernelGGL((ntt_inv_cols_ws<K>), dim3(blocks), dim3(NTT_COL_THREADS), 0, st, ws, wflags, base, g, cl, ipsi_br, ipsi_dp,
                       Ninv, tail, ql, qh, kl, kh);
}

template <int K>
void launch_cols_ws_k(unsigned blocks, hipStream_t st, const i64 *a, i64 *ws, unsigned char *wflags, const PassGeom &g,
                      const ClassLists &cl, const int64_t *psi_br, const double *psi_dp, const i64 *rs, const int64_t *ql,
                      const int64_t *qh, const int64_t *kl, const int64_t *kh) {
    hipLaunchKernelGGL((ntt_fwd_cols_ws<K>), dim3(blocks), dim3(NTT_COL_THREADS), 0, st, a, ws, wflags, g, cl, (const i64 *)psi_br,
                       psi_dp, rs, (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
}

}  // namespace

extern "C" {

int lf_twiddle_dp(const int64_t *mont, double *out, int rows, int64_t N, const int64_t *ql, const int64_t *qh,
                  const int64_t *kl, const int64_t *kh, int device, void *stream) {
    if (rows < 0 || N < 1) return LF_ERR_ARG;
    if (rows == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    dim3 grid((unsigned)((N + 255) / 256), (unsigned)rows);
    hipLaunchKernelGGL(twiddle_dp_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const i64 *)mont, out, (i64)N,
                       (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
    return (int)hipGetLastError();
}

int lf_ntt(int64_t *a, int batch, int rows, int logN, const int64_t *psi_br, const double *psi_dp,
           const int64_t *q_host, const int64_t *Rs, int flags, const int64_t *_2q, const int64_t *ql,
           const int64_t *qh, const int64_t *kl, const int64_t *kh, int device, void *stream) {
    (void)_2q;
    return ntt_forward(a, batch, rows, logN, psi_br, psi_dp, q_host, Rs, flags, ql, qh, kl, kh, device, stream, nullptr);
}

int64_t lf_ntt_ws_words(int batch, int rows, int logN) {
    if (batch < 0 || rows < 0 || logN < 1 || logN > 2 * NTT_TILE_LOG_MAX) return -1;
    return ((int64_t)batch * rows << logN) + (int64_t)batch * rows * 8;   // the stack + 64 flag bytes per (polynomial, limb)
}

int lf_ntt_ws(int64_t *a, int64_t *ws, int batch, int rows, int logN, const int64_t *psi_br, const double *psi_dp,
              const int64_t *q_host, const int64_t *Rs, int flags, const int64_t *ql, const int64_t *qh, const int64_t *kl,
              const int64_t *kh, int device, void *stream) {
    if (flags & LF_NTT_RELAXED) return LF_ERR_ARG;   // exact transforms only (the relaxed ones live inside the fused ops)
    return ntt_forward(a, batch, rows, logN, psi_br, psi_dp, q_host, Rs, flags, ql, qh, kl, kh, device, stream, nullptr, 0, ws);
}

int lf_ntt_pass_ws(int64_t *a, int64_t *ws, int batch, int rows, int logN, const int64_t *psi_br, const double *psi_dp,
                   const int64_t *q_host, const int64_t *Rs, int flags, int which, const int64_t *ql, const int64_t *qh,
                   const int64_t *kl, const int64_t *kh, int device, void *stream) {
    if (which != 1 && which != 2) return LF_ERR_ARG;
    if (logN <= NTT_TILE_LOG_MAX || (flags & LF_NTT_RELAXED) || !ws) return LF_ERR_ARG;
    return ntt_forward(a, batch, rows, logN, psi_br, psi_dp, q_host, Rs, flags, ql, qh, kl, kh, device, stream, nullptr, which, ws);
}

/* Measurement entry (NOT one of the reference's ops, not used by the engine): launch exactly ONE of the two pass
 * kernels of a two-pass forward transform — which = 1 the column pass, 2 the tiled pass — with the grid it has
 * inside lf_ntt, so that bench.py / tools can time the dominant kernel alone.  The buffer is scratch afterwards. */
int lf_ntt_pass(int64_t *a, int batch, int rows, int logN, const int64_t *psi_br, const double *psi_dp,
                const int64_t *q_host, const int64_t *Rs, int flags, int which, const int64_t *ql, const int64_t *qh,
                const int64_t *kl, const int64_t *kh, int device, void *stream) {
    if (which != 1 && which != 2) return LF_ERR_ARG;
    if (logN <= NTT_TILE_LOG_MAX) return LF_ERR_ARG;   // single-pass sizes have nothing to split
    return ntt_forward(a, batch, rows, logN, psi_br, psi_dp, q_host, Rs, flags, ql, qh, kl, kh, device, stream, nullptr, which);
}

int lf_rescale_ntt(const int64_t *const *in, const int64_t *const *row0, int count, int64_t *x, int rows, int logN,
                   const int64_t *scales, int64_t round_at, const int64_t *psi_br, const double *psi_dp,
                   const int64_t *q_host, const int64_t *Rs, int flags, const int64_t *_2q, const int64_t *ql,
                   const int64_t *qh, const int64_t *kl, const int64_t *kh, int device, void *stream) {
    (void)_2q;
    if (count < 0 || count > LF_NTT_RS_MAX || rows < 0 || rows > MAX_LIST_ROWS || logN < 1 || logN > 2 * NTT_TILE_LOG_MAX)
        return LF_ERR_ARG;
    if ((flags & LF_NTT_RELAXED) && (!psi_dp || !q_host)) return LF_ERR_ARG;   // before anything is launched
    if (count == 0 || rows == 0) return 0;
    const int S1 = logN - NTT_TILE_LOG_MAX;
    if ((flags & (LF_NTT_ONLY_COLS | LF_NTT_ONLY_TILED)) && !(S1 >= 1 && S1 <= 5)) return LF_ERR_ARG;   // two-launch sizes only
    if (S1 >= 1 && S1 <= 5) {
        RescaleSrc rsrc;
        for (int i = 0; i < count; ++i) rsrc.in[i] = (const i64 *)in[i], rsrc.row0[i] = (const i64 *)row0[i];
        rsrc.scales = (const i64 *)scales;
        rsrc.round_at = (i64)round_at;
        // LF_NTT_ONLY_COLS / LF_NTT_ONLY_TILED: one of the two launches (the first reads `in` / `row0`, the second only `x`)
        const int only = (flags & LF_NTT_ONLY_COLS) ? 1 : (flags & LF_NTT_ONLY_TILED) ? 2 : 0;
        if ((flags & LF_NTT_ONLY_COLS) && (flags & LF_NTT_ONLY_TILED)) return LF_ERR_ARG;
        return ntt_forward(x, count, rows, logN, psi_br, psi_dp, q_host, Rs, flags, ql, qh, kl, kh, device, stream, &rsrc, only);
    }
    // sizes without a column pass: the two steps one after the other
    int64_t *outs[LF_NTT_RS_MAX];
    for (int i = 0; i < count; ++i) outs[i] = x + ((int64_t)i * rows << logN);
    if (int e = lf_rescale_batch(in, row0, outs, count, rows, (int64_t)1 << logN, scales, round_at, ql, qh, kl, kh, device, stream))
        return e;
    return ntt_forward(x, count, rows, logN, psi_br, psi_dp, q_host, Rs, flags, ql, qh, kl, kh, device, stream, nullptr);
}

}  // extern "C"

namespace {

int ntt_forward(int64_t *a, int batch, int rows, int logN, const int64_t *psi_br, const double *psi_dp,
                const int64_t *q_host, const int64_t *Rs, int flags, const int64_t *ql, const int64_t *qh,
                const int64_t *kl, const int64_t *kh, int device, void *stream, const RescaleSrc *rsrc, int only_pass,
                int64_t *ws) {
    if (batch < 0 || rows < 0 || rows > MAX_LIST_ROWS || logN < 1 || logN > 2 * NTT_TILE_LOG_MAX) return LF_ERR_ARG;
    const int relaxed = flags & LF_NTT_RELAXED;
    // relaxed arithmetic lives in the auxiliary table (fp64 twiddles, Shoup pairs), whose row layout follows the prime:
    // without the host primes every row would be taken for the integer class and read doubles as Shoup pairs
    if (relaxed && (!psi_dp || !q_host)) return LF_ERR_ARG;
    if (batch == 0 || rows == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    const int tl = logN < NTT_TILE_LOG_MAX ? logN : NTT_TILE_LOG_MAX;
    const int S1 = logN - tl;
    const int plain = (relaxed && (flags & LF_NTT_PLAIN)) ? 1 : 0;
    RowList dp, in;
    classify(rows, q_host, psi_dp, dp, in);
    hipStream_t st = (hipStream_t)stream;
    const bool mixed = rsrc || (dp.n && in.n);   // both classes in one launch per pass
    // only_pass = 1 | 2 (lf_ntt_pass, measurement only): launch only the column pass / only the tiled pass
    i64 *base = (i64 *)a;
    const int nb = batch;
    const unsigned per_row = (unsigned)nb << (logN - tl);
    // through a workspace (lf_ntt_ws): exact two-launch sizes whose tiled pass is the 4096-word one; anything else ignores `ws`
    const bool through_ws = ws && !relaxed && !rsrc && S1 >= 1 && S1 <= 5 && tl == NTT_TILE_LOG_MAX;
    unsigned char *wflags = through_ws ? reinterpret_cast<unsigned char *>(ws + ((i64)nb * rows << logN)) : nullptr;
    // the opening of cc_mult (rescale + relaxed plain-domain transform of the operand stack): the stack is internal to the
    // library — read by lf_intt_mul(_digits) and lf_relin_* only — and with LF_NTT_PLANES keeps its fp64-class rows as planes
    // (callers ask where lf_stack_planes() says so): 6 bytes per word on each of its four trips
    const bool xpl = (flags & LF_NTT_PLANES) != 0;
    if (xpl && !(rsrc && relaxed && plain && S1 >= 1 && S1 <= 5 && tl == NTT_TILE_LOG_MAX && dp.n && in.n)) return LF_ERR_ARG;
    for (int pass = (S1 > 0 ? 0 : 1); pass < 2; ++pass) {
        if (only_pass && pass + 1 != only_pass) continue;   // measurement only: time one pass kernel by itself
        PassGeom g = pass == 0 ? PassGeom{logN, tl, 1, S1, 0, tl - S1, rows, nb, relaxed, 0, plain}
                               : PassGeom{logN, tl, 0, tl, S1, 0, rows, nb, relaxed, 1, plain};
        const i64 *rs = (pass == 0 || S1 == 0) ? (const i64 *)Rs : nullptr;
        if (xpl) {   // cc_mult's operand stack: the column pass writes planes, the tiled pass transforms them in place
            const size_t bytes = ((size_t)nb * rows << logN) * 8;
            if (pass == 0) {
                g.pln = PLN_OUT;
                lf_fmt_note(base, bytes, LF_FMT_PLANES);
            } else if (int e = lf_fmt_expect(base, bytes, LF_FMT_PLANES)) {
                return e;
            }
        } else if (rsrc && relaxed && plain) {
            const size_t bytes = ((size_t)nb * rows << logN) * 8;
            if (pass == 0) lf_fmt_note(base, bytes, LF_FMT_RAW);
            else if (int e = lf_fmt_expect(base, bytes, LF_FMT_RAW)) return e;
        }
        if (through_ws) {
            // the column pass is HBM-bound with half its issue slots free, the tiled pass issue-bound: through a workspace the
            // column pass takes ONE STAGE MORE (up to 5: 32 words per thread, 141 registers, still at the HBM rate) and the
            // tiles skip their first (lf_tune LF_TUNE_WS_EXTRA_STAGE)
            const bool extra = lf_g_ws_extra_stage && S1 <= 4;
            const int Kc = S1 + (extra ? 1 : 0);
            {   // the two launches may be two calls (lf_ntt_pass_ws): the tiled pass must find the split the column pass left
                const size_t bytes = ((size_t)nb * rows << logN) * 8;
                const int fmt = extra ? LF_FMT_WS_SPLIT1 : LF_FMT_WS_SPLIT0;
                if (pass == 0) lf_fmt_note(ws, bytes, fmt);
                else if (int e = lf_fmt_expect(ws, bytes, fmt)) return e;
            }
            if (pass == 0) {
                const unsigned per_limb = (unsigned)nb * ((1u << (logN - Kc)) / NTT_COL_THREADS);
                const ClassLists cl = class_lists(in, dp, per_limb * (unsigned)in.n);
                const unsigned blocks = (unsigned)cl.in_blocks + per_limb * (unsigned)dp.n;
                switch (Kc) {
                    case 1: launch_cols_ws_k<1>(blocks, st, base, (i64 *)ws, wflags, g, cl, psi_br, psi_dp, rs, ql, qh, kl, kh); break;
                    case 2: launch_cols_ws_k<2>(blocks, st, base, (i64 *)ws, wflags, g, cl, psi_br, psi_dp, rs, ql, qh, kl, kh); break;
                    case 3: launch_cols_ws_k<3>(blocks, st, base, (i64 *)ws, wflags, g, cl, psi_br, psi_dp, rs, ql, qh, kl, kh); break;
                    case 4: launch_cols_ws_k<4>(blocks, st, base, (i64 *)ws, wflags, g, cl, psi_br, psi_dp, rs, ql, qh, kl, kh); break;
                    case 5: launch_cols_ws_k<5>(blocks, st, base, (i64 *)ws, wflags, g, cl, psi_br, psi_dp, rs, ql, qh, kl, kh); break;
                }
            } else {
                launch_pass16_ws(nb, st, (const i64 *)ws, wflags, base, g, in, dp, (const i64 *)psi_br, psi_dp, (const i64 *)ql,
                                 (const i64 *)qh, (const i64 *)kl, (const i64 *)kh, extra);
            }
            continue;
        }
        if (pass == 0 && (S1 <= 4 || (S1 == 5 && rsrc))) {   // leading stages: one register step per column (logN 17: only the
            // rescale form — the general in-place body needs 256 registers at 32 words per thread; lf_ntt_ws has its own)
            const unsigned col_blocks = (unsigned)nb * ((1u << (logN - S1)) / NTT_COL_THREADS);
            if (rsrc) {
                launch_cols_mixed_rs(S1, col_blocks, st, base, g, in, dp, *rsrc, psi_br, psi_dp, rs, ql, qh, kl, kh);
                continue;
            }
            if (mixed) {
                launch_cols_mixed(S1, col_blocks, st, base, g, in, dp, psi_br, psi_dp, rs, ql, qh, kl, kh);
                continue;
            }
            if (dp.n) launch_cols<true>(S1, col_blocks * dp.n, st, base, g, dp, psi_br, psi_dp, rs, ql, qh, kl, kh);
            if (in.n) launch_cols<false>(S1, col_blocks * in.n, st, base, g, in, psi_br, psi_dp, rs, ql, qh, kl, kh);
            continue;
        }
        if (pass == 1 && S1 > 0 && tl == NTT_TILE_LOG_MAX && rs == nullptr) {
            // contiguous 12-stage pass: 16 words per thread (ckks_ntt_tile16.h), either class or both
            launch_pass16(false, relaxed, nb, st, base, base, g, in, dp, (const i64 *)psi_br, psi_dp, (const i64 *)ql,
                          (const i64 *)qh, (const i64 *)kl, (const i64 *)kh, nullptr, xpl);
            continue;
        }
        if (mixed) {
            const ClassLists cl = class_lists(in, dp, per_row * (unsigned)in.n);
            LF_LAUNCH_MIXED(ntt_fwd_pass_mixed, relaxed, dim3((unsigned)cl.in_blocks + per_row * dp.n), dim3(NTT_THREADS), 0, st,
                               base, g, cl, (const i64 *)psi_br, psi_dp, rs, (const i64 *)ql, (const i64 *)qh,
                               (const i64 *)kl, (const i64 *)kh);
            continue;
        }
        if (dp.n)
            LF_LAUNCH_CLASS(ntt_fwd_pass, true, relaxed, dim3(per_row * dp.n), dim3(NTT_THREADS), 0, st, base, g, dp,
                               (const i64 *)psi_br, psi_dp, rs, (const i64 *)ql, (const i64 *)qh, (const i64 *)kl,
                               (const i64 *)kh);
        if (in.n)
            LF_LAUNCH_CLASS(ntt_fwd_pass, false, relaxed, dim3(per_row * in.n), dim3(NTT_THREADS), 0, st, base, g, in,
                               (const i64 *)psi_br, psi_dp, rs, (const i64 *)ql, (const i64 *)qh, (const i64 *)kl,
                               (const i64 *)kh);
    }
    return (int)hipGetLastError();
}

}  // namespace

extern "C" {

// inverse transform of a [batch][rows][N] stack in place on `a`; ms (relaxed, logN >= 13 only): the transform of the
// element-wise product of two stacks, read from `src` / ms->b by the first pass and written to `a`
static int intt_impl(int64_t *a, const int64_t *src, const MulSrc *ms, int batch, int rows, int logN, const int64_t *ipsi_br,
                     const double *ipsi_dp, const int64_t *q_host, const int64_t *Ninv, int tail, int flags, const int64_t *ql,
                     const int64_t *qh, const int64_t *kl, const int64_t *kh, int device, void *stream, int64_t *ws = nullptr) {
    if (batch < 0 || rows < 0 || rows > MAX_LIST_ROWS || logN < 1 || logN > 2 * NTT_TILE_LOG_MAX || tail < 0 || tail > 3)
        return LF_ERR_ARG;
    const int relaxed = flags & LF_NTT_RELAXED;
    if (relaxed && tail < 2) return LF_ERR_ARG;   // relaxed results are canonical residues only
    if (relaxed && (!ipsi_dp || !q_host)) return LF_ERR_ARG;   // see ntt_forward
    if (ms && (!relaxed || logN <= NTT_TILE_LOG_MAX || !ms->b || !src)) return LF_ERR_ARG;
    if (batch == 0 || rows == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    const int tl = logN < NTT_TILE_LOG_MAX ? logN : NTT_TILE_LOG_MAX;
    const int SB = logN - tl;
    const int plain = (relaxed && (flags & LF_NTT_PLAIN)) ? 1 : 0;
    RowList dp, in;
    classify(rows, q_host, ipsi_dp, dp, in);
    hipStream_t st = (hipStream_t)stream;
    const bool mixed = dp.n && in.n;   // both classes in one launch per pass
    // LF_NTT_PLANES (product-on-load only): the factors are stacks whose fp64-class rows are planes (lf_rescale_ntt with the flag)
    const bool mpl = (flags & LF_NTT_PLANES) != 0;
    if (mpl && !(ms && relaxed && plain && mixed && SB >= 1 && tl == NTT_TILE_LOG_MAX)) return LF_ERR_ARG;
    if (ms) {
        const int want = mpl ? LF_FMT_PLANES : LF_FMT_RAW;
        for (int t = 0; t < batch; ++t) {
            if (int e = lf_fmt_expect(src + t * ms->a_stride, ((size_t)rows << logN) * 8, want)) return e;
            if (int e = lf_fmt_expect(ms->b + t * ms->b_stride, ((size_t)rows << logN) * 8, want)) return e;
        }
    }
    // (splitting a batch so that its two passes meet in the Infinity Cache was measured on MI355X and only adds launch
    // tails, DESIGN.md §4: the whole batch goes through each pass in one launch)
    i64 *base = (i64 *)a;
    const int nb = batch;
    const unsigned per_row = (unsigned)nb << (logN - tl);
    for (int pass = 0; pass < (SB > 0 ? 2 : 1); ++pass) {
        const PassGeom g = pass == 0 ? PassGeom{logN, tl, 0, tl, 0, 0, rows, nb, relaxed, SB == 0, plain}
                                     : PassGeom{logN, tl, 1, SB, tl, tl - SB, rows, nb, relaxed, 1, plain};
        const int t = g.last ? tail : TAIL_NONE;
        // through a workspace (lf_intt_ws): exact two-launch sizes; the tiled pass writes it, the column pass reads it
        if (ws && !relaxed && !ms && SB >= 1 && SB <= 5) {
            unsigned char *wflags = reinterpret_cast<unsigned char *>(ws + ((i64)nb * rows << logN));
            if (pass == 0) {
                const ClassLists cl = class_lists(in, dp, per_row * (unsigned)in.n);
                hipLaunchKernelGGL(ntt_pass16_inv_ws, dim3((unsigned)cl.in_blocks + per_row * (unsigned)dp.n), dim3(NTT16_THREADS), 0, st,
                                   (const i64 *)base, (i64 *)ws, wflags, g, cl, (const i64 *)ipsi_br, ipsi_dp, (const i64 *)ql,
                                   (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
            } else {
                const unsigned per_limb = (unsigned)nb * ((1u << (logN - SB)) / NTT_COL_THREADS);
                const ClassLists cl = class_lists(in, dp, per_limb * (unsigned)in.n);
                const unsigned blocks = (unsigned)cl.in_blocks + per_limb * (unsigned)dp.n;
#define LF_ICW(KK)                                                                                                             \
    launch_inv_cols_ws_k<KK>(blocks, st, (const i64 *)ws, wflags, base, g, cl, (const i64 *)ipsi_br, ipsi_dp, (const i64 *)Ninv, t, \
                             (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh)
                if (SB == 1) LF_ICW(1); else if (SB == 2) LF_ICW(2); else if (SB == 3) LF_ICW(3); else if (SB == 4) LF_ICW(4); else LF_ICW(5);
#undef LF_ICW
            }
            continue;
        }
        if (pass == 1 && (SB <= 4 || (SB == 5 && mixed))) {   // trailing stages + chain tail: one register step per column
            if (mixed) {
                launch_inv_cols_mixed(SB, nb, st, base, g, in, dp, (const i64 *)ipsi_br, ipsi_dp, (const i64 *)Ninv, t,
                                      (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
                continue;
            }
            if (dp.n)
                launch_inv_cols<true>(SB, nb, st, base, g, dp, (const i64 *)ipsi_br, ipsi_dp, (const i64 *)Ninv, t,
                                      (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
            if (in.n)
                launch_inv_cols<false>(SB, nb, st, base, g, in, (const i64 *)ipsi_br, ipsi_dp, (const i64 *)Ninv, t,
                                       (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
            continue;
        }
        if (pass == 0 && SB > 0 && tl == NTT_TILE_LOG_MAX) {
            launch_pass16(true, relaxed, nb, st, ms ? (const i64 *)src : (const i64 *)base, base, g, in, dp, (const i64 *)ipsi_br,
                          ipsi_dp, (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh, ms, mpl);
            continue;
        }
        if (mixed) {
            const ClassLists cl = class_lists(in, dp, per_row * (unsigned)in.n);
            LF_LAUNCH_MIXED(ntt_inv_pass_mixed, relaxed, dim3((unsigned)cl.in_blocks + per_row * dp.n), dim3(NTT_THREADS), 0, st,
                               (const i64 *)base, base, g, cl, (const i64 *)ipsi_br, ipsi_dp, (const i64 *)Ninv, t,
                               (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
            continue;
        }
        if (dp.n)
            LF_LAUNCH_CLASS(ntt_inv_pass_io, true, relaxed, dim3(per_row * dp.n), dim3(NTT_THREADS), 0, st, (const i64 *)base, base, g, dp,
                               (const i64 *)ipsi_br, ipsi_dp, (const i64 *)Ninv, t, (const i64 *)ql, (const i64 *)qh,
                               (const i64 *)kl, (const i64 *)kh);
        if (in.n)
            LF_LAUNCH_CLASS(ntt_inv_pass_io, false, relaxed, dim3(per_row * in.n), dim3(NTT_THREADS), 0, st, (const i64 *)base, base, g, in,
                               (const i64 *)ipsi_br, ipsi_dp, (const i64 *)Ninv, t, (const i64 *)ql, (const i64 *)qh,
                               (const i64 *)kl, (const i64 *)kh);
    }
    return (int)hipGetLastError();
}

int lf_intt(int64_t *a, int batch, int rows, int logN, const int64_t *ipsi_br, const double *ipsi_dp,
            const int64_t *q_host, const int64_t *Ninv, int tail, int flags, const int64_t *_2q, const int64_t *ql,
            const int64_t *qh, const int64_t *kl, const int64_t *kh, int device, void *stream) {
    (void)_2q;
    return intt_impl(a, nullptr, nullptr, batch, rows, logN, ipsi_br, ipsi_dp, q_host, Ninv, tail, flags, ql, qh, kl, kh, device, stream);
}

int lf_intt_ws(int64_t *a, int64_t *ws, int batch, int rows, int logN, const int64_t *ipsi_br, const double *ipsi_dp,
               const int64_t *q_host, const int64_t *Ninv, int tail, int flags, const int64_t *ql, const int64_t *qh,
               const int64_t *kl, const int64_t *kh, int device, void *stream) {
    if (flags & LF_NTT_RELAXED) return LF_ERR_ARG;
    return intt_impl(a, nullptr, nullptr, batch, rows, logN, ipsi_br, ipsi_dp, q_host, Ninv, tail, flags, ql, qh, kl, kh, device, stream, ws);
}

/* lf_intt_mul followed by lf_ks_digits of its result, the coefficient-domain product never written: the first inverse pass forms
 * a * b into `scratch` ([batch][rows][N], the tiled pass's output), the column pass of every digit's limbs ends in the Garner
 * step and stores the digit states ([batch][rows][N], state_stride words apart).  Returns LF_ERR_ARG — nothing launched — when
 * the shape does not qualify (two-pass degrees with max_alpha * 2^(logN - 12) <= 32 words per thread; 64 — gold's 4 limbs x 16 words,
 * 200 VGPRs at 2 waves per SIMD — for batches only: measured, in one process, gold cc_mult x 16 318 -> 311 us per ciphertext,
 * but a single cc_mult 414 -> 430 us: two waves per SIMD do not hide the latency of a launch that small): the caller then
 * takes lf_intt_mul + lf_ks_digits.  Same words in `state` either way. */
int lf_intt_mul_digits(int64_t *scratch, const int64_t *a, int64_t a_stride, const int64_t *b, int64_t b_stride, int batch, int rows,
                       int logN, int64_t *state, int nparts, int max_alpha, const int64_t *desc, const int64_t *tab,
                       const int64_t *ipsi_br, const double *ipsi_dp, const int64_t *q_host, const int64_t *Ninv, int flags,
                       const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh, int device, void *stream) {
    const int S1 = logN - NTT_TILE_LOG_MAX;
    if (!scratch || !a || !b || !state || !desc || !tab || !ipsi_dp || !q_host || batch < 1 || batch > LF_BATCH_MAX || rows < 1 ||
        rows > MAX_LIST_ROWS || nparts < 1 || max_alpha < 1 || max_alpha > KS_MAX_ALPHA || S1 < 1 || S1 > 4 || (max_alpha << S1) > ((S1 == 4 && batch >= 2) ? 64 : 32) ||
        !(flags & LF_NTT_RELAXED))
        return LF_ERR_ARG;
    if (int e = lf_set_device(device)) return e;
    const int tl = NTT_TILE_LOG_MAX;
    const int plain = (flags & LF_NTT_PLAIN) ? 1 : 0;
    RowList dp, in;
    classify(rows, q_host, ipsi_dp, dp, in);
    hipStream_t st = (hipStream_t)stream;
    const MulSrc ms{(const i64 *)b, (i64)a_stride, (i64)b_stride};
    const bool mpl = (flags & LF_NTT_PLANES) != 0;   // the factors' fp64-class rows are planes (lf_rescale_ntt with the flag)
    if (mpl && !(plain && dp.n && in.n)) return LF_ERR_ARG;
    for (int t = 0; t < batch; ++t) {
        if (int e = lf_fmt_expect(a + t * a_stride, ((size_t)rows << logN) * 8, mpl ? LF_FMT_PLANES : LF_FMT_RAW)) return e;
        if (int e = lf_fmt_expect(b + t * b_stride, ((size_t)rows << logN) * 8, mpl ? LF_FMT_PLANES : LF_FMT_RAW)) return e;
    }
    const PassGeom g0{logN, tl, 0, tl, 0, 0, rows, batch, 1, 0, plain};
    launch_pass16(true, 1, batch, st, (const i64 *)a, (i64 *)scratch, g0, in, dp, (const i64 *)ipsi_br, ipsi_dp, (const i64 *)ql,
                  (const i64 *)qh, (const i64 *)kl, (const i64 *)kh, &ms, mpl);
    const PassGeom g1{logN, tl, 1, S1, tl, tl - S1, rows, batch, 1, 1, plain};
    const dim3 grid((unsigned)nparts * ((1u << (logN - S1)) / NTT_COL_THREADS), (unsigned)batch), block(NTT_COL_THREADS);
    const int amax = max_alpha <= 1 ? 1 : max_alpha <= 2 ? 2 : max_alpha <= 4 ? 4 : 8;
#define LF_ICD(KK, AA)                                                                                                       \
    hipLaunchKernelGGL((ntt_inv_cols_digits<KK, AA>), grid, block, 0, st, (const i64 *)scratch, (i64 *)state, g1, (const i64 *)desc, \
                       (const i64 *)tab, (const i64 *)ipsi_br, ipsi_dp, (const i64 *)Ninv, (const i64 *)ql, (const i64 *)qh,     \
                       (const i64 *)kl, (const i64 *)kh)
    if (S1 == 1) { if (amax == 1) LF_ICD(1, 1); else if (amax == 2) LF_ICD(1, 2); else if (amax == 4) LF_ICD(1, 4); else LF_ICD(1, 8); }
    else if (S1 == 2) { if (amax == 1) LF_ICD(2, 1); else if (amax == 2) LF_ICD(2, 2); else if (amax == 4) LF_ICD(2, 4); else LF_ICD(2, 8); }
    else if (S1 == 3) { if (amax == 1) LF_ICD(3, 1); else if (amax == 2) LF_ICD(3, 2); else LF_ICD(3, 4); }
    else { if (amax == 1) LF_ICD(4, 1); else if (amax == 2) LF_ICD(4, 2); else LF_ICD(4, 4); }   // (gold: 4 limbs x 16 words per thread)
#undef LF_ICD
    return (int)hipGetLastError();
}

int lf_intt_mul(int64_t *dst, const int64_t *a, int64_t a_stride, const int64_t *b, int64_t b_stride, int batch, int rows, int logN,
                const int64_t *ipsi_br, const double *ipsi_dp, const int64_t *q_host, const int64_t *Ninv, int tail, int flags,
                const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh, int device, void *stream) {
    if (!a || !b || !dst) return LF_ERR_ARG;
    const MulSrc ms{(const i64 *)b, (i64)a_stride, (i64)b_stride};
    return intt_impl(dst, a, &ms, batch, rows, logN, ipsi_br, ipsi_dp, q_host, Ninv, tail, flags, ql, qh, kl, kh, device, stream);
}

}  // extern "C"
