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
#include "../../include/ckks_hip.h"
#include "ckks_ntt_core.h"
#include "ckks_ntt_tile16.h"
#include <stdlib.h>
#include <mutex>

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

// The two arithmetic classes of one transform are independent (disjoint limbs), but launches on one
// stream serialise.  The integer class (a handful of 60-bit limbs) therefore runs on a per-device side
// stream, forked from and joined back into the caller's stream with events, so its few blocks overlap
// the fp64 class instead of adding a tail after every pass.
struct SideStream {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};

SideStream *side_stream(int device) {
    static std::mutex mu;
    static SideStream table[64];
    int dev = device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    SideStream &s = table[dev];
    if (!s.stream) {
        if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) { s.stream = nullptr; return nullptr; }
        if (hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.join, hipEventDisableTiming) != hipSuccess) return nullptr;
    }
    return &s;
}

// polynomials per launch group of the inverse transform: the whole batch (splitting a batch so that its two passes
// meet in the Infinity Cache was measured on MI355X and only adds launch tails)
int chunk_polys(int batch, int, int, bool) { return batch; }

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
    }
}

// ---- chunked forward transform: tiled pass of one chunk and column pass of the NEXT chunk in one launch ---------
// The column pass is HBM-bound with idle issue slots, the tiled pass issue-bound with idle HBM; launches on one
// stream serialise and launches on two streams only overlap in their tails (measured).  So a large
// batch is cut into chunks and launch k carries BOTH roles: blocks alternate in runs of 8 (one per XCD, so every
// XCD — every CU — holds a mix) between tiles of chunk k-1 and columns of chunk k.  The dependency (tiles of a
// chunk after its columns) is the launch order.  `ratio` column blocks per tile block: 16 >> K.
struct DuoGeom {
    PassGeom tile, col;
    int tile_in_blocks, tile_in_real, tile_blocks;
    int col_in_blocks, col_in_real, col_blocks;
    int ratio;
};

template <int K, bool RLX>
__global__ void __launch_bounds__(NTT16_THREADS, 4) ntt_fwd_duo(i64 *tile_base, i64 *col_base, DuoGeom d, RowList in, RowList dp,
                                                                 const i64 *__restrict__ psi_br, const double *__restrict__ psi_dp,
                                                                 const i64 *__restrict__ Rs, const i64 *__restrict__ ql,
                                                                 const i64 *__restrict__ qh, const i64 *__restrict__ kl,
                                                                 const i64 *__restrict__ kh) {
    __shared__ i64 sm[NTT16_LDS_WORDS + 1];
    const int x = blockIdx.x & 7, oct = blockIdx.x >> 3;
    // ratio 1 (logN 16): strict alternation would hand every other workgroup of an XCD — i.e. one role — to the same
    // half of its shader engines / CUs (the dispatcher walks them round-robin; measured: each role then runs on half
    // the chip).  The Thue-Morse order (role = parity of the bits of the index) is balanced on every residue class
    // modulo a power of two, and the pair (2i, 2i+1) always holds one block of each role, so the rank is oct >> 1.
    const int period = 1 + d.ratio;
    int pos, grp;
    if (d.ratio == 1) pos = __builtin_popcount((unsigned)oct) & 1, grp = oct >> 1;
    else pos = oct % period, grp = oct / period;
    if (pos == 0) {
        const int b = grp * 8 + x;
        if (b >= d.tile_blocks) return;
        if (b < d.tile_in_blocks) {
            if (b < d.tile_in_real) pass16_body<false, RLX, false>(sm, b, tile_base, tile_base, d.tile, in, psi_br, psi_dp, ql, qh, kl, kh);
        } else {
            pass16_body<true, RLX, false>(sm, b - d.tile_in_blocks, tile_base, tile_base, d.tile, dp, psi_br, psi_dp, ql, qh, kl, kh);
        }
    } else {
        const int b = (grp * d.ratio + pos - 1) * 8 + x;
        if (b >= d.col_blocks) return;
        if (b < d.col_in_blocks) {
            if (b < d.col_in_real) fwd_cols_body<false, K>(b, col_base, d.col, in, psi_br, psi_dp, Rs, ql, qh, kl, kh);
        } else {
            fwd_cols_body<true, K>(b - d.col_in_blocks, col_base, d.col, dp, psi_br, psi_dp, Rs, ql, qh, kl, kh);
        }
    }
}

template <int K>
void launch_duo_k(bool relaxed, unsigned blocks, hipStream_t st, i64 *tile_base, i64 *col_base, const DuoGeom &d,
                  const RowList &in, const RowList &dp, const int64_t *psi_br, const double *psi_dp, const i64 *rs,
                  const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh) {
    if (relaxed)
        hipLaunchKernelGGL((ntt_fwd_duo<K, true>), dim3(blocks), dim3(NTT16_THREADS), 0, st, tile_base, col_base, d, in, dp,
                           (const i64 *)psi_br, psi_dp, rs, (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
    else
        hipLaunchKernelGGL((ntt_fwd_duo<K, false>), dim3(blocks), dim3(NTT16_THREADS), 0, st, tile_base, col_base, d, in, dp,
                           (const i64 *)psi_br, psi_dp, rs, (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
}

// tiles of `tile_polys` polynomials at tile_base + columns of `col_polys` polynomials at col_base
void launch_duo(int K, bool relaxed, hipStream_t st, i64 *tile_base, int tile_polys, i64 *col_base, int col_polys,
                const PassGeom &gt, const PassGeom &gc, const RowList &in, const RowList &dp, const int64_t *psi_br,
                const double *psi_dp, const i64 *rs, const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh) {
    DuoGeom d;
    d.tile = gt, d.col = gc;
    d.tile.batch = tile_polys, d.col.batch = col_polys;
    const unsigned per_row = (unsigned)tile_polys << K, per_limb = (unsigned)col_polys * ((1u << NTT_TILE_LOG_MAX) / NTT_COL_THREADS);
    d.tile_in_real = (int)(per_row * (unsigned)in.n), d.tile_in_blocks = (d.tile_in_real + 7) & ~7;
    d.tile_blocks = d.tile_in_blocks + (int)(per_row * (unsigned)dp.n);
    d.col_in_real = (int)(per_limb * (unsigned)in.n), d.col_in_blocks = (d.col_in_real + 7) & ~7;
    d.col_blocks = d.col_in_blocks + (int)(per_limb * (unsigned)dp.n);
    d.ratio = 16 >> K;
    const unsigned tile_octs = ((unsigned)d.tile_blocks + 7u) / 8u, col_octs = ((unsigned)d.col_blocks + 7u) / 8u;
    const unsigned col_groups = (col_octs + (unsigned)d.ratio - 1u) / (unsigned)d.ratio;
    const unsigned groups = tile_octs > col_groups ? tile_octs : col_groups;
    const unsigned blocks = groups * (1u + (unsigned)d.ratio) * 8u;
    switch (K) {
        case 1: launch_duo_k<1>(relaxed, blocks, st, tile_base, col_base, d, in, dp, psi_br, psi_dp, rs, ql, qh, kl, kh); break;
        case 2: launch_duo_k<2>(relaxed, blocks, st, tile_base, col_base, d, in, dp, psi_br, psi_dp, rs, ql, qh, kl, kh); break;
        case 3: launch_duo_k<3>(relaxed, blocks, st, tile_base, col_base, d, in, dp, psi_br, psi_dp, rs, ql, qh, kl, kh); break;
        case 4: launch_duo_k<4>(relaxed, blocks, st, tile_base, col_base, d, in, dp, psi_br, psi_dp, rs, ql, qh, kl, kh); break;
    }
}

// forward transform of a stack; `rsrc` (optional): the column pass takes its input from a rescale on the fly
int ntt_forward(int64_t *a, int batch, int rows, int logN, const int64_t *psi_br, const double *psi_dp,
                const int64_t *q_host, const int64_t *Rs, int flags, const int64_t *ql, const int64_t *qh,
                const int64_t *kl, const int64_t *kh, int device, void *stream, const RescaleSrc *rsrc, int only_pass = 0);

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
    if (count == 0 || rows == 0) return 0;
    const int S1 = logN - NTT_TILE_LOG_MAX;
    if (S1 >= 1 && S1 <= 4 && cols_enabled()) {
        RescaleSrc rsrc;
        for (int i = 0; i < count; ++i) rsrc.in[i] = (const i64 *)in[i], rsrc.row0[i] = (const i64 *)row0[i];
        rsrc.scales = (const i64 *)scales;
        rsrc.round_at = (i64)round_at;
        return ntt_forward(x, count, rows, logN, psi_br, psi_dp, q_host, Rs, flags, ql, qh, kl, kh, device, stream, &rsrc);
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
                const int64_t *kl, const int64_t *kh, int device, void *stream, const RescaleSrc *rsrc, int only_pass) {
    if (batch < 0 || rows < 0 || rows > MAX_LIST_ROWS || logN < 1 || logN > 2 * NTT_TILE_LOG_MAX) return LF_ERR_ARG;
    if (batch == 0 || rows == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    const int tl = logN < NTT_TILE_LOG_MAX ? logN : NTT_TILE_LOG_MAX;
    const int S1 = logN - tl;
    const int relaxed = flags & LF_NTT_RELAXED;
    const int plain = (relaxed && (flags & LF_NTT_PLAIN)) ? 1 : 0;
    if (relaxed && !psi_dp) return LF_ERR_ARG;   // relaxed arithmetic lives in the auxiliary table (fp64 twiddles, Shoup pairs)
    RowList dp, in;
    classify(rows, q_host, psi_dp, dp, in);
    hipStream_t st = (hipStream_t)stream;
    const bool mixed = rsrc || (dp.n && in.n && mixed_enabled());   // both classes in one launch per pass
    // Two-pass transforms of a large batch run as LF_NTT_PIPE chunks alternating between the caller's stream and a
    // side stream: the column pass of one chunk (HBM-bound, few VALU cycles) shares the chip with the tiled pass of
    // its neighbour (VALU-issue-bound, HBM half idle) instead of the two running back to back.  Both passes of a
    // chunk stay on one stream, so the dependency needs no event; fork / join events fence the side stream.
#ifndef LF_NTT_PIPE
#define LF_NTT_PIPE 1
#endif
#ifndef LF_NTT_PIPE_MIN_POLYS
#define LF_NTT_PIPE_MIN_POLYS 16   // polynomials per chunk below which launch tails eat the overlap
#endif
#ifndef LF_NTT_DUO
#define LF_NTT_DUO 1               // chunks of a large batch whose passes are co-scheduled (ntt_fwd_duo); 1 = off.
                                   // Parity-green and measured on MI355X at 2 / 4 / 8 chunks: 1.75-1.80 ms per step against
                                   // 1.79 ms — the two roles share the CU's register file (4 x 128 VGPRs per SIMD lane), so
                                   // each runs at the speed its share of the wave slots allows and the sum is unchanged
                                   // (DESIGN.md §4).  Kept as a compile-time switch.
#endif
#ifndef LF_NTT_DUO_MIN_BLOCKS
#define LF_NTT_DUO_MIN_BLOCKS 4096 // tile blocks per chunk below which the extra launch tails eat the overlap
#endif
    if (LF_NTT_DUO > 1 && LF_TILE16 && S1 >= 1 && S1 <= 4 && tl == NTT_TILE_LOG_MAX && !rsrc && !only_pass) {
        const int min_polys = (LF_NTT_DUO_MIN_BLOCKS + (rows << S1) - 1) / (rows << S1);
        int nd = batch / min_polys;
        if (nd > LF_NTT_DUO) nd = LF_NTT_DUO;
        if (nd >= 2) {
            const PassGeom gc{logN, tl, 1, S1, 0, tl - S1, rows, 0, relaxed, 0, plain};
            const PassGeom gt{logN, tl, 0, tl, S1, 0, rows, 0, relaxed, 1, plain, regtile_disabled()};
            const int chunk = (batch + nd - 1) / nd;
            i64 *prev = nullptr;
            int prev_nb = 0;
            for (int b0 = 0; b0 < batch; b0 += chunk) {
                const int nb = batch - b0 < chunk ? batch - b0 : chunk;
                i64 *base = (i64 *)a + ((i64)b0 * rows << logN);
                if (!prev) {
                    PassGeom g = gc;
                    g.batch = nb;
                    launch_cols_mixed(S1, (unsigned)nb * ((1u << tl) / NTT_COL_THREADS), st, base, g, in, dp, psi_br, psi_dp,
                                      (const i64 *)Rs, ql, qh, kl, kh);
                } else {
                    launch_duo(S1, relaxed != 0, st, prev, prev_nb, base, nb, gt, gc, in, dp, psi_br, psi_dp, (const i64 *)Rs,
                               ql, qh, kl, kh);
                }
                prev = base, prev_nb = nb;
            }
            PassGeom g = gt;
            g.batch = prev_nb;
            launch_pass16(false, relaxed, prev_nb, st, prev, prev, g, in, dp, (const i64 *)psi_br, psi_dp, (const i64 *)ql,
                          (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
            return (int)hipGetLastError();
        }
    }
    int nchunks = 1;
    if (LF_NTT_PIPE > 1 && S1 > 0 && !rsrc && !only_pass && batch >= 2 * LF_NTT_PIPE_MIN_POLYS) {
        nchunks = LF_NTT_PIPE;
        while (nchunks > 1 && batch / nchunks < LF_NTT_PIPE_MIN_POLYS) --nchunks;
    }
    SideStream *side = nchunks > 1 ? side_stream(device) : nullptr;
    if (side) {
        (void)hipEventRecord(side->fork, st);
        (void)hipStreamWaitEvent(side->stream, side->fork, 0);
    } else {
        nchunks = 1;
    }
    hipStream_t st_main = st;
    // only_pass = 1 | 2 (lf_ntt_pass, measurement only): launch only the column pass / only the tiled pass
    const int chunk = (batch + nchunks - 1) / nchunks;
    for (int b0 = 0, ci = 0; b0 < batch; b0 += chunk, ++ci) {
        const int nb = batch - b0 < chunk ? batch - b0 : chunk;
        i64 *base = (i64 *)a + ((i64)b0 * rows << logN);
        st = (side && (ci & 1)) ? side->stream : st_main;
        hipStream_t st_int = st;
        const unsigned per_row = (unsigned)nb << (logN - tl);
        for (int pass = (S1 > 0 ? 0 : 1); pass < 2; ++pass) {
            if (only_pass && pass + 1 != only_pass) continue;   // measurement only: time one pass kernel by itself
            const PassGeom g = pass == 0 ? PassGeom{logN, tl, 1, S1, 0, tl - S1, rows, nb, relaxed, 0, plain}
                                         : PassGeom{logN, tl, 0, tl, S1, 0, rows, nb, relaxed, 1, plain, regtile_disabled()};
            const i64 *rs = (pass == 0 || S1 == 0) ? (const i64 *)Rs : nullptr;
            if (pass == 0 && S1 <= 4 && cols_enabled()) {   // leading stages: one register step per column
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
                if (in.n) launch_cols<false>(S1, col_blocks * in.n, st_int, base, g, in, psi_br, psi_dp, rs, ql, qh, kl, kh);
                continue;
            }
            if (LF_TILE16 && pass == 1 && S1 > 0 && tl == NTT_TILE_LOG_MAX && rs == nullptr) {
                // contiguous 12-stage pass: 16 words per thread (ckks_ntt_tile16.h), either class or both
                launch_pass16(false, relaxed, nb, st, base, base, g, in, dp, (const i64 *)psi_br, psi_dp, (const i64 *)ql,
                              (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
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
                LF_LAUNCH_CLASS(ntt_fwd_pass, false, relaxed, dim3(per_row * in.n), dim3(NTT_THREADS), 0, st_int, base, g, in,
                                   (const i64 *)psi_br, psi_dp, rs, (const i64 *)ql, (const i64 *)qh, (const i64 *)kl,
                                   (const i64 *)kh);
        }
    }
    if (side) {
        (void)hipEventRecord(side->join, side->stream);
        (void)hipStreamWaitEvent(st_main, side->join, 0);
    }
    return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int lf_intt(int64_t *a, int batch, int rows, int logN, const int64_t *ipsi_br, const double *ipsi_dp,
            const int64_t *q_host, const int64_t *Ninv, int tail, int flags, const int64_t *_2q, const int64_t *ql,
            const int64_t *qh, const int64_t *kl, const int64_t *kh, int device, void *stream) {
    (void)_2q;
    if (batch < 0 || rows < 0 || rows > MAX_LIST_ROWS || logN < 1 || logN > 2 * NTT_TILE_LOG_MAX || tail < 0 || tail > 3)
        return LF_ERR_ARG;
    if ((flags & LF_NTT_RELAXED) && tail < 2) return LF_ERR_ARG;   // relaxed results are canonical residues only
    if (batch == 0 || rows == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    const int tl = logN < NTT_TILE_LOG_MAX ? logN : NTT_TILE_LOG_MAX;
    const int SB = logN - tl;
    const int relaxed = flags & LF_NTT_RELAXED;
    const int plain = (relaxed && (flags & LF_NTT_PLAIN)) ? 1 : 0;
    if (relaxed && !ipsi_dp) return LF_ERR_ARG;
    RowList dp, in;
    classify(rows, q_host, ipsi_dp, dp, in);
    hipStream_t st = (hipStream_t)stream;
    const bool mixed = dp.n && in.n && mixed_enabled();   // both classes in one launch per pass
    SideStream *side = (dp.n && in.n && !mixed) ? side_stream(device) : nullptr;
    hipStream_t st_int = side ? side->stream : st;
    if (side) {
        (void)hipEventRecord(side->fork, st);
        (void)hipStreamWaitEvent(side->stream, side->fork, 0);
    }
    const int chunk = chunk_polys(batch, rows, logN, SB > 0);
    for (int b0 = 0; b0 < batch; b0 += chunk) {
        const int nb = batch - b0 < chunk ? batch - b0 : chunk;
        i64 *base = (i64 *)a + ((i64)b0 * rows << logN);
        const unsigned per_row = (unsigned)nb << (logN - tl);
        for (int pass = 0; pass < (SB > 0 ? 2 : 1); ++pass) {
            const PassGeom g = pass == 0 ? PassGeom{logN, tl, 0, tl, 0, 0, rows, nb, relaxed, SB == 0, plain, regtile_disabled()}
                                         : PassGeom{logN, tl, 1, SB, tl, tl - SB, rows, nb, relaxed, 1, plain};
            const int t = g.last ? tail : TAIL_NONE;
            if (pass == 1 && SB <= 4 && cols_enabled()) {   // trailing stages + chain tail: one register step per column
                if (mixed) {
                    launch_inv_cols_mixed(SB, nb, st, base, g, in, dp, (const i64 *)ipsi_br, ipsi_dp, (const i64 *)Ninv, t,
                                          (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
                    continue;
                }
                if (dp.n)
                    launch_inv_cols<true>(SB, nb, st, base, g, dp, (const i64 *)ipsi_br, ipsi_dp, (const i64 *)Ninv, t,
                                          (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
                if (in.n)
                    launch_inv_cols<false>(SB, nb, st_int, base, g, in, (const i64 *)ipsi_br, ipsi_dp, (const i64 *)Ninv, t,
                                           (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
                continue;
            }
            if (LF_TILE16 && pass == 0 && SB > 0 && tl == NTT_TILE_LOG_MAX) {
                launch_pass16(true, relaxed, nb, st, (const i64 *)base, base, g, in, dp, (const i64 *)ipsi_br, ipsi_dp,
                              (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
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
                LF_LAUNCH_CLASS(ntt_inv_pass_io, false, relaxed, dim3(per_row * in.n), dim3(NTT_THREADS), 0, st_int, (const i64 *)base, base, g, in,
                                   (const i64 *)ipsi_br, ipsi_dp, (const i64 *)Ninv, t, (const i64 *)ql, (const i64 *)qh,
                                   (const i64 *)kl, (const i64 *)kh);
        }
    }
    if (side) {
        (void)hipEventRecord(side->join, side->stream);
        (void)hipStreamWaitEvent(st, side->join, 0);
    }
    return (int)hipGetLastError();
}

}  // extern "C"
