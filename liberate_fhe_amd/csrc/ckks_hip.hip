// ckks_hip.hip — gfx950 (MI355X / CDNA4) kernels and C ABI for RNS-CKKS limb arithmetic.
//
// Design notes (see DESIGN.md for the full account):
//  * Word model = the reference's 62-bit mode: int64 words, Montgomery radix R = 2^62, lazy values
//    in [0, 2q).  The reference builds REDC62 from 31-bit half-words (K.cu:12-59); here the same
//    function is evaluated in closed form with native 64x64->128 products:
//        mm(a,b) = hi62(a*b) + floor(4*s*q / 2^64) + [lo62(a*b) != 0],  s = lo62(a*b) * k mod 2^62
//    which is exactly (a*b + s*q) / 2^62 for all |a|,|b| < 2^62, i.e. bit-identical outputs.
//  * NTT = the reference's radix-2 Cooley-Tukey / Gentleman-Sande DAG (same butterflies, same
//    conditional subtractions) but executed as at most TWO kernels per transform instead of logN:
//    a tile of up to 4096 coefficients lives in LDS, each thread keeps 8 coefficients in registers
//    across three consecutive stages (12 butterflies per LDS round trip), and twiddles come from a
//    compact [limbs][N] table instead of the reference's [limbs][logN][N/2] per-stage table.
//    N > 4096 splits the DAG into a column-strided pass and a contiguous pass.
//  * No MFMA: this is 64-bit integer modular arithmetic.  No CUDA shims; gfx950 only.
#include "../../include/ckks_hip.h"
#include "ckks_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// Elementwise kernels.  grid = (ceil(N / (256*2)), rows); two words (16 B) per lane.
// ------------------------------------------------------------------------------------------------

enum EwOp { EW_MULT, EW_ENTER, EW_REDC, EW_REDUCE, EW_SIGNED, EW_UNSIGNED, EW_TILE, EW_ADD, EW_SUB };

template <int OP>
__global__ void __launch_bounds__(256) ew_kernel(const i64 *a, const i64 *__restrict__ b, i64 *c,
                                                 i64 N, const i64 *__restrict__ v0, const i64 *__restrict__ ql,
                                                 const i64 *__restrict__ qh, const i64 *__restrict__ kl,
                                                 const i64 *__restrict__ kh) {
    const int row = blockIdx.y;
    const i64 j = ((i64)blockIdx.x * 256 + threadIdx.x) * 2;
    if (j >= N) return;
    const i64 off = (i64)row * N + j;
    u64 q = 0, k = 0;
    i64 q2 = 0, sc = 0;
    if (OP == EW_MULT || OP == EW_ENTER || OP == EW_REDC) {
        q = ((u64)qh[row] << 31) | (u64)ql[row];
        k = ((u64)kh[row] << 31) | (u64)kl[row];
    }
    if (OP == EW_ENTER) sc = v0[row];
    if (OP == EW_REDUCE || OP == EW_SIGNED || OP == EW_UNSIGNED || OP == EW_TILE || OP == EW_ADD || OP == EW_SUB) q2 = v0[row];
    const i64 qq = q2 >> 1;

    longlong2 x, y = {0, 0}, r;
    if (OP == EW_TILE)
        x = *reinterpret_cast<const longlong2 *>(a + j);
    else
        x = *reinterpret_cast<const longlong2 *>(a + off);
    if (OP == EW_MULT || OP == EW_ADD || OP == EW_SUB) y = *reinterpret_cast<const longlong2 *>(b + off);

    auto f = [&](i64 xv, i64 yv) -> i64 {
        switch (OP) {
            case EW_MULT: return mm62s(xv, yv, q, k);
            case EW_ENTER: return mm62s(xv, sc, q, k);
            case EW_REDC: return redc62(xv, q, k);
            case EW_REDUCE: return xv < qq ? xv : xv - qq;
            case EW_SIGNED: return xv <= (qq >> 1) ? xv : xv - qq;
            case EW_UNSIGNED: return xv + qq;
            case EW_TILE: return xv + qq;
            case EW_ADD: return csub(xv + yv, q2);
            case EW_SUB: return csub(xv + q2 - yv, q2);
        }
        return 0;
    };
    r.x = f(x.x, y.x);
    r.y = f(x.y, y.y);
    *reinterpret_cast<longlong2 *>(c + off) = r;
}

template <int OP>
int launch_ew(const i64 *a, const i64 *b, i64 *c, int rows, i64 N, const i64 *v0, const i64 *ql, const i64 *qh,
              const i64 *kl, const i64 *kh, int device, void *stream) {
    if (rows < 0 || N < 2 || (N & 1)) return LF_ERR_ARG;
    if (rows == 0) return 0;
    if (device >= 0) {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid((unsigned)((N / 2 + 255) / 256), (unsigned)rows);
    hipLaunchKernelGGL((ew_kernel<OP>), grid, dim3(256), 0, (hipStream_t)stream, a, b, c, N, v0, ql, qh, kl, kh);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// NTT passes
// ------------------------------------------------------------------------------------------------
//
// A pass works on a tile of T = 2^tl coefficients of one limb held in LDS (tl <= 12):
//   contiguous pass: tile = coefficients [base, base + T) of the row;
//   strided pass   : tile = 2^S rows x C columns, C = T >> S, element (r, c) = coefficient
//                    r * D + c0 + c with D = N >> S  (the S largest pair-distances of the DAG).
// In tile-local indexing L (L = r*C + c for the strided form) the stages of a pass are plain
// radix-2 stages at local distances T/2, T/4, ... (forward) or ..., T/4, T/2 (inverse), and the
// twiddle index of a butterfly is a shift of the local index of its upper element:
//   forward stage s : psi_br[(1 << s) + ((base + L) >> (E - s))]       E = logN (contig) | tl (strided, base = 0)
//   inverse stage s : ipsi_br[(N >> (s+1)) + ((base + L) >> (s + 1 - adj))]   adj = 0 | log2(D) - log2(C)
// (reference index tables: ckks_context.py:89-142.)

#define NTT_THREADS 256
#define NTT_TILE_LOG_MAX 12

struct PassGeom {
    int logN;
    int tl;        // log2 of tile size
    int strided;   // 0 contiguous, 1 strided
    int S;         // stages in this pass
    int s0;        // global index of the first stage of this pass
    int logC;      // strided: log2 of columns per tile row
    int rows;      // limbs per polynomial (constants are indexed by blockIdx.y % rows)
};

__device__ __forceinline__ i64 tile_gaddr(const PassGeom &g, int tile, int L) {
    if (!g.strided) return ((i64)tile << g.tl) + L;
    const int r = L >> g.logC, c = L & ((1 << g.logC) - 1);
    return ((i64)r << (g.logN - g.S)) + ((i64)tile << g.logC) + c;
}

// Forward radix-2^K step over local distances (dl << (K-1)), ..., dl at stages s, s+1, ..
template <int K, bool SIGNED>
__device__ __forceinline__ void fwd_step(i64 *sm, int T, int log_dl, int s, int E, i64 base,
                                         const i64 *__restrict__ psi, const RowMod &m) {
    const int items = T >> K;
    for (int w = threadIdx.x; w < items; w += NTT_THREADS) {
        const int p = ((w >> log_dl) << (log_dl + K)) | (w & ((1 << log_dl) - 1));
        i64 x[1 << K];
#pragma unroll
        for (int e = 0; e < (1 << K); ++e) x[e] = sm[p + (e << log_dl)];
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const int du = 1 << (K - 1 - u);
            const int st = s + u;
#pragma unroll
            for (int e = 0; e < (1 << K); ++e) {
                if (e & du) continue;
                const i64 L = base + p + (e << log_dl);
                const i64 S = psi[((i64)1 << st) + (L >> (E - st))];
                const i64 U = x[e];
                const i64 V = SIGNED ? mm62s(S, x[e + du], m.q, m.k) : mm62u((u64)S, (u64)x[e + du], m.q, m.k);
                x[e] = csub(U + V, m.q2);
                x[e + du] = csub(U + m.q2 - V, m.q2);
            }
        }
#pragma unroll
        for (int e = 0; e < (1 << K); ++e) sm[p + (e << log_dl)] = x[e];
    }
}

// Inverse radix-2^K step over local distances dl, 2dl, .. at stages s, s+1, ..
template <int K, bool SIGNED>
__device__ __forceinline__ void inv_step(i64 *sm, int T, int log_dl, int s, int adj, int logN, i64 base,
                                         const i64 *__restrict__ ipsi, const RowMod &m) {
    const int items = T >> K;
    for (int w = threadIdx.x; w < items; w += NTT_THREADS) {
        const int p = ((w >> log_dl) << (log_dl + K)) | (w & ((1 << log_dl) - 1));
        i64 x[1 << K];
#pragma unroll
        for (int e = 0; e < (1 << K); ++e) x[e] = sm[p + (e << log_dl)];
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const int du = 1 << u;
            const int st = s + u;
#pragma unroll
            for (int e = 0; e < (1 << K); ++e) {
                if (e & du) continue;
                const i64 L = base + p + (e << log_dl);
                const i64 S = ipsi[((i64)1 << (logN - st - 1)) + (L >> (st + 1 - adj))];
                const i64 U = x[e], V = x[e + du];
                const i64 O = csub(U + m.q2 - V, m.q2);
                x[e + du] = SIGNED ? mm62s(S, O, m.q, m.k) : mm62u((u64)S, (u64)O, m.q, m.k);
                x[e] = csub(U + V, m.q2);
            }
        }
#pragma unroll
        for (int e = 0; e < (1 << K); ++e) sm[p + (e << log_dl)] = x[e];
    }
}

// tail codes of the inverse chain
#define TAIL_NONE (-1)  // not the last pass: store as is

__global__ void __launch_bounds__(NTT_THREADS) ntt_fwd_pass(i64 *__restrict__ a, PassGeom g, const i64 *__restrict__ psi_br,
                                                            const i64 *__restrict__ Rs, const i64 *__restrict__ ql,
                                                            const i64 *__restrict__ qh, const i64 *__restrict__ kl,
                                                            const i64 *__restrict__ kh) {
    __shared__ i64 sm[1 << NTT_TILE_LOG_MAX];
    const int T = 1 << g.tl;
    const int prow = blockIdx.y;           // row in the [batch*rows] stack
    const int crow = prow % g.rows;        // constant / twiddle row
    const int tile = blockIdx.x;
    const RowMod m = load_mod(ql, qh, kl, kh, crow);
    i64 *row = a + ((i64)prow << g.logN);
    const i64 *psi = psi_br + ((i64)crow << g.logN);

    // load (two consecutive local indices per lane -> 16-byte accesses; C >= 2 always)
    const bool enter = (Rs != nullptr);
    const i64 rs = enter ? Rs[crow] : 0;
    int odd_word = 0;  // any word outside [0, 2q): the rare signed-lazy inputs
    for (int L = threadIdx.x * 2; L < T; L += NTT_THREADS * 2) {
        longlong2 v = *reinterpret_cast<const longlong2 *>(row + tile_gaddr(g, tile, L));
        if (enter) {
            v.x = mm62s(v.x, rs, m.q, m.k);
            v.y = mm62s(v.y, rs, m.q, m.k);
        }
        odd_word |= ((u64)v.x >= (u64)m.q2) | ((u64)v.y >= (u64)m.q2);
        *reinterpret_cast<longlong2 *>(sm + L) = v;
    }
    // With every word in [0, 2q) all later words stay there and the unsigned product is exact;
    // otherwise run the tile with the fully signed REDC, as the reference's scalar code does.
    const bool sgn = __syncthreads_or(odd_word);

    const int E = g.strided ? g.tl : g.logN;
    const i64 base = g.strided ? 0 : ((i64)tile << g.tl);
    int s = g.s0, left = g.S, log_d = g.tl - 1;  // log2 of the current largest local distance
    while (left > 0) {
        if (left >= 3) {
            if (sgn) fwd_step<3, true>(sm, T, log_d - 2, s, E, base, psi, m);
            else fwd_step<3, false>(sm, T, log_d - 2, s, E, base, psi, m);
            s += 3; left -= 3; log_d -= 3;
        } else if (left == 2) {
            if (sgn) fwd_step<2, true>(sm, T, log_d - 1, s, E, base, psi, m);
            else fwd_step<2, false>(sm, T, log_d - 1, s, E, base, psi, m);
            s += 2; left -= 2; log_d -= 2;
        } else {
            if (sgn) fwd_step<1, true>(sm, T, log_d, s, E, base, psi, m);
            else fwd_step<1, false>(sm, T, log_d, s, E, base, psi, m);
            s += 1; left -= 1; log_d -= 1;
        }
        __syncthreads();
    }

    for (int L = threadIdx.x * 2; L < T; L += NTT_THREADS * 2)
        *reinterpret_cast<longlong2 *>(row + tile_gaddr(g, tile, L)) = *reinterpret_cast<const longlong2 *>(sm + L);
}

__global__ void __launch_bounds__(NTT_THREADS) ntt_inv_pass(i64 *__restrict__ a, PassGeom g, const i64 *__restrict__ ipsi_br,
                                                            const i64 *__restrict__ Ninv, int tail,
                                                            const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                            const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    __shared__ i64 sm[1 << NTT_TILE_LOG_MAX];
    const int T = 1 << g.tl;
    const int prow = blockIdx.y;
    const int crow = prow % g.rows;
    const int tile = blockIdx.x;
    const RowMod m = load_mod(ql, qh, kl, kh, crow);
    i64 *row = a + ((i64)prow << g.logN);
    const i64 *ipsi = ipsi_br + ((i64)crow << g.logN);

    int odd_word = 0;
    for (int L = threadIdx.x * 2; L < T; L += NTT_THREADS * 2) {
        const longlong2 v = *reinterpret_cast<const longlong2 *>(row + tile_gaddr(g, tile, L));
        odd_word |= ((u64)v.x >= (u64)m.q2) | ((u64)v.y >= (u64)m.q2);
        *reinterpret_cast<longlong2 *>(sm + L) = v;
    }
    const bool sgn = __syncthreads_or(odd_word);

    // local distances grow: contiguous pass starts at 1; strided pass starts at C.
    const int adj = g.strided ? (g.logN - g.S - g.logC) : 0;
    const i64 base = g.strided ? 0 : ((i64)tile << g.tl);
    int s = g.s0, left = g.S, log_d = g.strided ? g.logC : 0;
    while (left > 0) {
        if (left >= 3) {
            if (sgn) inv_step<3, true>(sm, T, log_d, s, adj, g.logN, base, ipsi, m);
            else inv_step<3, false>(sm, T, log_d, s, adj, g.logN, base, ipsi, m);
            s += 3; left -= 3; log_d += 3;
        } else if (left == 2) {
            if (sgn) inv_step<2, true>(sm, T, log_d, s, adj, g.logN, base, ipsi, m);
            else inv_step<2, false>(sm, T, log_d, s, adj, g.logN, base, ipsi, m);
            s += 2; left -= 2; log_d += 2;
        } else {
            if (sgn) inv_step<1, true>(sm, T, log_d, s, adj, g.logN, base, ipsi, m);
            else inv_step<1, false>(sm, T, log_d, s, adj, g.logN, base, ipsi, m);
            s += 1; left -= 1; log_d += 1;
        }
        __syncthreads();
    }

    const i64 ninv = (tail != TAIL_NONE) ? Ninv[crow] : 0;
    const i64 qq = m.q2 >> 1;
    for (int L = threadIdx.x * 2; L < T; L += NTT_THREADS * 2) {
        longlong2 v = *reinterpret_cast<const longlong2 *>(sm + L);
        if (tail != TAIL_NONE) {
            // K.cu:527-529 (x Ninv), then the chain tails K.cu:754-766 / 817-832 / 886-902
            i64 t[2] = {v.x, v.y};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                i64 z = mm62s(t[e], ninv, m.q, m.k);
                if (tail >= 1) z = redc62(z, m.q, m.k);
                if (tail >= 2) z = z < qq ? z : z - qq;
                if (tail >= 3) z = z <= (qq >> 1) ? z : z - qq;
                t[e] = z;
            }
            v.x = t[0];
            v.y = t[1];
        }
        *reinterpret_cast<longlong2 *>(row + tile_gaddr(g, tile, L)) = v;
    }
}

__global__ void __launch_bounds__(256) galois_kernel(const i64 *__restrict__ a, i64 *__restrict__ dst, int logN, i64 p,
                                                     const i64 *__restrict__ _2q) {
    const int row = blockIdx.y;
    const i64 N = (i64)1 << logN;
    const i64 n = (i64)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const u64 pn = ((u64)p * (u64)n) & (u64)(2 * N - 1);   // p < 2N, n < N: product < 2^35
    i64 v = a[((i64)row << logN) + n];
    if (pn >= (u64)N) v = -v;
    if (_2q) {
        const i64 q = _2q[row] >> 1;
        v += q;
        v = v < q ? v : v - q;
    }
    dst[((i64)row << logN) + (i64)(pn & (u64)(N - 1))] = v;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int lf_abi_version(void) { return 1; }

int lf_mont_mult(const int64_t *a, const int64_t *b, int64_t *c, int rows, int64_t N, const int64_t *ql,
                 const int64_t *qh, const int64_t *kl, const int64_t *kh, int device, void *stream) {
    return launch_ew<EW_MULT>((const i64 *)a, (const i64 *)b, (i64 *)c, rows, N, nullptr, (const i64 *)ql, (const i64 *)qh,
                              (const i64 *)kl, (const i64 *)kh, device, stream);
}

int lf_mont_enter(int64_t *a, const int64_t *Rs, int rows, int64_t N, const int64_t *ql, const int64_t *qh,
                  const int64_t *kl, const int64_t *kh, int device, void *stream) {
    return launch_ew<EW_ENTER>((const i64 *)a, nullptr, (i64 *)a, rows, N, (const i64 *)Rs, (const i64 *)ql, (const i64 *)qh,
                               (const i64 *)kl, (const i64 *)kh, device, stream);
}

int lf_mont_redc(int64_t *a, int rows, int64_t N, const int64_t *ql, const int64_t *qh, const int64_t *kl,
                 const int64_t *kh, int device, void *stream) {
    return launch_ew<EW_REDC>((const i64 *)a, nullptr, (i64 *)a, rows, N, nullptr, (const i64 *)ql, (const i64 *)qh,
                              (const i64 *)kl, (const i64 *)kh, device, stream);
}

int lf_reduce_2q(int64_t *a, int rows, int64_t N, const int64_t *_2q, int device, void *stream) {
    return launch_ew<EW_REDUCE>((const i64 *)a, nullptr, (i64 *)a, rows, N, (const i64 *)_2q, nullptr, nullptr, nullptr, nullptr,
                                device, stream);
}

int lf_make_signed(int64_t *a, int rows, int64_t N, const int64_t *_2q, int device, void *stream) {
    return launch_ew<EW_SIGNED>((const i64 *)a, nullptr, (i64 *)a, rows, N, (const i64 *)_2q, nullptr, nullptr, nullptr, nullptr,
                                device, stream);
}

int lf_make_unsigned(int64_t *a, int rows, int64_t N, const int64_t *_2q, int device, void *stream) {
    return launch_ew<EW_UNSIGNED>((const i64 *)a, nullptr, (i64 *)a, rows, N, (const i64 *)_2q, nullptr, nullptr, nullptr,
                                  nullptr, device, stream);
}

int lf_tile_unsigned(const int64_t *a, int64_t *dst, int rows, int64_t N, const int64_t *_2q, int device, void *stream) {
    return launch_ew<EW_TILE>((const i64 *)a, nullptr, (i64 *)dst, rows, N, (const i64 *)_2q, nullptr, nullptr, nullptr, nullptr,
                              device, stream);
}

int lf_mont_add(const int64_t *a, const int64_t *b, int64_t *c, int rows, int64_t N, const int64_t *_2q, int device,
                void *stream) {
    return launch_ew<EW_ADD>((const i64 *)a, (const i64 *)b, (i64 *)c, rows, N, (const i64 *)_2q, nullptr, nullptr, nullptr,
                             nullptr, device, stream);
}

int lf_mont_sub(const int64_t *a, const int64_t *b, int64_t *c, int rows, int64_t N, const int64_t *_2q, int device,
                void *stream) {
    return launch_ew<EW_SUB>((const i64 *)a, (const i64 *)b, (i64 *)c, rows, N, (const i64 *)_2q, nullptr, nullptr, nullptr,
                             nullptr, device, stream);
}

int lf_ntt(int64_t *a, int batch, int rows, int logN, const int64_t *psi_br, const int64_t *Rs, const int64_t *_2q,
           const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh, int device, void *stream) {
    (void)_2q;
    if (batch < 0 || rows < 0 || logN < 1 || logN > 2 * NTT_TILE_LOG_MAX) return LF_ERR_ARG;
    if (batch == 0 || rows == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    const int tl = logN < NTT_TILE_LOG_MAX ? logN : NTT_TILE_LOG_MAX;
    const int S1 = logN - tl;  // stages of the strided pass
    const unsigned tiles = 1u << (logN - tl);
    dim3 grid(tiles, (unsigned)(batch * rows));
    if (S1 > 0) {
        PassGeom g{logN, tl, 1, S1, 0, tl - S1, rows};
        hipLaunchKernelGGL(ntt_fwd_pass, grid, dim3(NTT_THREADS), 0, (hipStream_t)stream, (i64 *)a, g, (const i64 *)psi_br,
                           (const i64 *)Rs, (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
    }
    PassGeom g{logN, tl, 0, tl, S1, 0, rows};
    hipLaunchKernelGGL(ntt_fwd_pass, grid, dim3(NTT_THREADS), 0, (hipStream_t)stream, (i64 *)a, g, (const i64 *)psi_br,
                       S1 > 0 ? nullptr : (const i64 *)Rs, (const i64 *)ql, (const i64 *)qh, (const i64 *)kl,
                       (const i64 *)kh);
    return (int)hipGetLastError();
}

int lf_intt(int64_t *a, int batch, int rows, int logN, const int64_t *ipsi_br, const int64_t *Ninv, int tail,
            const int64_t *_2q, const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh, int device,
            void *stream) {
    (void)_2q;
    if (batch < 0 || rows < 0 || logN < 1 || logN > 2 * NTT_TILE_LOG_MAX || tail < 0 || tail > 3) return LF_ERR_ARG;
    if (batch == 0 || rows == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    const int tl = logN < NTT_TILE_LOG_MAX ? logN : NTT_TILE_LOG_MAX;
    const int SB = logN - tl;  // stages of the strided (second) pass
    const unsigned tiles = 1u << (logN - tl);
    dim3 grid(tiles, (unsigned)(batch * rows));
    PassGeom ga{logN, tl, 0, tl, 0, 0, rows};
    hipLaunchKernelGGL(ntt_inv_pass, grid, dim3(NTT_THREADS), 0, (hipStream_t)stream, (i64 *)a, ga, (const i64 *)ipsi_br,
                       (const i64 *)Ninv, SB > 0 ? TAIL_NONE : tail, (const i64 *)ql, (const i64 *)qh, (const i64 *)kl,
                       (const i64 *)kh);
    if (SB > 0) {
        PassGeom gb{logN, tl, 1, SB, tl, tl - SB, rows};
        hipLaunchKernelGGL(ntt_inv_pass, grid, dim3(NTT_THREADS), 0, (hipStream_t)stream, (i64 *)a, gb,
                           (const i64 *)ipsi_br, (const i64 *)Ninv, tail, (const i64 *)ql, (const i64 *)qh, (const i64 *)kl,
                           (const i64 *)kh);
    }
    return (int)hipGetLastError();
}

int lf_galois(const int64_t *a, int64_t *dst, int rows, int logN, int64_t p, const int64_t *_2q, int device, void *stream) {
    if (rows < 0 || logN < 1 || logN > 30 || p < 1 || !(p & 1) || p >= ((int64_t)2 << logN) || a == dst) return LF_ERR_ARG;
    if (rows == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    const i64 N = (i64)1 << logN;
    dim3 grid((unsigned)((N + 255) / 256), (unsigned)rows);
    hipLaunchKernelGGL(galois_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const i64 *)a, (i64 *)dst, logN, (i64)p,
                       (const i64 *)_2q);
    return (int)hipGetLastError();
}

}  // extern "C"
