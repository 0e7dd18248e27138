// ckks_fused.hip — engine-level fused kernels: rescale, tensor product, hybrid key switching.
//
// Each kernel replaces a run of `ntt_cuda` calls plus plain torch ops that the reference engine
// issues from Python (src/liberate/fhe/ckks_engine.py: rescale 967-1052, cc_mult 1095-1101,
// pre_extend 654-705, extend 707-743, switcher_later_part 931-934, create_switcher 832-901).
// The per-word arithmetic follows the reference op by op (same REDC, same conditional subtractions,
// same order), so every intermediate word — including the signed-lazy Garner digits that the result
// depends on as INTEGERS — is bit-identical; what changes is that a column of a digit / a ciphertext
// stays in registers across the whole chain instead of round-tripping through HBM per op.
#include "../../include/ckks_hip.h"
#include "ckks_common.h"
#include <stdlib.h>


namespace {

// (a * w) mod q in fp64 for q < 2^41, |a| < 2^52, 0 <= w < q (see ckks_ntt_core.h)
__device__ __forceinline__ double dp_mulmod_q(double a, double w, double q, double qinv) {
    const double hi = a * w;
    const double lo = __builtin_fma(a, w, -hi);
    const double quo = __builtin_rint(hi * qinv);
    const double r = __builtin_fma(-quo, q, hi) + lo;
    return r < 0.0 ? r + q : r;
}

// ---- rescale (ckks_engine.py:1029-1041) ----------------------------------------------------------
// out = reduce_q( REDC(in - row0, q_l^-1 * R mod q_i) + [row0 > q_l/2] )
// Up to LF_BATCH_MAX independent operand sets per launch (blockIdx.z): the two components of a ciphertext, or
// the four polynomials cc_mult rescales, share one launch instead of paying a launch gap each.
struct PtrBatch {
    const i64 *in[LF_BATCH_MAX];
    const i64 *aux[LF_BATCH_MAX];
    i64 *out[LF_BATCH_MAX];
};

__global__ void __launch_bounds__(256) rescale_kernel(PtrBatch pb, i64 N, const i64 *__restrict__ scales,
                                                      i64 round_at, const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                      const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    const i64 *__restrict__ in = pb.in[blockIdx.z];
    const i64 *__restrict__ row0 = pb.aux[blockIdx.z];
    i64 *__restrict__ out = pb.out[blockIdx.z];
    const int r = blockIdx.y;
    const i64 j = ((i64)blockIdx.x * 256 + threadIdx.x) * 2;
    if (j >= N) return;
    const RowMod m = load_mod(ql, qh, kl, kh, r);
    const i64 sc = scales[r];
    const longlong2 x = *reinterpret_cast<const longlong2 *>(in + (i64)r * N + j);
    const longlong2 z = *reinterpret_cast<const longlong2 *>(row0 + j);
    longlong2 o;
    i64 v = mm62s(x.x - z.x, sc, m.q, m.k) + (i64)(z.x > round_at);
    o.x = v < (i64)m.q ? v : v - (i64)m.q;
    v = mm62s(x.y - z.y, sc, m.q, m.k) + (i64)(z.y > round_at);
    o.y = v < (i64)m.q ? v : v - (i64)m.q;
    *reinterpret_cast<longlong2 *>(out + (i64)r * N + j) = o;
}

// ---- tensor product (ckks_engine.py:1095-1101) ---------------------------------------------------
__global__ void __launch_bounds__(256) tensor_kernel(const i64 *__restrict__ x0, const i64 *__restrict__ x1,
                                                     const i64 *__restrict__ y0, const i64 *__restrict__ y1,
                                                     i64 *__restrict__ d0, i64 *__restrict__ d1, i64 *__restrict__ d2, i64 N,
                                                     int plain, const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                     const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    const int r = blockIdx.y;
    const i64 j = ((i64)blockIdx.x * 256 + threadIdx.x) * 2;
    if (j >= N) return;
    const RowMod m = load_mod(ql, qh, kl, kh, r);
    const i64 off = (i64)r * N + j;
    const longlong2 a0 = *reinterpret_cast<const longlong2 *>(x0 + off), a1 = *reinterpret_cast<const longlong2 *>(x1 + off);
    const longlong2 b0 = *reinterpret_cast<const longlong2 *>(y0 + off), b1 = *reinterpret_cast<const longlong2 *>(y1 + off);
    longlong2 o0, o1, o2;
    if (plain && m.q < (1ull << 41)) {
        // plain canonical residues in, plain canonical residues out: one fp64 product per term
        const double q = (double)m.q, qinv = 1.0 / q;
        const double A0[2] = {dp_from_word(a0.x), dp_from_word(a0.y)}, A1[2] = {dp_from_word(a1.x), dp_from_word(a1.y)};
        const double B0[2] = {dp_from_word(b0.x), dp_from_word(b0.y)}, B1[2] = {dp_from_word(b1.x), dp_from_word(b1.y)};
        i64 r0[2], r1[2], r2[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            r0[e] = dp_to_word(dp_mulmod_q(A0[e], B0[e], q, qinv));
            double t = dp_mulmod_q(A0[e], B1[e], q, qinv) + dp_mulmod_q(A1[e], B0[e], q, qinv);
            r1[e] = dp_to_word(t >= q ? t - q : t);
            r2[e] = dp_to_word(dp_mulmod_q(A1[e], B1[e], q, qinv));
        }
        o0.x = r0[0]; o0.y = r0[1]; o1.x = r1[0]; o1.y = r1[1]; o2.x = r2[0]; o2.y = r2[1];
        *reinterpret_cast<longlong2 *>(d0 + off) = o0;
        *reinterpret_cast<longlong2 *>(d1 + off) = o1;
        *reinterpret_cast<longlong2 *>(d2 + off) = o2;
        return;
    }
    o0.x = mm62s(a0.x, b0.x, m.q, m.k);
    o0.y = mm62s(a0.y, b0.y, m.q, m.k);
    o1.x = csub(mm62s(a0.x, b1.x, m.q, m.k) + mm62s(a1.x, b0.x, m.q, m.k), m.q2);
    o1.y = csub(mm62s(a0.y, b1.y, m.q, m.k) + mm62s(a1.y, b0.y, m.q, m.k), m.q2);
    o2.x = mm62s(a1.x, b1.x, m.q, m.k);
    o2.y = mm62s(a1.y, b1.y, m.q, m.k);
    *reinterpret_cast<longlong2 *>(d0 + off) = o0;
    *reinterpret_cast<longlong2 *>(d1 + off) = o1;
    *reinterpret_cast<longlong2 *>(d2 + off) = o2;
}

// Coefficient j of a(X^p) read straight from a (gather form of the Galois permutation, encdec.py:224-270):
// with n = p^-1 * j mod 2N it is a[n] for n < N and -a[n - N] otherwise; `q2` != 0 additionally makes it
// canonical the way rotate_single does (make_unsigned + reduce_2q, eng.py:1198-1200).  pinv = 0: plain read.
static __device__ __forceinline__ i64 galois_read(const i64 *row, i64 j, i64 N, i64 pinv, i64 q2) {
    if (pinv == 0) return row[j];
    const u64 n = ((u64)pinv * (u64)j) & (u64)(2 * N - 1);
    i64 v = row[n & (u64)(N - 1)];
    if (n >= (u64)N) v = -v;
    if (q2) {
        const i64 q = q2 >> 1;
        v += q;
        v = v < q ? v : v - q;
    }
    return v;
}

// ---- key-switch step 1: mixed-radix digits of each key-switch part (pre_extend, 654-705) ---------
// desc[p] = {row_start, alpha, y_off, l_off}; Y_scalar[i] = tab[y_off + i];
// L_scalar[i][j-(i+2)] = tab[l_off + running index in (i, j) order].
// blockIdx.z: one of up to LF_BATCH_MAX polynomials (pb.in[z] -> pb.out[z]) sharing the tables
__global__ void __launch_bounds__(256) ks_digits_kernel(PtrBatch pb,
                                                        const i64 *__restrict__ desc, const i64 *__restrict__ tab, i64 N,
                                                        i64 gal_pinv, const i64 *__restrict__ gal_2q,
                                                        const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                        const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    const i64 *__restrict__ a = pb.in[blockIdx.z];
    i64 *__restrict__ state = pb.out[blockIdx.z];
    const int p = blockIdx.y;
    const i64 j = (i64)blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const int row_start = (int)desc[p * 4 + 0], alpha = (int)desc[p * 4 + 1];
    const i64 *Y = tab + desc[p * 4 + 2];
    const i64 *Ls = tab + desc[p * 4 + 3];
    i64 x[KS_MAX_ALPHA], st[KS_MAX_ALPHA];
#pragma unroll
    for (int i = 0; i < KS_MAX_ALPHA; ++i)
        if (i < alpha)
            x[i] = galois_read(a + (i64)(row_start + i) * N, j, N, gal_pinv, gal_2q ? gal_2q[row_start + i] : 0);
#pragma unroll
    for (int i = 0; i < KS_MAX_ALPHA; ++i) st[i] = x[0];
    int lc = 0;
#pragma unroll
    for (int i = 0; i < KS_MAX_ALPHA - 1; ++i) {
        if (i + 1 < alpha) {
            const RowMod m = load_mod(ql, qh, kl, kh, row_start + i + 1);
            const i64 y = mm62s(x[i + 1] - st[i + 1], Y[i], m.q, m.k);
            st[i + 1] = y;
#pragma unroll
            for (int jj = i + 2; jj < KS_MAX_ALPHA; ++jj) {
                if (jj < alpha) {
                    const RowMod mj = load_mod(ql, qh, kl, kh, row_start + jj);
                    st[jj] += mm62s(y, Ls[lc], mj.q, mj.k);
                    ++lc;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < KS_MAX_ALPHA; ++i)
        if (i < alpha) state[(i64)(row_start + i) * N + j] = st[i];
}

// ---- key-switch step 2: extend every digit to every target row (extend, 707-743) -----------------
// desc[p] = {row_start, alpha, e_off};  E[e_off + i*rows + r]: i = 0 -> R^2 mod q_r, i >= 1 -> L_{i-1} R^2 mod q_r
__global__ void __launch_bounds__(256) ks_extend_kernel(const i64 *__restrict__ state, i64 *__restrict__ ext, int rows,
                                                        i64 N, const i64 *__restrict__ desc, const i64 *__restrict__ E,
                                                        const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                        const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    const int r = blockIdx.y, p = blockIdx.z;
    const i64 j = ((i64)blockIdx.x * 256 + threadIdx.x) * 2;
    if (j >= N) return;
    const int row_start = (int)desc[p * 3 + 0], alpha = (int)desc[p * 3 + 1] & 0xff;   // bit 8: lf_ks_core's wide flag
    const i64 *e = E + desc[p * 3 + 2] + r;
    const RowMod m = load_mod(ql, qh, kl, kh, r);
    longlong2 acc;
    {
        const longlong2 y = *reinterpret_cast<const longlong2 *>(state + (i64)row_start * N + j);
        acc.x = mm62s(y.x, e[0], m.q, m.k);
        acc.y = mm62s(y.y, e[0], m.q, m.k);
    }
    for (int i = 1; i < alpha; ++i) {
        const longlong2 y = *reinterpret_cast<const longlong2 *>(state + (i64)(row_start + i) * N + j);
        const i64 c = e[(i64)i * rows];
        acc.x = csub(acc.x + mm62s(y.x, c, m.q, m.k), m.q2);
        acc.y = csub(acc.y + mm62s(y.y, c, m.q, m.k), m.q2);
    }
    *reinterpret_cast<longlong2 *>(ext + ((i64)p * rows + r) * N + j) = acc;
}

// ---- key-switch step 3: inner product with the key, summed over digits (931-934, 832-840) --------
__global__ void __launch_bounds__(256) ks_inner_kernel(const i64 *__restrict__ ext, const i64 *__restrict__ ksk,
                                                       i64 part_stride, i64 comp_stride, i64 row_off, i64 *__restrict__ s0,
                                                       i64 *__restrict__ s1, int nparts, int rows, i64 N,
                                                       const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                       const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    const int r = blockIdx.y;
    const i64 j = ((i64)blockIdx.x * 256 + threadIdx.x) * 2;
    if (j >= N) return;
    const RowMod m = load_mod(ql, qh, kl, kh, r);
    longlong2 a0 = {0, 0}, a1 = {0, 0};
    for (int p = 0; p < nparts; ++p) {
        const longlong2 e = *reinterpret_cast<const longlong2 *>(ext + ((i64)p * rows + r) * N + j);
        const i64 *kp = ksk + (i64)p * part_stride + (row_off + r) * N + j;
        const longlong2 k0 = *reinterpret_cast<const longlong2 *>(kp);
        const longlong2 k1 = *reinterpret_cast<const longlong2 *>(kp + comp_stride);
        const i64 p0x = mm62s(e.x, k0.x, m.q, m.k), p0y = mm62s(e.y, k0.y, m.q, m.k);
        const i64 p1x = mm62s(e.x, k1.x, m.q, m.k), p1y = mm62s(e.y, k1.y, m.q, m.k);
        if (p == 0) {
            a0.x = p0x; a0.y = p0y; a1.x = p1x; a1.y = p1y;
        } else {
            a0.x = csub(a0.x + p0x, m.q2); a0.y = csub(a0.y + p0y, m.q2);
            a1.x = csub(a1.x + p1x, m.q2); a1.y = csub(a1.y + p1y, m.q2);
        }
    }
    *reinterpret_cast<longlong2 *>(s0 + (i64)r * N + j) = a0;
    *reinterpret_cast<longlong2 *>(s1 + (i64)r * N + j) = a1;
}

// ---- key-switch step 5: exact division by P = prod(special primes) (create_switcher, 850-901) ----
// s: [ell + K][N] coefficient domain, canonical (after intt_exit_reduce).  Special primes are
// eliminated last-first; PiR[P_ind][row] = P_j^-1 * R mod q_row.  Optional `addend` (relinearize's
// d0/d1 or the rotated c0): out = reduce_q(result + addend)  (ckks_engine.py:1135-1140, 952-953).
#define MD_ROWS 8
__global__ void __launch_bounds__(256) ks_moddown_kernel(PtrBatch pb, int ell, int K, i64 N, i64 gal_pinv,
                                                         const i64 *__restrict__ gal_2q,
                                                         const i64 *__restrict__ PiR, const double *__restrict__ PiP,
                                                         const i64 *__restrict__ Rs,
                                                         const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                         const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    const i64 *__restrict__ s = pb.in[blockIdx.z];
    const i64 *__restrict__ addend = pb.aux[blockIdx.z];
    i64 *__restrict__ out = pb.out[blockIdx.z];
    const i64 j = (i64)blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const int rows = ell + K;
    i64 sp[KS_MAX_K], pv[KS_MAX_K];
#pragma unroll
    for (int t = 0; t < KS_MAX_K; ++t)
        if (t < K) sp[t] = s[(i64)(ell + t) * N + j];
    // special rows among themselves (plain form)
#pragma unroll
    for (int pi = 0; pi < KS_MAX_K; ++pi) {
        if (pi < K) {
            const int t = K - 1 - pi;
            i64 P = 0;
#pragma unroll
            for (int u = 0; u < KS_MAX_K; ++u)
                if (u == t) P = sp[u];
            pv[pi] = P;
#pragma unroll
            for (int u = 0; u < KS_MAX_K; ++u) {
                if (u < t) {
                    const RowMod m = load_mod(ql, qh, kl, kh, ell + u);
                    i64 d = csub(sp[u] + m.q2 - P, m.q2);
                    d = mm62s(d, PiR[(i64)pi * rows + ell + u], m.q, m.k);
                    sp[u] = d < (i64)m.q ? d : d - (i64)m.q;
                }
            }
        }
    }
    const int r0 = blockIdx.y * MD_ROWS;
    for (int r = r0; r < r0 + MD_ROWS && r < ell; ++r) {
        const RowMod m = load_mod(ql, qh, kl, kh, r);
        i64 d;
        if (PiP != nullptr && m.q < (1ull << 41)) {
            // fp64 class: the chain ((s - p_0) / P_0 - p_1) / P_1 ... in the plain domain, one modular product
            // per special prime; the 60-bit pivots enter fp64 as 31-bit halves.  Same canonical result.
            const double q = (double)m.q, qinv = 1.0 / q;
            double x = (double)s[(i64)r * N + j];                    // canonical < q
            const double two31 = 2147483648.0;
#pragma unroll
            for (int pi = 0; pi < KS_MAX_K; ++pi) {
                if (pi < K) {
                    const double ph = (double)(pv[pi] >> 31), pl = (double)(pv[pi] & 0x7fffffffll);
                    // pivot mod q (< 3q after the two terms), then (x - pivot) * P^-1 mod q
                    double pm = dp_mulmod_q(ph, two31, q, qinv) + pl;
                    pm = pm >= q ? pm - q : pm;
                    pm = pm >= q ? pm - q : pm;
                    x = dp_mulmod_q(x - pm + q, PiP[(i64)pi * rows + r], q, qinv);
                }
            }
            d = (i64)x;
        } else {
            const i64 rs = Rs[r];
            d = mm62s(s[(i64)r * N + j], rs, m.q, m.k);
#pragma unroll
            for (int pi = 0; pi < KS_MAX_K; ++pi) {
                if (pi < K) {
                    const i64 Q = mm62s(pv[pi], rs, m.q, m.k);
                    d = csub(d + m.q2 - Q, m.q2);
                    d = mm62s(d, PiR[(i64)pi * rows + r], m.q, m.k);
                    d = d < (i64)m.q ? d : d - (i64)m.q;
                }
            }
            d = redc62(d, m.q, m.k);
            d = d < (i64)m.q ? d : d - (i64)m.q;
        }
        if (addend) {
            d += galois_read(addend + (i64)r * N, j, N, gal_pinv, gal_2q ? gal_2q[r] : 0);
            d = d < (i64)m.q ? d : d - (i64)m.q;
        }
        out[(i64)r * N + j] = d;
    }
}

// ---- mod-down with a workspace: the special-prime chain is evaluated ONCE per coefficient -----------------
// ks_moddown_kernel recomputes the chain among the K special rows in every 8-row chunk (5 times at gold: half
// of its instructions).  Here a first small launch leaves, in a caller-provided workspace,
//   piv[c][j][.]  the pivots p_0 .. p_{K-1} of every coefficient (mixed-radix digits of the special part:
//                 X = p_0 + P_0 (p_1 + P_1 (p_2 + ..))), and
//   per ordinary fp64-class row r the plain constants
//                 B_j[r] = prod_{i<j} P_i mod q_r,  A_j[r] = 2^31 B_j[r] mod q_r,  Pinv[r] = prod_j P_j^-1 mod q_r;
// the second launch then evaluates, two coefficients per thread,
//   fp64 class : out = (s_r - sum_j (ph_j A_j + pl_j B_j)) * Pinv  — 2K balanced products and one canonical product
//                per word instead of the chained form's 2K canonical ones; the canonical residue is the same number;
//   integer    : the reference's chain with REDC62, word for word (ckks_engine.py:850-901).
// (A single-launch form — wave 0 of a 64-coefficient block runs the chain, one barrier, four waves share the rows
// — was measured slower than the chunked kernel, 63 vs 53 us at gold: one word per lane and the chain's latency
// in front of every block.)
#define MD3_ROWS 4

// the (2K + 1) x ell plain constants of the fp64 rows (one block of 256 threads): B_j, A_j = 2^31 B_j, Pinv — level constants
__device__ __forceinline__ void md_row_constants(double *cst, int ell, int K, const double *__restrict__ PiP,
                                                 const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                 const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    const int rows = ell + K;
    const double two31 = 2147483648.0;
    for (int r = threadIdx.x; r < ell; r += 256) {
        const RowMod m = load_mod(ql, qh, kl, kh, r);
        const bool dp = m.q < (1ull << 41);
        const double q = (double)m.q, qinv = 1.0 / q;
        double B = 1.0, pinv = 1.0;
        for (int pi = 0; pi < K; ++pi) {
            cst[(2 * pi) * ell + r] = dp ? dp_mulmod_q(B, two31, q, qinv) : 0.0;   // A_pi
            cst[(2 * pi + 1) * ell + r] = dp ? B : 0.0;                             // B_pi
            if (dp) {
                // P_pi mod q_r from its 31-bit halves, then the running product
                const int t = ell + K - 1 - pi;
                double Pm = dp_mulmod_q((double)qh[t], two31, q, qinv) + (double)ql[t];
                Pm = Pm >= q ? Pm - q : Pm;
                Pm = Pm >= q ? Pm - q : Pm;
                B = dp_mulmod_q(B, Pm, q, qinv);
                pinv = dp_mulmod_q(pinv, PiP[(i64)pi * rows + r], q, qinv);
            }
        }
        cst[(2 * K) * ell + r] = dp ? pinv : 0.0;
    }
}

__global__ void __launch_bounds__(256) ks_md_consts_kernel(double *cst, int ell, int K, const double *__restrict__ PiP,
                                                           const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                           const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    md_row_constants(cst, ell, K, PiP, ql, qh, kl, kh);
}

// workspace layout (words): [count][K][N] pivots, then (2K + 1) x ell doubles
__global__ void __launch_bounds__(256) ks_pivots_kernel(PtrBatch pb, int count, int ell, int K, i64 N, i64 *__restrict__ ws,
                                                        const i64 *__restrict__ PiR, const double *__restrict__ PiP,
                                                        const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                        const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    const int rows = ell + K;
    if ((int)blockIdx.y == count) {
        // constants of the ordinary rows (one block)
        if (blockIdx.x == 0 && PiP != nullptr) md_row_constants(reinterpret_cast<double *>(ws + (i64)count * K * N), ell, K, PiP, ql, qh, kl, kh);
        return;
    }
    const i64 *__restrict__ s = pb.in[blockIdx.y];
    const i64 j = (i64)blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    i64 sp[KS_MAX_K];
#pragma unroll
    for (int t = 0; t < KS_MAX_K; ++t)
        if (t < K) sp[t] = s[(i64)(ell + t) * N + j];
    // special rows among themselves (plain form): the reference's elimination, last prime first
#pragma unroll
    for (int pi = 0; pi < KS_MAX_K; ++pi) {
        if (pi < K) {
            const int t = K - 1 - pi;
            i64 P = 0;
#pragma unroll
            for (int u = 0; u < KS_MAX_K; ++u)
                if (u == t) P = sp[u];
            ws[((i64)blockIdx.y * K + pi) * N + j] = P;
#pragma unroll
            for (int u = 0; u < KS_MAX_K; ++u) {
                if (u < t) {
                    const RowMod m = load_mod(ql, qh, kl, kh, ell + u);
                    i64 d = csub(sp[u] + m.q2 - P, m.q2);
                    d = mm62s(d, PiR[(i64)pi * rows + ell + u], m.q, m.k);
                    sp[u] = d < (i64)m.q ? d : d - (i64)m.q;
                }
            }
        }
    }
}

// KK = K as a compile-time constant: the pivot halves live in 4 K doubles per thread, nothing is reserved for
// the K the call does not have (8 slots cost 142 VGPRs = 3 waves per SIMD)
// INLINE: the pivots are not read from the workspace but eliminated here, from the special rows of s, by every block for its
// own two coefficients per thread (KK (KK - 1) / 2 REDC products per coefficient, redone by each of the ceil(ell / 4) row
// chunks): for KK <= 2 — one product — cheaper than the pivots launch it replaces (silver: 4.9 us + a launch gap of a
// 150 us op); the constants behind the pivots in the workspace must have been written once (lf_ks_moddown_consts).
template <int KK, bool INLINE>
__global__ void __launch_bounds__(256) ks_moddown_ws_kernel(PtrBatch pb, int count, int ell, i64 N, i64 gal_pinv,
                                                            const i64 *__restrict__ gal_2q, const i64 *__restrict__ ws,
                                                            const i64 *__restrict__ PiR, const i64 *__restrict__ Rs,
                                                            int use_dp, const i64 *__restrict__ ql,
                                                            const i64 *__restrict__ qh, const i64 *__restrict__ kl,
                                                            const i64 *__restrict__ kh) {
    const i64 *__restrict__ s = pb.in[blockIdx.z];
    const i64 *__restrict__ addend = pb.aux[blockIdx.z];
    i64 *__restrict__ out = pb.out[blockIdx.z];
    const i64 j = ((i64)blockIdx.x * 256 + threadIdx.x) * 2;
    if (j >= N) return;
    const int rows = ell + KK;
    const double *__restrict__ cst = reinterpret_cast<const double *>(ws + (i64)count * KK * N);
    const i64 *__restrict__ piv = ws + (i64)blockIdx.z * KK * N + j;
    double ph[KK][2], pl[KK][2];
    longlong2 pvr[INLINE ? KK : 1];   // INLINE: the pivots themselves, for the integer-class rows
    if (INLINE) {
        // the reference's elimination among the special rows, last prime first (ks_pivots_kernel, two coefficients)
        longlong2 sp[KK];
#pragma unroll
        for (int t = 0; t < KK; ++t) sp[t] = *reinterpret_cast<const longlong2 *>(s + (i64)(ell + t) * N + j);
#pragma unroll
        for (int pi = 0; pi < KK; ++pi) {
            const int t = KK - 1 - pi;
            const longlong2 P = sp[t];
            pvr[pi] = P;
#pragma unroll
            for (int u = 0; u < KK; ++u) {
                if (u < t) {
                    const RowMod m = load_mod(ql, qh, kl, kh, ell + u);
                    const i64 pir = PiR[(i64)pi * rows + ell + u];
                    i64 d = mm62s(csub(sp[u].x + m.q2 - P.x, m.q2), pir, m.q, m.k);
                    sp[u].x = d < (i64)m.q ? d : d - (i64)m.q;
                    d = mm62s(csub(sp[u].y + m.q2 - P.y, m.q2), pir, m.q, m.k);
                    sp[u].y = d < (i64)m.q ? d : d - (i64)m.q;
                }
            }
        }
    }
#pragma unroll
    for (int pi = 0; pi < KK; ++pi) {
        const longlong2 pv = INLINE ? pvr[INLINE ? pi : 0] : *reinterpret_cast<const longlong2 *>(piv + (i64)pi * N);
        ph[pi][0] = (double)(unsigned)(pv.x >> 31); pl[pi][0] = (double)(unsigned)(pv.x & 0x7fffffffll);
        ph[pi][1] = (double)(unsigned)(pv.y >> 31); pl[pi][1] = (double)(unsigned)(pv.y & 0x7fffffffll);
    }
    const int r0 = blockIdx.y * MD3_ROWS;
    const int r1 = r0 + MD3_ROWS < ell ? r0 + MD3_ROWS : ell;
    longlong2 sv = *reinterpret_cast<const longlong2 *>(s + (i64)r0 * N + j);
    for (int r = r0; r < r1; ++r) {
        const int rn = r + 1 < r1 ? r + 1 : r;
        const longlong2 sv_n = *reinterpret_cast<const longlong2 *>(s + (i64)rn * N + j);   // requested before this row's arithmetic
        const RowMod m = load_mod(ql, qh, kl, kh, r);
        i64 d[2];
        if (use_dp && m.q < (1ull << 41)) {
            const double q = (double)m.q, qinv = 1.0 / q;
            double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
            for (int pi = 0; pi < KK; ++pi) {
                const double wa = cst[(2 * pi) * ell + r], wb = cst[(2 * pi + 1) * ell + r];
                // balanced products (no sign fix): |.| <= q / 2 each
                double hi = ph[pi][0] * wa, lo = __builtin_fma(ph[pi][0], wa, -hi);
                acc0 += __builtin_fma(-__builtin_rint(hi * qinv), q, hi) + lo;
                hi = pl[pi][0] * wb; lo = __builtin_fma(pl[pi][0], wb, -hi);
                acc0 += __builtin_fma(-__builtin_rint(hi * qinv), q, hi) + lo;
                hi = ph[pi][1] * wa; lo = __builtin_fma(ph[pi][1], wa, -hi);
                acc1 += __builtin_fma(-__builtin_rint(hi * qinv), q, hi) + lo;
                hi = pl[pi][1] * wb; lo = __builtin_fma(pl[pi][1], wb, -hi);
                acc1 += __builtin_fma(-__builtin_rint(hi * qinv), q, hi) + lo;
            }
            const double pinv = cst[(2 * KK) * ell + r];
            // s canonical (< 2^41): exact int <-> double through the 2^52 trick (ckks_common.h); |s - acc| <= (K + 1) q
            d[0] = dp_to_word(dp_mulmod_q(dp_from_word(sv.x) - acc0, pinv, q, qinv));
            d[1] = dp_to_word(dp_mulmod_q(dp_from_word(sv.y) - acc1, pinv, q, qinv));
        } else {
            const i64 rs = Rs[r];
            d[0] = mm62s(sv.x, rs, m.q, m.k);
            d[1] = mm62s(sv.y, rs, m.q, m.k);
            for (int pi = 0; pi < KK; ++pi) {
                const longlong2 pv = INLINE ? pvr[INLINE ? pi : 0] : *reinterpret_cast<const longlong2 *>(piv + (i64)pi * N);   // rare rows: re-read
                const i64 pir = PiR[(i64)pi * rows + r];
                const i64 Q0 = mm62s(pv.x, rs, m.q, m.k), Q1 = mm62s(pv.y, rs, m.q, m.k);
                d[0] = mm62s(csub(d[0] + m.q2 - Q0, m.q2), pir, m.q, m.k);
                d[1] = mm62s(csub(d[1] + m.q2 - Q1, m.q2), pir, m.q, m.k);
                d[0] = d[0] < (i64)m.q ? d[0] : d[0] - (i64)m.q;
                d[1] = d[1] < (i64)m.q ? d[1] : d[1] - (i64)m.q;
            }
            d[0] = redc62(d[0], m.q, m.k);
            d[1] = redc62(d[1], m.q, m.k);
            d[0] = d[0] < (i64)m.q ? d[0] : d[0] - (i64)m.q;
            d[1] = d[1] < (i64)m.q ? d[1] : d[1] - (i64)m.q;
        }
        if (addend) {
            const i64 g2 = gal_2q ? gal_2q[r] : 0;
            d[0] += galois_read(addend + (i64)r * N, j, N, gal_pinv, g2);
            d[1] += galois_read(addend + (i64)r * N, j + 1, N, gal_pinv, g2);
            d[0] = d[0] < (i64)m.q ? d[0] : d[0] - (i64)m.q;
            d[1] = d[1] < (i64)m.q ? d[1] : d[1] - (i64)m.q;
        }
        longlong2 o;
        o.x = d[0];
        o.y = d[1];
        *reinterpret_cast<longlong2 *>(out + (i64)r * N + j) = o;
        sv = sv_n;
    }
}

}  // namespace

extern "C" {

int64_t lf_ks_moddown_ws_words(int count, int ell, int K, int64_t N) {
    return (int64_t)count * K * N + (int64_t)(2 * K + 1) * ell;
}

int lf_rescale_batch(const int64_t *const *in, const int64_t *const *row0, int64_t *const *out, int count, int rows,
                     int64_t N, const int64_t *scales, int64_t round_at, const int64_t *ql, const int64_t *qh,
                     const int64_t *kl, const int64_t *kh, int device, void *stream) {
    if (count < 0 || count > LF_BATCH_MAX || rows < 0 || N < 2 || (N & 1)) return LF_ERR_ARG;
    if (count == 0 || rows == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    PtrBatch pb;
    for (int i = 0; i < count; ++i) pb.in[i] = (const i64 *)in[i], pb.aux[i] = (const i64 *)row0[i], pb.out[i] = (i64 *)out[i];
    dim3 grid((unsigned)((N / 2 + 255) / 256), (unsigned)rows, (unsigned)count);
    hipLaunchKernelGGL(rescale_kernel, grid, dim3(256), 0, (hipStream_t)stream, pb, (i64)N, (const i64 *)scales, (i64)round_at,
                       (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
    return (int)hipGetLastError();
}

int lf_rescale(const int64_t *in, const int64_t *row0, int64_t *out, int rows, int64_t N, const int64_t *scales,
               int64_t round_at, const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh, int device,
               void *stream) {
    return lf_rescale_batch(&in, &row0, &out, 1, rows, N, scales, round_at, ql, qh, kl, kh, device, stream);
}

int lf_tensor(const int64_t *x0, const int64_t *x1, const int64_t *y0, const int64_t *y1, int64_t *d0, int64_t *d1,
              int64_t *d2, int rows, int64_t N, int plain, const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh,
              int device, void *stream) {
    if (rows < 0 || N < 2 || (N & 1)) return LF_ERR_ARG;
    if (rows == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    dim3 grid((unsigned)((N / 2 + 255) / 256), (unsigned)rows);
    hipLaunchKernelGGL(tensor_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const i64 *)x0, (const i64 *)x1, (const i64 *)y0,
                       (const i64 *)y1, (i64 *)d0, (i64 *)d1, (i64 *)d2, (i64)N, plain, (const i64 *)ql, (const i64 *)qh,
                       (const i64 *)kl, (const i64 *)kh);
    return (int)hipGetLastError();
}

int lf_ks_digits_batch(const int64_t *const *a, int64_t *const *state, int count, int nparts, const int64_t *desc,
                       const int64_t *tab, int64_t N, int64_t gal_pinv, const int64_t *gal_2q, const int64_t *ql,
                       const int64_t *qh, const int64_t *kl, const int64_t *kh, int device, void *stream) {
    if (count < 0 || count > LF_BATCH_MAX || nparts < 0 || N < 1 || gal_pinv < 0 || gal_pinv >= 2 * N ||
        (gal_pinv && (!(gal_pinv & 1) || (N & (N - 1)))))
        return LF_ERR_ARG;
    if (nparts == 0 || count == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    PtrBatch pb;
    for (int i = 0; i < count; ++i) pb.in[i] = (const i64 *)a[i], pb.aux[i] = nullptr, pb.out[i] = (i64 *)state[i];
    dim3 grid((unsigned)((N + 255) / 256), (unsigned)nparts, (unsigned)count);
    hipLaunchKernelGGL(ks_digits_kernel, grid, dim3(256), 0, (hipStream_t)stream, pb, (const i64 *)desc, (const i64 *)tab,
                       (i64)N, (i64)gal_pinv, (const i64 *)gal_2q, (const i64 *)ql, (const i64 *)qh, (const i64 *)kl,
                       (const i64 *)kh);
    return (int)hipGetLastError();
}

int lf_ks_digits_galois(const int64_t *a, int64_t *state, int nparts, const int64_t *desc, const int64_t *tab, int64_t N,
                        int64_t gal_pinv, const int64_t *gal_2q, const int64_t *ql, const int64_t *qh, const int64_t *kl,
                        const int64_t *kh, int device, void *stream) {
    return lf_ks_digits_batch(&a, &state, 1, nparts, desc, tab, N, gal_pinv, gal_2q, ql, qh, kl, kh, device, stream);
}

int lf_ks_digits(const int64_t *a, int64_t *state, int nparts, const int64_t *desc, const int64_t *tab, int64_t N,
                 const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh, int device, void *stream) {
    return lf_ks_digits_batch(&a, &state, 1, nparts, desc, tab, N, 0, nullptr, ql, qh, kl, kh, device, stream);
}

int lf_ks_extend(const int64_t *state, int64_t *ext, int nparts, int rows, int64_t N, const int64_t *desc, const int64_t *E,
                 const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh, int device, void *stream) {
    if (nparts < 0 || rows < 0 || N < 2 || (N & 1)) return LF_ERR_ARG;
    if (nparts == 0 || rows == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    dim3 grid((unsigned)((N / 2 + 255) / 256), (unsigned)rows, (unsigned)nparts);
    hipLaunchKernelGGL(ks_extend_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const i64 *)state, (i64 *)ext, rows, (i64)N,
                       (const i64 *)desc, (const i64 *)E, (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
    return (int)hipGetLastError();
}

int lf_ks_inner(const int64_t *ext, const int64_t *ksk, int64_t part_stride, int64_t comp_stride, int64_t row_off,
                int64_t *s0, int64_t *s1, int nparts, int rows, int64_t N, const int64_t *ql, const int64_t *qh,
                const int64_t *kl, const int64_t *kh, int device, void *stream) {
    if (nparts < 1 || rows < 0 || N < 2 || (N & 1)) return LF_ERR_ARG;
    if (rows == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    dim3 grid((unsigned)((N / 2 + 255) / 256), (unsigned)rows);
    hipLaunchKernelGGL(ks_inner_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const i64 *)ext, (const i64 *)ksk,
                       (i64)part_stride, (i64)comp_stride, (i64)row_off, (i64 *)s0, (i64 *)s1, nparts, rows, (i64)N,
                       (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
    return (int)hipGetLastError();
}

int lf_ks_moddown_batch(const int64_t *const *s, int64_t *const *out, const int64_t *const *addend, int count, int ell, int K,
                        int64_t N, const int64_t *PiR, const double *PiP, const int64_t *Rs, int64_t gal_pinv,
                        const int64_t *gal_2q, const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh,
                        int device, void *stream) {
    if (count < 0 || count > LF_BATCH_MAX || ell < 0 || K < 1 || K > KS_MAX_K || N < 1) return LF_ERR_ARG;
    if (gal_pinv < 0 || gal_pinv >= 2 * N || (gal_pinv && (!(gal_pinv & 1) || (N & (N - 1))))) return LF_ERR_ARG;
    if (count == 0 || ell == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    PtrBatch pb;
    for (int i = 0; i < count; ++i)
        pb.in[i] = (const i64 *)s[i], pb.aux[i] = addend ? (const i64 *)addend[i] : nullptr, pb.out[i] = (i64 *)out[i];
    dim3 grid((unsigned)((N + 255) / 256), (unsigned)((ell + MD_ROWS - 1) / MD_ROWS), (unsigned)count);
    hipLaunchKernelGGL(ks_moddown_kernel, grid, dim3(256), 0, (hipStream_t)stream, pb, ell, K, (i64)N, (i64)gal_pinv,
                       (const i64 *)gal_2q, (const i64 *)PiR, PiP,
                       (const i64 *)Rs, (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
    return (int)hipGetLastError();
}

static int moddown_ws_launch(bool one, const int64_t *const *s, int64_t *const *out, const int64_t *const *addend, int count, int ell,
                            int K, int64_t N, int64_t *ws, int64_t ws_words, const int64_t *PiR, const double *PiP, const int64_t *Rs,
                            int64_t gal_pinv, const int64_t *gal_2q, const int64_t *ql, const int64_t *qh, const int64_t *kl,
                            const int64_t *kh, int device, void *stream) {
    if (count < 0 || count > LF_BATCH_MAX || ell < 0 || K < 1 || K > KS_MAX_K || N < 2 || (N & 1)) return LF_ERR_ARG;
    if (gal_pinv < 0 || gal_pinv >= 2 * N || (gal_pinv && (!(gal_pinv & 1) || (N & (N - 1))))) return LF_ERR_ARG;
    if (!ws || ws_words < lf_ks_moddown_ws_words(count, ell, K, N)) return LF_ERR_ARG;
    if (one && K > LF_MODDOWN_ONE_MAX_K) return LF_ERR_ARG;
    if (count == 0 || ell == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    PtrBatch pb;
    for (int i = 0; i < count; ++i)
        pb.in[i] = (const i64 *)s[i], pb.aux[i] = addend ? (const i64 *)addend[i] : nullptr, pb.out[i] = (i64 *)out[i];
    hipStream_t st = (hipStream_t)stream;
    if (!one) {
        dim3 g1((unsigned)((N + 255) / 256), (unsigned)count + 1u);
        hipLaunchKernelGGL(ks_pivots_kernel, g1, dim3(256), 0, st, pb, count, ell, K, (i64)N, (i64 *)ws, (const i64 *)PiR, PiP,
                           (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
    }
    dim3 g2((unsigned)((N / 2 + 255) / 256), (unsigned)((ell + MD3_ROWS - 1) / MD3_ROWS), (unsigned)count);
#define LF_MD_ARGS                                                                                                        \
    g2, dim3(256), 0, st, pb, count, ell, (i64)N, (i64)gal_pinv, (const i64 *)gal_2q, (const i64 *)ws, (const i64 *)PiR,  \
        (const i64 *)Rs, PiP != nullptr ? 1 : 0, (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh
#define LF_MD_CASE(KK)                                                        \
    case KK:                                                                  \
        hipLaunchKernelGGL((ks_moddown_ws_kernel<KK, false>), LF_MD_ARGS);    \
        break;
    if (one) {
        if (K == 1) hipLaunchKernelGGL((ks_moddown_ws_kernel<1, true>), LF_MD_ARGS);
        else hipLaunchKernelGGL((ks_moddown_ws_kernel<2, true>), LF_MD_ARGS);
    } else {
        switch (K) {
            LF_MD_CASE(1) LF_MD_CASE(2) LF_MD_CASE(3) LF_MD_CASE(4) LF_MD_CASE(5) LF_MD_CASE(6) LF_MD_CASE(7) LF_MD_CASE(8)
        }
    }
#undef LF_MD_CASE
#undef LF_MD_ARGS
    return (int)hipGetLastError();
}

int lf_ks_moddown_ws(const int64_t *const *s, int64_t *const *out, const int64_t *const *addend, int count, int ell, int K,
                     int64_t N, int64_t *ws, int64_t ws_words, const int64_t *PiR, const double *PiP, const int64_t *Rs,
                     int64_t gal_pinv, const int64_t *gal_2q, const int64_t *ql, const int64_t *qh, const int64_t *kl,
                     const int64_t *kh, int device, void *stream) {
    return moddown_ws_launch(false, s, out, addend, count, ell, K, N, ws, ws_words, PiR, PiP, Rs, gal_pinv, gal_2q, ql, qh, kl, kh,
                             device, stream);
}

int lf_ks_moddown_one(const int64_t *const *s, int64_t *const *out, const int64_t *const *addend, int count, int ell, int K,
                      int64_t N, const int64_t *ws, int64_t ws_words, const int64_t *PiR, const double *PiP, const int64_t *Rs,
                      int64_t gal_pinv, const int64_t *gal_2q, const int64_t *ql, const int64_t *qh, const int64_t *kl,
                      const int64_t *kh, int device, void *stream) {
    return moddown_ws_launch(true, s, out, addend, count, ell, K, N, (int64_t *)ws, ws_words, PiR, PiP, Rs, gal_pinv, gal_2q, ql, qh,
                             kl, kh, device, stream);
}

int lf_ks_moddown_consts(int64_t *ws, int64_t ws_words, int count, int ell, int K, int64_t N, const double *PiP, const int64_t *ql,
                         const int64_t *qh, const int64_t *kl, const int64_t *kh, int device, void *stream) {
    if (count < 0 || count > LF_BATCH_MAX || ell < 0 || K < 1 || K > KS_MAX_K || N < 2 || (N & 1) || !ws ||
        ws_words < lf_ks_moddown_ws_words(count, ell, K, N))
        return LF_ERR_ARG;
    if (ell == 0 || PiP == nullptr) return 0;
    if (int e = lf_set_device(device)) return e;
    hipLaunchKernelGGL(ks_md_consts_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<double *>(ws + (int64_t)count * K * N),
                       ell, K, PiP, (const i64 *)ql, (const i64 *)qh, (const i64 *)kl, (const i64 *)kh);
    return (int)hipGetLastError();
}

int lf_ks_moddown(const int64_t *s, int64_t *out, const int64_t *addend, int ell, int K, int64_t N, const int64_t *PiR,
                  const double *PiP, const int64_t *Rs, const int64_t *ql, const int64_t *qh, const int64_t *kl,
                  const int64_t *kh, int device, void *stream) {
    return lf_ks_moddown_batch(&s, &out, &addend, 1, ell, K, N, PiR, PiP, Rs, 0, nullptr, ql, qh, kl, kh, device, stream);
}

}  // extern "C"
