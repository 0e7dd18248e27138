/*
 * ckks_oracle.c — CPU restatement of the reference's integer kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (liberate_fhe_amd/)
 * may link, import or call this file; it is the checker used by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * Parity status: the reference (Desilo/liberate-fhe v0.9.0) has no CPU path
 * and no golden vectors of its own for these kernels.  This restatement is
 * pinned two ways (see tests/test_oracle_*.py and tests/golden/):
 *   (1) against independent big-integer definitions (REDC closed form,
 *       NTT = evaluation at psi^(2*brev(k)+1), iNTT o NTT = id), and
 *   (2) by plugging it in as the `ntt_cuda` stand-in underneath the
 *       reference's own Python engine (imported from /root/reference in the
 *       build container only) and committing the captured input/output
 *       vectors under tests/golden/ (generator: tests/golden/make_golden.py).
 *
 * Every function cites the reference lines it follows; "K.cu" is
 * src/liberate/ntt/ntt_cuda_kernel.cu of the reference.
 *
 * Word model: signed 64-bit two's complement, wrapping arithmetic
 * (compile with -fwrapv), arithmetic right shift of negative values
 * (gcc/clang behaviour, same as the CUDA device code).
 */
#define _GNU_SOURCE
#include <sched.h>
#include <omp.h>
#include <stdint.h>
#include <string.h>
#include <stddef.h>

#define HALF 31
#define LB_MASK ((((int64_t)1) << 31) - 1)
#define FB_MASK ((((int64_t)1) << 62) - 1)

/* K.cu:12-59  mont_mult_scalar_cuda_kernel, scalar_t = int64 (nbits 62, half 31). */
static inline int64_t mm(int64_t a, int64_t b, int64_t ql, int64_t qh, int64_t kl, int64_t kh)
{
    const int64_t al = a & LB_MASK, ah = a >> HALF;
    const int64_t bl = b & LB_MASK, bh = b >> HALF;
    const int64_t alpha = ah * bh;
    const int64_t beta = ah * bl + al * bh;
    const int64_t gamma = al * bl;

    const int64_t gammal = gamma & LB_MASK, gammah = gamma >> HALF;
    const int64_t betal = beta & LB_MASK, betah = beta >> HALF;

    int64_t upper = gammal * kh;
    upper = upper + (gammah + betal) * kl;
    upper = (int64_t)((uint64_t)upper << HALF);
    int64_t s = upper + gammal * kl;
    s = s & FB_MASK;

    const int64_t sl = s & LB_MASK, sh = s >> HALF;
    const int64_t sqb = sh * ql + sl * qh;
    const int64_t sqbl = sqb & LB_MASK, sqbh = sqb >> HALF;

    int64_t carry = (gamma + sl * ql) >> HALF;
    carry = (carry + betal + sqbl) >> HALF;
    return alpha + betah + sqbh + carry + sh * qh;
}

/* K.cu:587-606  mont_redc_cuda_kernel body. */
static inline int64_t redc(int64_t x, int64_t ql, int64_t qh, int64_t kl, int64_t kh)
{
    const int64_t xl = x & LB_MASK, xh = x >> HALF;
    const int64_t xkb = xh * kl + xl * kh;
    int64_t s = (int64_t)((uint64_t)xkb << HALF) + xl * kl;
    s = s & FB_MASK;
    const int64_t sl = s & LB_MASK, sh = s >> HALF;
    const int64_t sqb = sh * ql + sl * qh;
    const int64_t sqbl = sqb & LB_MASK, sqbh = sqb >> HALF;
    int64_t carry = (x + sl * ql) >> HALF;
    carry = (carry + sqbl) >> HALF;
    return sqbh + carry + sh * qh;
}

int64_t lfo_mm_scalar(int64_t a, int64_t b, int64_t ql, int64_t qh, int64_t kl, int64_t kh)
{
    return mm(a, b, ql, qh, kl, kh);
}

int64_t lfo_redc_scalar(int64_t x, int64_t ql, int64_t qh, int64_t kl, int64_t kh)
{
    return redc(x, ql, qh, kl, kh);
}

/* K.cu:66-146 mont_mult: c[i][j] = mm(a[i][j], b[i][j]); extent = rows of a. */
void lfo_mont_mult(const int64_t *a, const int64_t *b, int64_t *c, int rows, int64_t N,
                   const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < rows; ++i)
        for (int64_t j = 0; j < N; ++j)
            c[i * N + j] = mm(a[i * N + j], b[i * N + j], ql[i], qh[i], kl[i], kh[i]);
}

/* K.cu:154-226 mont_enter: a[i][j] = mm(a[i][j], Rs[i]) in place. */
void lfo_mont_enter(int64_t *a, const int64_t *Rs, int rows, int64_t N,
                    const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < rows; ++i)
        for (int64_t j = 0; j < N; ++j)
            a[i * N + j] = mm(a[i * N + j], Rs[i], ql[i], qh[i], kl[i], kh[i]);
}

/* K.cu:559-653 mont_redc in place. */
void lfo_mont_redc(int64_t *a, int rows, int64_t N,
                   const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < rows; ++i)
        for (int64_t j = 0; j < N; ++j)
            a[i * N + j] = redc(a[i * N + j], ql[i], qh[i], kl[i], kh[i]);
}

/* ---- table-driven NTT stages, literally as launched by the reference ----
 * K.cu:236-275 (forward stage kernel), K.cu:318-322 (stage loop),
 * tables: even/odd [logN][N/2] int32, psi [rows][logN][N/2] int64.
 * The extent is `rows` = ql.size(0) (K.cu:298), the row pitch of `a` is N. */
static void ntt_stage_tab(int64_t *a, const int32_t *even, const int32_t *odd, const int64_t *psi_row,
                          int64_t half, int64_t _2q, int64_t ql, int64_t qh, int64_t kl, int64_t kh)
{
    for (int64_t j = 0; j < half; ++j) {
        const int32_t e = even[j], o = odd[j];
        const int64_t U = a[e];
        const int64_t S = psi_row[j];
        const int64_t O = a[o];
        const int64_t V = mm(S, O, ql, qh, kl, kh);
        const int64_t UplusV = U + V;
        const int64_t UminusV = U + _2q - V;
        a[e] = (UplusV < _2q) ? UplusV : UplusV - _2q;
        a[o] = (UminusV < _2q) ? UminusV : UminusV - _2q;
    }
}

/* K.cu:433-473 inverse stage kernel. */
static void intt_stage_tab(int64_t *a, const int32_t *even, const int32_t *odd, const int64_t *psi_row,
                           int64_t half, int64_t _2q, int64_t ql, int64_t qh, int64_t kl, int64_t kh)
{
    for (int64_t j = 0; j < half; ++j) {
        const int32_t e = even[j], o = odd[j];
        const int64_t U = a[e];
        const int64_t S = psi_row[j];
        const int64_t V = a[o];
        const int64_t UminusV = U + _2q - V;
        const int64_t O = (UminusV < _2q) ? UminusV : UminusV - _2q;
        const int64_t W = mm(S, O, ql, qh, kl, kh);
        a[o] = W;
        const int64_t UplusV = U + V;
        a[e] = (UplusV < _2q) ? UplusV : UplusV - _2q;
    }
}

/* K.cu:278-323 ntt_cuda_typed. */
void lfo_ntt_tab(int64_t *a, const int32_t *even, const int32_t *odd, const int64_t *psi,
                 int rows, int logN, int64_t N,
                 const int64_t *_2q, const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh)
{
    const int64_t half = N / 2;
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < rows; ++i)
        for (int s = 0; s < logN; ++s)
            ntt_stage_tab(a + (int64_t)i * N, even + s * half, odd + s * half,
                          psi + ((int64_t)i * logN + s) * half, half, _2q[i], ql[i], qh[i], kl[i], kh[i]);
}

/* K.cu:349-404 enter_ntt_cuda_typed: mont_enter(Rs) then the forward stages. */
void lfo_enter_ntt_tab(int64_t *a, const int64_t *Rs, const int32_t *even, const int32_t *odd, const int64_t *psi,
                       int rows, int logN, int64_t N,
                       const int64_t *_2q, const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh)
{
    lfo_mont_enter(a, Rs, rows, N, ql, qh, kl, kh);
    lfo_ntt_tab(a, even, odd, psi, rows, logN, N, _2q, ql, qh, kl, kh);
}

/* K.cu:476-530 intt_cuda_typed: inverse stages, then mont_enter with Ninv (K.cu:527-529). */
void lfo_intt_tab(int64_t *a, const int32_t *even, const int32_t *odd, const int64_t *psi, const int64_t *Ninv,
                  int rows, int logN, int64_t N,
                  const int64_t *_2q, const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh)
{
    const int64_t half = N / 2;
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < rows; ++i) {
        for (int s = 0; s < logN; ++s)
            intt_stage_tab(a + (int64_t)i * N, even + s * half, odd + s * half,
                           psi + ((int64_t)i * logN + s) * half, half, _2q[i], ql[i], qh[i], kl[i], kh[i]);
        for (int64_t j = 0; j < N; ++j)
            a[i * N + j] = mm(a[i * N + j], Ninv[i], ql[i], qh[i], kl[i], kh[i]);
    }
}

/* ---- range fix-ups ---- */

/* K.cu:664-680 reduce_cuda_kernel ("reduce_2q"): ONE conditional subtraction of q = _2q >> 1. */
void lfo_reduce_2q(int64_t *a, int rows, int64_t N, const int64_t *_2q)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < rows; ++i) {
        const int64_t q = _2q[i] >> 1;
        for (int64_t j = 0; j < N; ++j) {
            const int64_t v = a[i * N + j];
            a[i * N + j] = (v < q) ? v : v - q;
        }
    }
}

/* K.cu:682-699 make_signed. */
void lfo_make_signed(int64_t *a, int rows, int64_t N, const int64_t *_2q)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < rows; ++i) {
        const int64_t q = _2q[i] >> 1, q_half = q >> 1;
        for (int64_t j = 0; j < N; ++j) {
            const int64_t v = a[i * N + j];
            a[i * N + j] = (v <= q_half) ? v : v - q;
        }
    }
}

/* K.cu:980-995 make_unsigned: a += q. */
void lfo_make_unsigned(int64_t *a, int rows, int64_t N, const int64_t *_2q)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < rows; ++i) {
        const int64_t q = _2q[i] >> 1;
        for (int64_t j = 0; j < N; ++j) a[i * N + j] += q;
    }
}

/* K.cu:997-1014 tile_unsigned: dst[i][j] = a[j] + q_i, rows = _2q.size(0) (K.cu:1207). */
void lfo_tile_unsigned(const int64_t *a, int64_t *dst, int rows, int64_t N, const int64_t *_2q)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < rows; ++i) {
        const int64_t q = _2q[i] >> 1;
        for (int64_t j = 0; j < N; ++j) dst[i * N + j] = a[j] + q;
    }
}

/* K.cu:1016-1036 mont_add. */
void lfo_mont_add(const int64_t *a, const int64_t *b, int64_t *c, int rows, int64_t N, const int64_t *_2q)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < rows; ++i)
        for (int64_t j = 0; j < N; ++j) {
            const int64_t s = a[i * N + j] + b[i * N + j];
            c[i * N + j] = (s < _2q[i]) ? s : s - _2q[i];
        }
}

/* K.cu:1038-1058 mont_sub. */
void lfo_mont_sub(const int64_t *a, const int64_t *b, int64_t *c, int rows, int64_t N, const int64_t *_2q)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < rows; ++i)
        for (int64_t j = 0; j < N; ++j) {
            const int64_t s = a[i * N + j] + _2q[i] - b[i * N + j];
            c[i * N + j] = (s < _2q[i]) ? s : s - _2q[i];
        }
}

/* ---- formula-indexed variants over the compact per-prime table ----
 * Same butterfly DAG and per-butterfly formulas as the table-driven stages
 * above; the gather indices and the twiddle index are computed instead of
 * looked up, following the paint loops of the reference's
 * src/liberate/fhe/context/ckks_context.py:89-112 (forward: stage s, m = 2^s,
 * t = N/2m, block i < m pairs j and j+t for j in [2it, 2it+t), twiddle index
 * m+i) and :115-142 (backward: stage s, t = 2^s, h = N/2t, block i < h pairs
 * j and j+t for j in [2it, 2it+t), twiddle index h+i).
 * psi_br / ipsi_br: [rows][N] int64, entry x = Montgomery form of
 * psi^brev(x) (resp. psi^-brev(x)) exactly as produced by the reference's
 * on-device mont_enter of the table (ntt_context.py:115-130). */
void lfo_ntt(int64_t *a, const int64_t *psi_br, int rows, int logN,
             const int64_t *_2q, const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh)
{
    const int64_t N = (int64_t)1 << logN;
    /* static schedule: row r always runs on the same thread, the one lfo_place_rows let touch its pages first */
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        int64_t *x = a + (int64_t)r * N;
        const int64_t *w = psi_br + (int64_t)r * N;
        int64_t t = N;
        for (int64_t m = 1; m < N; m <<= 1) {
            t >>= 1;
            for (int64_t i = 0; i < m; ++i) {
                const int64_t S = w[m + i];
                const int64_t j1 = 2 * i * t;
                for (int64_t j = j1; j < j1 + t; ++j) {
                    const int64_t U = x[j];
                    const int64_t V = mm(S, x[j + t], ql[r], qh[r], kl[r], kh[r]);
                    const int64_t p = U + V, d = U + _2q[r] - V;
                    x[j] = (p < _2q[r]) ? p : p - _2q[r];
                    x[j + t] = (d < _2q[r]) ? d : d - _2q[r];
                }
            }
        }
    }
}

void lfo_intt(int64_t *a, const int64_t *ipsi_br, const int64_t *Ninv, int rows, int logN,
              const int64_t *_2q, const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh)
{
    const int64_t N = (int64_t)1 << logN;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        int64_t *x = a + (int64_t)r * N;
        const int64_t *w = ipsi_br + (int64_t)r * N;
        int64_t t = 1;
        for (int64_t h = N >> 1; h >= 1; h >>= 1) {
            for (int64_t i = 0; i < h; ++i) {
                const int64_t S = w[h + i];
                const int64_t j1 = 2 * i * t;
                for (int64_t j = j1; j < j1 + t; ++j) {
                    const int64_t U = x[j], V = x[j + t];
                    const int64_t d = U + _2q[r] - V;
                    const int64_t O = (d < _2q[r]) ? d : d - _2q[r];
                    x[j + t] = mm(S, O, ql[r], qh[r], kl[r], kh[r]);
                    const int64_t p = U + V;
                    x[j] = (p < _2q[r]) ? p : p - _2q[r];
                }
            }
            t <<= 1;
        }
        for (int64_t j = 0; j < N; ++j) x[j] = mm(x[j], Ninv[r], ql[r], qh[r], kl[r], kh[r]);
    }
}

/* Timing aid of bench.py's cpu_baseline (no reference counterpart): OpenMP thread t of the team of `n` pins itself to
 * logical CPU cpus[t] (one per physical core, chosen by the caller).  Returns the number of threads that could not. */
int lfo_pin_threads(const int *cpus, int n)
{
    int failed = 0;
#pragma omp parallel num_threads(n) reduction(+ : failed)
    {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(cpus[omp_get_thread_num()], &set);
        if (sched_setaffinity(0, sizeof(set), &set) != 0) failed += 1;
    }
    return failed;
}

/* Timing aid of bench.py's cpu_baseline (no reference counterpart): dst[r] = src[r mod src_rows] for r < rows, copied by
 * the thread that lfo_ntt / lfo_intt will run row r on (same static schedule), so that on a multi-socket host every
 * row's pages are first touched — and therefore placed — next to the core that transforms it.  dst must be fresh,
 * untouched memory (np.empty) for the placement to happen. */
void lfo_place_rows(int64_t *dst, const int64_t *src, int rows, int src_rows, int64_t N)
{
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r)
        memcpy(dst + (int64_t)r * N, src + (int64_t)(r % src_rows) * N, (size_t)N * sizeof(int64_t));
}

/* Galois automorphism on coefficient rows, following the reference's
 * src/liberate/fhe/encdec/encdec.py:224-270 (rotate / conjugate):
 * coefficient n goes to index (p*n mod 2N) mod N with sign -1 iff
 * (p*n mod 2N) >= N; p = 3^delta mod 2N for rotate, 2N-1 for conjugate.
 * Caller passes p.  Output is signed (no reduction), as in the reference. */
void lfo_galois(const int64_t *a, int64_t *dst, int rows, int64_t N, int64_t p)
{
    const int64_t M = 2 * N;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < rows; ++i)
        for (int64_t n = 0; n < N; ++n) {
            const int64_t pn = (int64_t)(((__int128)p * n) % M);
            const int64_t v = a[i * N + n];
            dst[i * N + (pn % N)] = (pn >= N) ? -v : v;
        }
}
