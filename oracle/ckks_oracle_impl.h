/* ckks_oracle_impl.h — the body of the CPU oracle, written once over the word type and compiled twice by ckks_oracle.c:
 *   word_t = int64_t, HALF 31  -> lfo_*    the reference's 62-bit mode  (K.cu: scalar_t = int64)
 *   word_t = int32_t, HALF 15  -> lfo30_*  its 30-bit mode              (K.cu: scalar_t = int32; ckks_context.py:213-216)
 * The reference's kernels are templates over scalar_t (K.cu:12-59: nbits = 8 sizeof - 2, half = 4 sizeof - 1); this is
 * the same text over the same two types.  TEST INFRASTRUCTURE ONLY (see ckks_oracle.c).  Index arithmetic is done in
 * word_t as well: rows * N stays below 2^31 for every ring the reference ships. */
#define LB_MASK ((((word_t)1) << HALF) - 1)
#define FB_MASK ((((word_t)1) << (2 * HALF)) - 1)


/* K.cu:12-59  mont_mult_scalar_cuda_kernel, scalar_t = int64 (nbits 62, half 31). */
static inline word_t LFO_S(mm)(word_t a, word_t b, word_t ql, word_t qh, word_t kl, word_t kh)
{
    const word_t al = a & LB_MASK, ah = a >> HALF;
    const word_t bl = b & LB_MASK, bh = b >> HALF;
    const word_t alpha = ah * bh;
    const word_t beta = ah * bl + al * bh;
    const word_t gamma = al * bl;

    const word_t gammal = gamma & LB_MASK, gammah = gamma >> HALF;
    const word_t betal = beta & LB_MASK, betah = beta >> HALF;

    word_t upper = gammal * kh;
    upper = upper + (gammah + betal) * kl;
    upper = (word_t)((uword_t)upper << HALF);
    word_t s = upper + gammal * kl;
    s = s & FB_MASK;

    const word_t sl = s & LB_MASK, sh = s >> HALF;
    const word_t sqb = sh * ql + sl * qh;
    const word_t sqbl = sqb & LB_MASK, sqbh = sqb >> HALF;

    word_t carry = (gamma + sl * ql) >> HALF;
    carry = (carry + betal + sqbl) >> HALF;
    return alpha + betah + sqbh + carry + sh * qh;
}

/* K.cu:587-606  mont_redc_cuda_kernel body. */
static inline word_t LFO_S(redc)(word_t x, word_t ql, word_t qh, word_t kl, word_t kh)
{
    const word_t xl = x & LB_MASK, xh = x >> HALF;
    const word_t xkb = xh * kl + xl * kh;
    word_t s = (word_t)((uword_t)xkb << HALF) + xl * kl;
    s = s & FB_MASK;
    const word_t sl = s & LB_MASK, sh = s >> HALF;
    const word_t sqb = sh * ql + sl * qh;
    const word_t sqbl = sqb & LB_MASK, sqbh = sqb >> HALF;
    word_t carry = (x + sl * ql) >> HALF;
    carry = (carry + sqbl) >> HALF;
    return sqbh + carry + sh * qh;
}

word_t LFO(mm_scalar)(word_t a, word_t b, word_t ql, word_t qh, word_t kl, word_t kh)
{
    return LFO_S(mm)(a, b, ql, qh, kl, kh);
}

word_t LFO(redc_scalar)(word_t x, word_t ql, word_t qh, word_t kl, word_t kh)
{
    return LFO_S(redc)(x, ql, qh, kl, kh);
}

/* K.cu:66-146 mont_mult: c[i][j] = LFO_S(mm)(a[i][j], b[i][j]); extent = rows of a. */
void LFO(mont_mult)(const word_t *a, const word_t *b, word_t *c, int rows, word_t N,
                   const word_t *ql, const word_t *qh, const word_t *kl, const word_t *kh)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < rows; ++i)
        for (word_t j = 0; j < N; ++j)
            c[i * N + j] = LFO_S(mm)(a[i * N + j], b[i * N + j], ql[i], qh[i], kl[i], kh[i]);
}

/* K.cu:154-226 mont_enter: a[i][j] = LFO_S(mm)(a[i][j], Rs[i]) in place. */
void LFO(mont_enter)(word_t *a, const word_t *Rs, int rows, word_t N,
                    const word_t *ql, const word_t *qh, const word_t *kl, const word_t *kh)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < rows; ++i)
        for (word_t j = 0; j < N; ++j)
            a[i * N + j] = LFO_S(mm)(a[i * N + j], Rs[i], ql[i], qh[i], kl[i], kh[i]);
}

/* K.cu:559-653 mont_redc in place. */
void LFO(mont_redc)(word_t *a, int rows, word_t N,
                   const word_t *ql, const word_t *qh, const word_t *kl, const word_t *kh)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < rows; ++i)
        for (word_t j = 0; j < N; ++j)
            a[i * N + j] = LFO_S(redc)(a[i * N + j], ql[i], qh[i], kl[i], kh[i]);
}

/* ---- table-driven NTT stages, literally as launched by the reference ----
 * K.cu:236-275 (forward stage kernel), K.cu:318-322 (stage loop),
 * tables: even/odd [logN][N/2] int32, psi [rows][logN][N/2] int64.
 * The extent is `rows` = ql.size(0) (K.cu:298), the row pitch of `a` is N. */
static void LFO_S(ntt_stage_tab)(word_t *a, const int32_t *even, const int32_t *odd, const word_t *psi_row,
                          word_t half, word_t _2q, word_t ql, word_t qh, word_t kl, word_t kh)
{
    for (word_t j = 0; j < half; ++j) {
        const int32_t e = even[j], o = odd[j];
        const word_t U = a[e];
        const word_t S = psi_row[j];
        const word_t O = a[o];
        const word_t V = LFO_S(mm)(S, O, ql, qh, kl, kh);
        const word_t UplusV = U + V;
        const word_t UminusV = U + _2q - V;
        a[e] = (UplusV < _2q) ? UplusV : UplusV - _2q;
        a[o] = (UminusV < _2q) ? UminusV : UminusV - _2q;
    }
}

/* K.cu:433-473 inverse stage kernel. */
static void LFO_S(intt_stage_tab)(word_t *a, const int32_t *even, const int32_t *odd, const word_t *psi_row,
                           word_t half, word_t _2q, word_t ql, word_t qh, word_t kl, word_t kh)
{
    for (word_t j = 0; j < half; ++j) {
        const int32_t e = even[j], o = odd[j];
        const word_t U = a[e];
        const word_t S = psi_row[j];
        const word_t V = a[o];
        const word_t UminusV = U + _2q - V;
        const word_t O = (UminusV < _2q) ? UminusV : UminusV - _2q;
        const word_t W = LFO_S(mm)(S, O, ql, qh, kl, kh);
        a[o] = W;
        const word_t UplusV = U + V;
        a[e] = (UplusV < _2q) ? UplusV : UplusV - _2q;
    }
}

/* K.cu:278-323 ntt_cuda_typed. */
void LFO(ntt_tab)(word_t *a, const int32_t *even, const int32_t *odd, const word_t *psi,
                 int rows, int logN, word_t N,
                 const word_t *_2q, const word_t *ql, const word_t *qh, const word_t *kl, const word_t *kh)
{
    const word_t half = N / 2;
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < rows; ++i)
        for (int s = 0; s < logN; ++s)
            LFO_S(ntt_stage_tab)(a + (word_t)i * N, even + s * half, odd + s * half,
                          psi + ((word_t)i * logN + s) * half, half, _2q[i], ql[i], qh[i], kl[i], kh[i]);
}

/* K.cu:349-404 enter_ntt_cuda_typed: mont_enter(Rs) then the forward stages. */
void LFO(enter_ntt_tab)(word_t *a, const word_t *Rs, const int32_t *even, const int32_t *odd, const word_t *psi,
                       int rows, int logN, word_t N,
                       const word_t *_2q, const word_t *ql, const word_t *qh, const word_t *kl, const word_t *kh)
{
    LFO(mont_enter)(a, Rs, rows, N, ql, qh, kl, kh);
    LFO(ntt_tab)(a, even, odd, psi, rows, logN, N, _2q, ql, qh, kl, kh);
}

/* K.cu:476-530 intt_cuda_typed: inverse stages, then mont_enter with Ninv (K.cu:527-529). */
void LFO(intt_tab)(word_t *a, const int32_t *even, const int32_t *odd, const word_t *psi, const word_t *Ninv,
                  int rows, int logN, word_t N,
                  const word_t *_2q, const word_t *ql, const word_t *qh, const word_t *kl, const word_t *kh)
{
    const word_t half = N / 2;
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < rows; ++i) {
        for (int s = 0; s < logN; ++s)
            LFO_S(intt_stage_tab)(a + (word_t)i * N, even + s * half, odd + s * half,
                           psi + ((word_t)i * logN + s) * half, half, _2q[i], ql[i], qh[i], kl[i], kh[i]);
        for (word_t j = 0; j < N; ++j)
            a[i * N + j] = LFO_S(mm)(a[i * N + j], Ninv[i], ql[i], qh[i], kl[i], kh[i]);
    }
}

/* ---- range fix-ups ---- */

/* K.cu:664-680 reduce_cuda_kernel ("reduce_2q"): ONE conditional subtraction of q = _2q >> 1. */
void LFO(reduce_2q)(word_t *a, int rows, word_t N, const word_t *_2q)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < rows; ++i) {
        const word_t q = _2q[i] >> 1;
        for (word_t j = 0; j < N; ++j) {
            const word_t v = a[i * N + j];
            a[i * N + j] = (v < q) ? v : v - q;
        }
    }
}

/* K.cu:682-699 make_signed. */
void LFO(make_signed)(word_t *a, int rows, word_t N, const word_t *_2q)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < rows; ++i) {
        const word_t q = _2q[i] >> 1, q_half = q >> 1;
        for (word_t j = 0; j < N; ++j) {
            const word_t v = a[i * N + j];
            a[i * N + j] = (v <= q_half) ? v : v - q;
        }
    }
}

/* K.cu:980-995 make_unsigned: a += q. */
void LFO(make_unsigned)(word_t *a, int rows, word_t N, const word_t *_2q)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < rows; ++i) {
        const word_t q = _2q[i] >> 1;
        for (word_t j = 0; j < N; ++j) a[i * N + j] += q;
    }
}

/* K.cu:997-1014 tile_unsigned: dst[i][j] = a[j] + q_i, rows = _2q.size(0) (K.cu:1207). */
void LFO(tile_unsigned)(const word_t *a, word_t *dst, int rows, word_t N, const word_t *_2q)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < rows; ++i) {
        const word_t q = _2q[i] >> 1;
        for (word_t j = 0; j < N; ++j) dst[i * N + j] = a[j] + q;
    }
}

/* K.cu:1016-1036 mont_add. */
void LFO(mont_add)(const word_t *a, const word_t *b, word_t *c, int rows, word_t N, const word_t *_2q)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < rows; ++i)
        for (word_t j = 0; j < N; ++j) {
            const word_t s = a[i * N + j] + b[i * N + j];
            c[i * N + j] = (s < _2q[i]) ? s : s - _2q[i];
        }
}

/* K.cu:1038-1058 mont_sub. */
void LFO(mont_sub)(const word_t *a, const word_t *b, word_t *c, int rows, word_t N, const word_t *_2q)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < rows; ++i)
        for (word_t j = 0; j < N; ++j) {
            const word_t s = a[i * N + j] + _2q[i] - b[i * N + j];
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
void LFO(ntt)(word_t *a, const word_t *psi_br, int rows, int logN,
             const word_t *_2q, const word_t *ql, const word_t *qh, const word_t *kl, const word_t *kh)
{
    const word_t N = (word_t)1 << logN;
    /* static schedule: row r always runs on the same thread, the one LFO(place_rows) let touch its pages first */
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        word_t *x = a + (word_t)r * N;
        const word_t *w = psi_br + (word_t)r * N;
        word_t t = N;
        for (word_t m = 1; m < N; m <<= 1) {
            t >>= 1;
            for (word_t i = 0; i < m; ++i) {
                const word_t S = w[m + i];
                const word_t j1 = 2 * i * t;
                for (word_t j = j1; j < j1 + t; ++j) {
                    const word_t U = x[j];
                    const word_t V = LFO_S(mm)(S, x[j + t], ql[r], qh[r], kl[r], kh[r]);
                    const word_t p = U + V, d = U + _2q[r] - V;
                    x[j] = (p < _2q[r]) ? p : p - _2q[r];
                    x[j + t] = (d < _2q[r]) ? d : d - _2q[r];
                }
            }
        }
    }
}

void LFO(intt)(word_t *a, const word_t *ipsi_br, const word_t *Ninv, int rows, int logN,
              const word_t *_2q, const word_t *ql, const word_t *qh, const word_t *kl, const word_t *kh)
{
    const word_t N = (word_t)1 << logN;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        word_t *x = a + (word_t)r * N;
        const word_t *w = ipsi_br + (word_t)r * N;
        word_t t = 1;
        for (word_t h = N >> 1; h >= 1; h >>= 1) {
            for (word_t i = 0; i < h; ++i) {
                const word_t S = w[h + i];
                const word_t j1 = 2 * i * t;
                for (word_t j = j1; j < j1 + t; ++j) {
                    const word_t U = x[j], V = x[j + t];
                    const word_t d = U + _2q[r] - V;
                    const word_t O = (d < _2q[r]) ? d : d - _2q[r];
                    x[j + t] = LFO_S(mm)(S, O, ql[r], qh[r], kl[r], kh[r]);
                    const word_t p = U + V;
                    x[j] = (p < _2q[r]) ? p : p - _2q[r];
                }
            }
            t <<= 1;
        }
        for (word_t j = 0; j < N; ++j) x[j] = LFO_S(mm)(x[j], Ninv[r], ql[r], qh[r], kl[r], kh[r]);
    }
}



/* Galois automorphism on coefficient rows, following the reference's
 * src/liberate/fhe/encdec/encdec.py:224-270 (rotate / conjugate):
 * coefficient n goes to index (p*n mod 2N) mod N with sign -1 iff
 * (p*n mod 2N) >= N; p = 3^delta mod 2N for rotate, 2N-1 for conjugate.
 * Caller passes p.  Output is signed (no reduction), as in the reference. */
void LFO(galois)(const word_t *a, word_t *dst, int rows, word_t N, word_t p)
{
    const word_t M = 2 * N;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < rows; ++i)
        for (word_t n = 0; n < N; ++n) {
            const word_t pn = (word_t)(((__int128)p * n) % M);
            const word_t v = a[i * N + n];
            dst[i * N + (pn % N)] = (pn >= N) ? -v : v;
        }
}

#undef LB_MASK
#undef FB_MASK
