/*
 * ckks_hip.h — C ABI of libckks_hip.so, the MI355X (gfx950) RNS-CKKS arithmetic library.
 *
 * This is the drop-in boundary for the reference's pybind11 module `liberate.ntt.ntt_cuda`
 * (reference: src/liberate/ntt/ntt.cpp:421-437 exports 15 functions over std::vector<torch::Tensor>,
 * one tensor per GPU).  Each entry point below replaces ONE of those functions for ONE device: the
 * per-GPU loop of ntt.cpp:130-141 lives in the caller (one process per GPU, or the Python shim
 * liberate_fhe_amd/ntt/ntt_cuda.py which walks the tensor lists).
 *
 * Conventions
 *   - all pointers are DEVICE pointers on `device`; data and constants are int64 (the reference's
 *     62-bit word mode, Montgomery radix R = 2^62, values lazily in [0, 2q));
 *   - polynomials are row-major [rows][N], one RNS limb per row, row pitch = N words;
 *   - per-row constant vectors are indexed by row id exactly as the reference kernels do
 *     (K.cu = src/liberate/ntt/ntt_cuda_kernel.cu): ql/qh = q & (2^31-1), q >> 31; kl/kh likewise
 *     for k = -q^-1 mod 2^62; _2q = 2q;
 *   - the reference's implicit extent rules are explicit arguments: elementwise ops run over
 *     `rows` = a.size(0) (K.cu:110-114), NTT-family ops over `rows` = ql.size(0) (K.cu:298);
 *   - twiddles are the compact table psi_br[rows][N] (entry x = Montgomery form of psi^brev(x)),
 *     not the reference's [rows][logN][N/2] per-stage table; the butterfly DAG and per-butterfly
 *     formulas are the reference's, so every lazy output word is bit-identical;
 *   - `stream` is a hipStream_t (0 = the null stream); launches are asynchronous, nothing syncs;
 *   - return value: 0 on success, otherwise the hipError_t of the failed call/launch, or
 *     LF_ERR_ARG for an invalid argument.  (The reference returns nothing and checks nothing.)
 */
#ifndef CKKS_HIP_H
#define CKKS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LF_ERR_ARG 10001
/* Two halves of one operation were called under different lf_tune settings: the scratch the first half left is in another
 * format than the second half is about to read (lf_ks_fwd -> lf_ks_tail, lf_cc_mult_evk_pre -> _post, lf_ntt_pass_ws 1 -> 2).
 * Nothing is launched; repeat the first half. */
#define LF_ERR_STATE 10002

/* Library probe: returns the ABI version (currently LF_ABI_VERSION; __graft_entry__.build() asserts it). */
#define LF_ABI_VERSION 15
int lf_abi_version(void);

/* Compile-time capacities of the fused kernels, so that callers can refuse a parameter set BEFORE any launch
 * (the digit width alpha and K live in device-side descriptors the ABI cannot check):
 *   which = 0: limbs per key-switch digit (alpha), 1: special primes K, 2: limb rows per call (`rows`),
 *           3: operand sets per batched call (count), 4: largest logN of the NTT family.  Other: -1. */
#define LF_LIMIT_DIGIT_LIMBS 0
#define LF_LIMIT_SPECIAL_PRIMES 1
#define LF_LIMIT_ROWS 2
#define LF_LIMIT_BATCH 3
#define LF_LIMIT_LOGN 4
int lf_limits(int which);

/* Launch-shape thresholds (never results: every setting produces the same words).  Returns the previous value, -1 for an
 * unknown `which`; value < 0 only reads.  PROCESS-WIDE mutable state, read by every entry at launch time on whatever
 * thread calls it and not synchronised: set it once, before any other thread launches (the engine never touches it; only
 * the A/B tools under tools/ do).
 *   LF_TUNE_KS_EXT_COLS_MAX   largest logN - 12 (0 .. 5, default 5) for which the key switch's extension + leading stages run as
 *                             the column kernel (one register step per column, no LDS); above it the LDS-tiled form.
 *   LF_TUNE_INTT_DIGITS       1 (default): lf_cc_mult_evk(_batch / _pre) form the digits of x1 * y1 inside the last inverse pass
 *                             where a digit's limbs fit a column thread (lf_intt_mul_digits); 0: always the two launches.
 *   LF_TUNE_DIGIT_PLANES      1 (default): between the halves of a key switch (lf_ks_fwd -> lf_ks_tail and every entry built on
 *                             them) the fp64-class rows of the scratch `tmp` hold their words as two planes, 6 bytes per word
 *                             (u32 low[N], u16 high[N] behind them) instead of 8, at logN >= 13 when the limbs are of both
 *                             classes; 0: raw words.  `tmp` is scratch either way; the knob must not change
 *                             between an lf_ks_fwd and its lf_ks_tail: the tail then returns LF_ERR_STATE instead of reading the wrong format.
 *   LF_TUNE_WS_EXTRA_STAGE    1 (default): in lf_ntt_ws at logN 13 .. 16 the column pass takes one stage more than logN - 12 (it is
 *                             HBM-bound with issue slots to spare) and the tiled pass, which is issue-bound, skips its first; 0: the
 *                             split of lf_ntt.  A change between the two launches of a transform (lf_ntt_pass_ws 1 -> 2): LF_ERR_STATE.
 *   (knob 0 was the one-launch key-switch transform of round 3: slower at every preset size on MI355X, removed.) */
#define LF_TUNE_KS_EXT_COLS_MAX 1
#define LF_TUNE_INTT_DIGITS 2
#define LF_TUNE_DIGIT_PLANES 3
#define LF_TUNE_WS_EXTRA_STAGE 4
/*   LF_TUNE_MORE_PLANES       with LF_TUNE_DIGIT_PLANES on: bit 0 (default 1) the SUMS of a key switch travel from the inner product through
 *                             the tiled inverse pass to the column pass as planes (through `tmp`, two digits or more); bit 1
 *                             (default 1) lf_stack_planes() answers 1: the op entries keep cc_mult's operand stack as planes. */
#define LF_TUNE_MORE_PLANES 5
int lf_tune(int which, int value);

/* Measurement entry (not one of the reference's ops; the engine never calls it): ONE wave, launched on `stream`, takes
 * `samples` readings of the shader clock — out[2 i] = core-clock cycles (s_memtime) that passed during out[2 i + 1] ticks
 * (>= `ticks`) of the constant 100 MHz counter (s_memrealtime), so MHz = 100 * out[2 i] / out[2 i + 1].  Launched on a
 * second stream beside a kernel under test it reports the clock that kernel actually runs at (bench.py: the headline
 * kernels run with the package at its 1 400 W cap).  out: device, 2 * samples 64-bit words;
 * samples <= 4096, samples * ticks <= 10^9 (ten seconds of spinning), else LF_ERR_ARG. */
int lf_clock_probe(uint64_t *out, int samples, uint64_t ticks, int device, void *stream);

/* ---- elementwise family --------------------------------------------------------------------- */

/* ntt_cuda.mont_mult (ntt.cpp:120-144, K.cu:66-146): c[i][j] = REDC62(a[i][j] * b[i][j]). */
int lf_mont_mult(const int64_t *a, const int64_t *b, int64_t *c, int rows, int64_t N,
                 const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh,
                 int device, void *stream);

/* ntt_cuda.mont_enter (ntt.cpp:146-163, K.cu:154-226): a[i][j] = REDC62(a[i][j] * Rs[i]) in place. */
int lf_mont_enter(int64_t *a, const int64_t *Rs, int rows, int64_t N,
                  const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh,
                  int device, void *stream);

/* ntt_cuda.mont_redc (ntt.cpp:248-263, K.cu:559-653): a = (a + ((a*k) mod R) * q) / R in place. */
int lf_mont_redc(int64_t *a, int rows, int64_t N,
                 const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh,
                 int device, void *stream);

/* ntt_cuda.reduce_2q (K.cu:664-680, 1187-1191): a = a < q ? a : a - q, q = _2q >> 1. */
int lf_reduce_2q(int64_t *a, int rows, int64_t N, const int64_t *_2q, int device, void *stream);
/* ntt_cuda.make_signed (K.cu:682-699): a = a <= q/2 ? a : a - q. */
int lf_make_signed(int64_t *a, int rows, int64_t N, const int64_t *_2q, int device, void *stream);
/* ntt_cuda.make_unsigned (K.cu:980-995): a += q. */
int lf_make_unsigned(int64_t *a, int rows, int64_t N, const int64_t *_2q, int device, void *stream);
/* ntt_cuda.tile_unsigned (K.cu:997-1014, 1205-1214): dst[i][j] = a[j] + q_i, rows = _2q.size(0). */
int lf_tile_unsigned(const int64_t *a, int64_t *dst, int rows, int64_t N, const int64_t *_2q,
                     int device, void *stream);
/* ntt_cuda.mont_add / mont_sub (K.cu:1016-1058): c = (a +/- b) csub 2q. */
int lf_mont_add(const int64_t *a, const int64_t *b, int64_t *c, int rows, int64_t N, const int64_t *_2q,
                int device, void *stream);
int lf_mont_sub(const int64_t *a, const int64_t *b, int64_t *c, int rows, int64_t N, const int64_t *_2q,
                int device, void *stream);

/* ---- NTT family -------------------------------------------------------------------------------
 * `batch` polynomials of `rows` limbs each, stored back to back ([batch][rows][N]); limb i of every
 * polynomial uses constant/twiddle row i.  The reference API is batch = 1.
 *
 * psi_dp / ipsi_dp (optional for the exact ops, may be NULL; REQUIRED with LF_NTT_RELAXED and by lf_ks_*): the
 * AUXILIARY twiddle table, one row of 2N 8-byte words per limb, filled by lf_twiddle_dp.  Limbs whose prime is
 * below 2^41 hold the plain canonical twiddles as doubles in words [0, N) and run the fp64-FMA butterfly path;
 * results are bit-identical to the integer path.  Limbs with a larger prime hold N pairs (floor(w 2^64 / q), w)
 * with w the plain twiddle: the Shoup products of the relaxed transforms (the exact ops never read them).
 * q_host (optional for the exact ops, may be NULL; REQUIRED with LF_NTT_RELAXED): HOST array of the `rows` primes,
 * used to split the rows into the two arithmetic classes at launch time (each class has its own kernel
 * instantiation); with NULL every row of an exact op runs the integer class.  A relaxed call without psi_dp or
 * without q_host returns LF_ERR_ARG before anything is launched: lf_twiddle_dp lays each auxiliary row out by the
 * size of its prime, so the launch must know the primes to read it.
 * flags: LF_NTT_RELAXED = the caller only needs the result modulo q (outputs are then canonical
 * residues instead of the reference's lazy representatives) — for fused internal use, never for the
 * drop-in ops.  A relaxed FORWARD transform accepts the reference's signed-lazy words (|a| < 2q); a relaxed
 * INVERSE transform takes non-negative words: below 2^52 on fp64-class limbs, lazy words in [0, 2q) on integer-class
 * limbs (what lf_tensor / lf_ks_inner / the fused core write). */
#define LF_NTT_RELAXED 1
/* with LF_NTT_RELAXED: fp64-class limbs stay in the PLAIN domain — lf_ntt applies Rs to integer-class limbs
 * only, lf_intt (tail >= 2) multiplies fp64-class limbs by N^-1 instead of N^-1 R^-1.  Used by the fused
 * cc_mult, whose tensor product then needs one plain modular product per term (lf_tensor, plain = 1). */
#define LF_NTT_PLAIN 2
/* lf_rescale_ntt at two-launch ring degrees (logN 13 .. 16) only: enqueue ONE of its two launches — the column pass, the only
 * one that reads `in` / `row0` (addresses that change from call to call), or the tiled pass, which touches `x` alone (a
 * caller may keep it, with the launches behind it, in a HIP graph: lf_cc_mult_evk_pre's `which`).  Not both. */
#define LF_NTT_ONLY_COLS 4
#define LF_NTT_ONLY_TILED 8
/* The operand stack of the fused cc_mult — what lf_rescale_ntt writes with LF_NTT_RELAXED | LF_NTT_PLAIN and only lf_intt_mul(_digits)
 * and lf_relin_* read — may keep its fp64-class rows (primes below 2^41: canonical words below 2^41) as two PLANES: u32 low[N] at
 * byte 0 of the row's 8 N bytes, u16 high[N] at byte 4 N, 6 bytes per word on each of the stack's four trips through HBM;
 * integer-class rows stay raw words.  lf_rescale_ntt with LF_NTT_PLANES writes that format (two-launch ring degrees, rows of both
 * classes: lf_stack_planes() says when; otherwise LF_ERR_ARG), lf_intt_mul(_digits) with LF_NTT_PLANES read their factors in
 * it, lf_relin_core_batch / lf_relin_tail with LF_STACK_PLANES OR-ed into `key_format` read `x` in it.  A stack written in one
 * format and read in the other: LF_ERR_STATE. */
#define LF_NTT_PLANES 16
#define LF_STACK_PLANES 4
/* 1 when internal stacks of the rows [0, rows) of q_host (HOST) keep fp64-class rows as planes: lf_tune(LF_TUNE_DIGIT_PLANES)
 * is on, 13 <= logN <= 24, and the rows hold primes of both arithmetic classes.  (The engine-op entries ask this themselves.) */
int lf_stack_planes(int logN, int rows, const int64_t *q_host);

/* The auxiliary table from the Montgomery-form compact table mont[rows][N]: out[rows][2N] (8-byte words).
 * Primes below 2^41: out[r][j] = (double)reduce_q(redc(mont[r][j])) for j < N; entry 0 of the row (psi^0 = 1, which
 * no butterfly stage reads) receives 1 / q_row instead — the fp64-class kernels take the reciprocal from there; words
 * [N, 2N) are unused.  Larger primes: pairs (out[r][2j], out[r][2j + 1]) = (floor(w 2^64 / q), w), w = reduce_q(redc(
 * mont[r][j])), as unsigned 64-bit integers.  Tables handed to lf_ntt / lf_intt / lf_ks_* must come from here. */
int lf_twiddle_dp(const int64_t *mont, double *out, int rows, int64_t N, const int64_t *ql, const int64_t *qh,
                  const int64_t *kl, const int64_t *kh, int device, void *stream);

/* ntt_cuda.ntt (ntt.cpp:166-188, K.cu:236-342): forward negacyclic NTT, natural in -> bit-reversed out.
 * ntt_cuda.enter_ntt (ntt.cpp:191-216, K.cu:349-423) when Rs != NULL: mont_enter(Rs) first. */
int lf_ntt(int64_t *a, int batch, int rows, int logN, const int64_t *psi_br, const double *psi_dp,
           const int64_t *q_host, const int64_t *Rs, int flags, const int64_t *_2q, const int64_t *ql,
           const int64_t *qh, const int64_t *kl, const int64_t *kh, int device, void *stream);

/* lf_ntt through a caller-provided WORKSPACE (exact transforms; same result words as lf_ntt, `a` in place).  The two passes
 * of a logN >= 13 transform hand every word over through HBM; with a workspace the column pass leaves the words of the
 * fp64-class limbs (primes below 2^41: lazy words below 2^42) there as 6-byte planes instead of 8-byte words and the tiled
 * pass reads those — 12.5 % fewer bytes per pass; operands outside [0, 2q), whose words can be anything, travel with a third
 * plane and a flag per column wave, so the reference's result on ANY int64 input is reproduced as by lf_ntt.
 * ws: lf_ntt_ws_words(batch, rows, logN) words of device memory, 16-byte aligned, contents irrelevant before and scratch
 * after; it must not be used by another stream while the call runs.  ws = NULL, logN <= 12 or > 17: lf_ntt.  The reference's
 * ntt (ntt.cpp:421-437) allocates nothing because it makes logN passes in place; the workspace is this design's price for
 * making two.  LF_NTT_RELAXED: LF_ERR_ARG. */
int64_t lf_ntt_ws_words(int batch, int rows, int logN);
int lf_ntt_ws(int64_t *a, int64_t *ws, int batch, int rows, int logN, const int64_t *psi_br, const double *psi_dp,
              const int64_t *q_host, const int64_t *Rs, int flags, const int64_t *ql, const int64_t *qh, const int64_t *kl,
              const int64_t *kh, int device, void *stream);
/* The inverse chains the same way (lf_intt with tail 0 .. 3 = intt / intt_exit / intt_exit_reduce / intt_exit_reduce_signed): the
 * tiled pass comes first and writes the workspace, the column pass with the chain tail reads it; a tile that meets an operand
 * outside [0, 2q) ships the third plane and raises its flag.  Same words as lf_intt on any input; same `ws` rules as lf_ntt_ws. */
int lf_intt_ws(int64_t *a, int64_t *ws, int batch, int rows, int logN, const int64_t *ipsi_br, const double *ipsi_dp,
               const int64_t *q_host, const int64_t *Ninv, int tail, int flags, const int64_t *ql, const int64_t *qh,
               const int64_t *kl, const int64_t *kh, int device, void *stream);
/* (measurement, as lf_ntt_pass: one of the two launches of lf_ntt_ws; which = 1 reads `a` and writes `ws`, 2 the reverse) */
int lf_ntt_pass_ws(int64_t *a, int64_t *ws, int batch, int rows, int logN, const int64_t *psi_br, const double *psi_dp,
                   const int64_t *q_host, const int64_t *Rs, int flags, int which, const int64_t *ql, const int64_t *qh,
                   const int64_t *kl, const int64_t *kh, int device, void *stream);

/* Measurement entry (not one of the reference's ops; the engine never calls it): launch exactly ONE of the two
 * pass kernels of a two-pass forward transform (logN >= 13) with the grid it has inside lf_ntt —
 * which = 1: the column pass, 2: the tiled pass — so that bench.py / tools can time the dominant kernel alone
 * with HIP events.  `a` is scratch afterwards (half a transform).  lf_ntt itself has no knobs. */
int lf_ntt_pass(int64_t *a, int batch, int rows, int logN, const int64_t *psi_br, const double *psi_dp,
                const int64_t *q_host, const int64_t *Rs, int flags, int which, const int64_t *ql, const int64_t *qh,
                const int64_t *kl, const int64_t *kh, int device, void *stream);

/* ntt_cuda.intt / intt_exit / intt_exit_reduce / intt_exit_reduce_signed
 * (ntt.cpp:219-345, K.cu:433-548, 709-973): inverse NTT, bit-reversed in -> natural out, then
 * x Ninv (= N^-1 * R mod q); `tail` selects the fused chain:
 *   0 intt, 1 + mont_redc, 2 + reduce (canonical [0,q)), 3 + make_signed.
 * LF_NTT_RELAXED requires tail >= 2 (whose outputs are canonical anyway). */
int lf_intt(int64_t *a, int batch, int rows, int logN, const int64_t *ipsi_br, const double *ipsi_dp,
            const int64_t *q_host, const int64_t *Ninv, int tail, int flags, const int64_t *_2q,
            const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh, int device, void *stream);

/* lf_intt of an element-wise PRODUCT, formed as the first pass reads its tiles (no product tensor in HBM):
 *   dst[p] = intt(a[p] * b[p]),  polynomial p of `a` / `b` at a + p * a_stride / b + p * b_stride (words), dst [batch][rows][N].
 * cc_mult's third tensor component x1 * y1 (ckks_engine.py:1099-1101, 1129) enters the key switch this way.  Requires
 * LF_NTT_RELAXED (tail >= 2) and logN >= 13; with LF_NTT_PLAIN the fp64-class limbs hold plain residues and get a plain
 * product, integer-class limbs Montgomery-form words (below 2q) and the REDC62 product — lf_tensor's d2 (plain = 1). */
int lf_intt_mul(int64_t *dst, const int64_t *a, int64_t a_stride, const int64_t *b, int64_t b_stride, int batch, int rows, int logN,
                const int64_t *ipsi_br, const double *ipsi_dp, const int64_t *q_host, const int64_t *Ninv, int tail, int flags,
                const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh, int device, void *stream);

/* Not one of the reference's ops: `count` (<= 8) rows of N words, each anywhere in device memory (16-byte aligned; HOST array
 * of device pointers), into the consecutive rows of dst — one launch.  The engine stages the dropped limb's rows of all
 * operands of a rescale with it before they fan out to the other ranks (ckks_engine.py:999-1011 stages them through the host). */
int lf_gather_rows(const int64_t *const *src, int64_t *dst, int count, int64_t N, int device, void *stream);

/* Galois automorphism of coefficient-domain rows (reference: encdec.py:224-270 `rotate`/`conjugate`,
 * done there with torch advanced indexing): dst[i][(p*n mod 2N) mod N] = +/- a[i][n], sign - iff
 * (p*n mod 2N) >= N.  If _2q != NULL the reference's follow-up make_unsigned + reduce_2q
 * (ckks_engine.py:1198-1200) is fused: the stored value is canonical in [0, q). */
int lf_galois(const int64_t *a, int64_t *dst, int rows, int logN, int64_t p, const int64_t *_2q,
              int device, void *stream);

/* ---- engine-level fused ops -------------------------------------------------------------------
 * Each replaces a run of ntt_cuda calls + torch elementwise ops issued by the reference's Python
 * engine (ckks_engine.py = src/liberate/fhe/ckks_engine.py); arithmetic is op-for-op the reference's. */

/* ckks_engine.rescale body (ckks_engine.py:1017-1041): out[i] = reduce_q(REDC((in[i] - row0) * scales[i])
 * + [row0 > round_at]); `in` points at the first surviving row, constants are those of the surviving rows. */
int lf_rescale(const int64_t *in, const int64_t *row0, int64_t *out, int rows, int64_t N, const int64_t *scales,
               int64_t round_at, const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh,
               int device, void *stream);

/* cc_mult's tensor product (ckks_engine.py:1095-1101): d0 = x0*y0, d1 = x0*y1 (+) x1*y0, d2 = x1*y1 (REDC, lazy).
 * plain = 1 (operands from lf_ntt with LF_NTT_RELAXED|LF_NTT_PLAIN): limbs with a prime below 2^41 hold plain
 * canonical residues and get plain fp64 products (canonical outputs); other limbs as above. */
int lf_tensor(const int64_t *x0, const int64_t *x1, const int64_t *y0, const int64_t *y1, int64_t *d0, int64_t *d1,
              int64_t *d2, int rows, int64_t N, int plain, const int64_t *ql, const int64_t *qh, const int64_t *kl,
              const int64_t *kh, int device, void *stream);

/* pre_extend (ckks_engine.py:654-705) for all local key-switch digits at once: mixed-radix (Garner) digits.
 * desc[p] = {row_start, alpha, y_off, l_off} (int64 x4); tab holds Y_scalar / L_scalar (ntt_context.py:328-345). */
int lf_ks_digits(const int64_t *a, int64_t *state, int nparts, const int64_t *desc, const int64_t *tab, int64_t N,
                 const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh, int device, void *stream);

/* extend (ckks_engine.py:707-743) of every digit to every local target row: ext[p][r] in Montgomery form.
 * desc[p] = {row_start, alpha, e_off} (int64 x3); E[e_off + i*rows + r] = R^2 (i = 0) or L_{i-1} R^2 mod q_r. */
int lf_ks_extend(const int64_t *state, int64_t *ext, int nparts, int rows, int64_t N, const int64_t *desc,
                 const int64_t *E, const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh,
                 int device, void *stream);

/* switcher_later_part's two mont_mults + the sum over digits (ckks_engine.py:931-934, 832-840); the key is
 * addressed as ksk[p*part_stride + comp*comp_stride + (row_off + r)*N + j], comp 0 = b, 1 = a. */
int lf_ks_inner(const int64_t *ext, const int64_t *ksk, int64_t part_stride, int64_t comp_stride, int64_t row_off,
                int64_t *s0, int64_t *s1, int nparts, int rows, int64_t N, const int64_t *ql, const int64_t *qh,
                const int64_t *kl, const int64_t *kh, int device, void *stream);

/* division by P = prod(special primes) (ckks_engine.py:850-901) on canonical coefficient rows s[ell+K][N];
 * PiR[P_ind][row] = P_j^-1 R mod q_row ([K][ell+K], specials last-first); optional addend:
 * out = reduce_q(result + addend) (relinearize 1135-1140 / switch_key 952-953).
 * PiP (optional, may be NULL): the same table as plain residues P_j^-1 mod q_row in doubles; when given,
 * rows with a prime below 2^41 take the fp64 path (identical canonical output). */
int lf_ks_moddown(const int64_t *s, int64_t *out, const int64_t *addend, int ell, int K, int64_t N,
                  const int64_t *PiR, const double *PiP, const int64_t *Rs, const int64_t *ql, const int64_t *qh, const int64_t *kl,
                  const int64_t *kh, int device, void *stream);

/* Fused key-switch core for two-pass ring degrees (logN >= 13): extend + NTT + inner product with the key +
 * sum over digits + inverse NTT to canonical coefficients, i.e. lf_ks_extend -> lf_ntt -> lf_ks_inner ->
 * lf_intt(tail 2) (ckks_engine.py:707-743, 919, 931-934, 832-848) without materialising the extended digits.
 *   state      [*, N] Garner digits in storage order (output of lf_ks_digits, gathered)
 *   desc, E    as for lf_ks_extend (alpha field: bit 8 set = the digit's words exceed 53 bits, i.e. a digit of
 *              60-bit primes; lf_ks_extend ignores the flag); Ed = the same constants as PLAIN residues in doubles
 *              (Ed[e_off + i*rows + r] = L_{i-1} mod q_r, i = 0: 1.0) for the fp64 class
 *   ksk        key, addressed as in lf_ks_inner
 *   tmp        scratch [nparts][rows][N]; s out [2][rows][N]
 *   q_host     HOST primes of the `rows` limbs (required: selects the arithmetic class per limb) */
int lf_ks_core(const int64_t *state, int nparts, int rows, int logN, const int64_t *desc, const int64_t *E,
               const double *Ed, const int64_t *ksk, int64_t part_stride, int64_t comp_stride, int64_t row_off,
               int key_format, int64_t *tmp, int64_t *s, const int64_t *psi_br, const double *psi_dp,
               const int64_t *ipsi_br, const double *ipsi_dp, const int64_t *Ninv, const int64_t *q_host, const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh, int device,
               void *stream);

/* Key formats of the fused key-switch entries (`key_format`).  The inner product with the key is the one launch of a key
 * switch that runs at the HBM rate, and most of what it reads is the key (gold: 450 of 687 MB), read once per call and never
 * modified: a binding may therefore keep a second, smaller copy of a key for these entries.
 *   LF_KEY_RAW     the reference's layout: 64-bit words, ksk[p*part_stride + comp*comp_stride + (row_off + r)*N + j];
 *   LF_KEY_PLANES  what lf_key_planes writes: same strides and row slots, but for a row r whose prime is below 2^41 the
 *                  slot of component 0 holds N / 2 groups of 16 bytes { lo32 b[j], lo32 b[j+1], lo32 a[j], lo32 a[j+1] } and
 *                  the first 4 N bytes of the slot of component 1 hold N / 2 groups of 8 bytes { hi16 b[j], hi16 b[j+1],
 *                  hi16 a[j], hi16 a[j+1] } (j even) of the CANONICAL residues of both components: 12 N instead of 16 N
 *                  bytes, and one 16-byte + one 8-byte load per thread and digit instead of two 16-byte ones.  Rows of
 *                  larger primes are raw words in their own slots.  16-byte aligned base and strides.
 * The sums are the same residues either way, so every output word of the entries below is identical.
 * lf_key_planes converts the `rows` rows of ONE key part (src_b / src_a: its two components, any lazy / signed-lazy words;
 * dst_b / dst_a: the part's two slots in a tensor of the raw key's shape, distinct from the sources; ql / qh: DEVICE 31-bit
 * halves of the rows' primes as everywhere in this header); a key is converted part by part. */
#define LF_KEY_RAW 0
#define LF_KEY_PLANES 1
int lf_key_planes(const int64_t *src_b, const int64_t *src_a, int64_t *dst_b, int64_t *dst_a, int rows, int64_t N,
                  const int64_t *ql, const int64_t *qh, int device, void *stream);

/* lf_ks_digits(_galois) of `count` (<= 8) polynomials in one launch: a / state are HOST arrays of device pointers
 * (gal_pinv = 0: no Galois map). */
int lf_ks_digits_batch(const int64_t *const *a, int64_t *const *state, int count, int nparts, const int64_t *desc,
                       const int64_t *tab, int64_t N, int64_t gal_pinv, const int64_t *gal_2q, const int64_t *ql,
                       const int64_t *qh, const int64_t *kl, const int64_t *kh, int device, void *stream);

/* lf_ks_core for `nct` (1, 2 or 4) ciphertexts switched under the SAME key (a batch of rotations by one step,
 * config "rotate batched 64 ciphertexts"): every launch covers all of them and the inner product reads each key
 * word once for the whole batch.  state: nct digit states `state_stride` words apart; tmp scratch
 * [nct][nparts][rows][N]; s out [nct][2][rows][N].  Results equal nct calls of lf_ks_core. */
int lf_ks_core_batch(const int64_t *state, int64_t state_stride, int nct, int nparts, int rows, int logN, const int64_t *desc,
                     const int64_t *E, const double *Ed, const int64_t *ksk, int64_t part_stride, int64_t comp_stride,
                     int64_t row_off, int key_format, int64_t *tmp, int64_t *s, const int64_t *psi_br, const double *psi_dp,
                     const int64_t *ipsi_br, const double *ipsi_dp, const int64_t *Ninv, const int64_t *q_host, const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh, int device,
                     void *stream);

/* The two halves of lf_ks_core as separate calls (single ciphertext), so that a limb-sharded engine can start on the
 * digits that have already arrived while the others are still travelling over xGMI (ckks_engine.py:778-829 stages
 * every digit through the host before any extension starts):
 *   lf_ks_fwd   extension + forward NTT of `nparts` digits, descriptors desc[0 .. nparts); the caller offsets desc
 *               and tmp to the first digit of the group (desc + 3 * first, tmp + first * rows * N);
 *   lf_ks_tail  after the last group: inner product of ALL nparts digits in tmp with the key + inverse NTT (tmp is scratch
 *               afterwards: with two digits or more the sums' inverse transform passes through it).
 * lf_ks_fwd over all digits followed by lf_ks_tail == lf_ks_core. */
int lf_ks_fwd(const int64_t *state, int nparts, int rows, int logN, const int64_t *desc, const int64_t *E, const double *Ed,
              int64_t *tmp, const int64_t *psi_br, const double *psi_dp, const int64_t *q_host, const int64_t *ql,
              const int64_t *qh, const int64_t *kl, const int64_t *kh, int device, void *stream);
int lf_ks_tail(int nparts, int rows, int logN, const int64_t *ksk, int64_t part_stride, int64_t comp_stride, int64_t row_off,
               int key_format, int64_t *tmp, int64_t *s, const int64_t *ipsi_br, const double *ipsi_dp, const int64_t *Ninv,
               const int64_t *q_host, const int64_t *ql, const int64_t *qh, const int64_t *kl,
               const int64_t *kh, int device, void *stream);

/* Relinearisation inside cc_mult (ckks_engine.py:1095-1101, 1117-1151) without inverse transforms of d0 and d1:
 * dividing by P is linear and P * d vanishes modulo every special prime, so moddown(s) + d == moddown(s + P * d on the
 * ordinary rows).  lf_relin_core_batch / lf_relin_tail are lf_ks_core_batch / lf_ks_tail whose sums additionally receive,
 * in the NTT domain, on the first `ell` (ordinary) of the `rows` limbs,
 *     s[0] += P * (x0 * y0),    s[1] += P * (x0 * y1 + x1 * y0),
 * from x = [nct][4][ell][N] (x0, x1, y0, y1 as lf_rescale_ntt with LF_NTT_RELAXED | LF_NTT_PLAIN leaves them; stacks of
 * consecutive ciphertext pairs x_ct_stride words apart) and PR[r] = P * R mod q_r.  lf_ks_moddown_* of the result, with no
 * addend, is the relinearised ciphertext: the same canonical words as the reference's chain. */
int lf_relin_core_batch(const int64_t *state, int64_t state_stride, int nct, int nparts, int rows, int logN, const int64_t *desc,
                        const int64_t *E, const double *Ed, const int64_t *ksk, int64_t part_stride, int64_t comp_stride,
                        int64_t row_off, int key_format, int64_t *tmp, int64_t *s, const int64_t *psi_br, const double *psi_dp,
                        const int64_t *ipsi_br, const double *ipsi_dp, const int64_t *Ninv, const int64_t *x, int64_t x_ct_stride,
                        const int64_t *PR, int ell, const uint8_t *own, const int64_t *q_host,
                        const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh, int device, void *stream);
/* own (optional DEVICE table of `rows` bytes, may be NULL): own[r] = the digit (storage order) whose primes include limb
 * r, 255 for the special limbs.  The extension of a digit's mixed-radix form to one of its own primes is the residue it
 * was built from, i.e. the switched polynomial x1 * y1 itself: those (digit, limb) pairs are neither extended nor
 * transformed, and the inner product forms x1 * y1 from the stack in their place (35 of gold's 390 limb transforms).
 * lf_relin_fwd = lf_ks_fwd of digits first .. first + nparts - 1 with that table (desc / tmp are NOT offset by the caller). */
int lf_relin_fwd(const int64_t *state, int first, int nparts, int rows, int logN, const int64_t *desc, const int64_t *E,
                 const double *Ed, int64_t *tmp, const int64_t *psi_br, const double *psi_dp, const uint8_t *own,
                 const int64_t *q_host, const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh, int device,
                 void *stream);
int lf_relin_tail(int nparts, int rows, int logN, const int64_t *ksk, int64_t part_stride, int64_t comp_stride, int64_t row_off,
                  int key_format, int64_t *tmp, int64_t *s, const int64_t *ipsi_br, const double *ipsi_dp, const int64_t *Ninv,
                  const int64_t *x, const int64_t *PR, int ell, const uint8_t *own, const int64_t *q_host, const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh, int device,
                  void *stream);

/* Batched forms: `count` (<= 8) independent operand sets in ONE launch — the two components of a ciphertext, the
 * four polynomials cc_mult rescales.  The arrays of pointers are HOST arrays of device pointers; constants are
 * shared by all sets.  addend may be NULL, or hold NULL entries. */
int lf_rescale_batch(const int64_t *const *in, const int64_t *const *row0, int64_t *const *out, int count, int rows,
                     int64_t N, const int64_t *scales, int64_t round_at, const int64_t *ql, const int64_t *qh,
                     const int64_t *kl, const int64_t *kh, int device, void *stream);
int lf_ks_moddown_batch(const int64_t *const *s, int64_t *const *out, const int64_t *const *addend, int count, int ell, int K,
                        int64_t N, const int64_t *PiR, const double *PiP, const int64_t *Rs, int64_t gal_pinv,
                        const int64_t *gal_2q, const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh,
                        int device, void *stream);

/* lf_ks_moddown_batch with a caller-provided workspace `ws` of at least lf_ks_moddown_ws_words(count, ell, K, N)
 * int64 words: the elimination among the K special rows (ckks_engine.py:850-870) is evaluated once per coefficient
 * by a first launch instead of once per row chunk, and the fp64-class rows take the closed form
 * (s - sum_j p_j prod_{i<j} P_i) / P.  Same canonical outputs; s is not modified. */
int64_t lf_ks_moddown_ws_words(int count, int ell, int K, int64_t N);
int lf_ks_moddown_ws(const int64_t *const *s, int64_t *const *out, const int64_t *const *addend, int count, int ell, int K,
                     int64_t N, int64_t *ws, int64_t ws_words, const int64_t *PiR, const double *PiP, const int64_t *Rs,
                     int64_t gal_pinv, const int64_t *gal_2q, const int64_t *ql, const int64_t *qh, const int64_t *kl,
                     const int64_t *kh, int device, void *stream);
/* The same in ONE launch for K <= LF_MODDOWN_ONE_MAX_K special primes: every block eliminates the special rows for its own
 * coefficients (K = 2: one REDC product per coefficient and row chunk — cheaper than a launch).  The per-row constants that
 * lf_ks_moddown_ws writes behind the pivots on every call are level constants: lf_ks_moddown_consts writes them ONCE into a
 * workspace, lf_ks_moddown_one only reads them (the pivot part of the workspace is not used).  Same outputs. */
#define LF_MODDOWN_ONE_MAX_K 2
int lf_ks_moddown_consts(int64_t *ws, int64_t ws_words, int count, int ell, int K, int64_t N, const double *PiP, const int64_t *ql,
                         const int64_t *qh, const int64_t *kl, const int64_t *kh, int device, void *stream);
int lf_ks_moddown_one(const int64_t *const *s, int64_t *const *out, const int64_t *const *addend, int count, int ell, int K,
                      int64_t N, const int64_t *ws, int64_t ws_words, const int64_t *PiR, const double *PiP, const int64_t *Rs,
                      int64_t gal_pinv, const int64_t *gal_2q, const int64_t *ql, const int64_t *qh, const int64_t *kl,
                      const int64_t *kh, int device, void *stream);

/* Galois permutation in gather form, so that rotate / conjugate need no permutation pass of their own
 * (switch_key / rotate_single, ckks_engine.py:939-961, 1180-1206; encdec.py:224-270):
 *   lf_ks_digits_galois  = lf_ks_digits of a(X^p);  lf_ks_moddown_batch with gal_pinv != 0 adds addend(X^p).
 * gal_pinv = p^-1 mod 2N (0: no permutation); gal_2q != NULL: the permuted words are made canonical as
 * rotate_single does (make_unsigned + reduce_2q), NULL: they stay signed as conjugate leaves them. */
int lf_ks_digits_galois(const int64_t *a, int64_t *state, int nparts, const int64_t *desc, const int64_t *tab, int64_t N,
                        int64_t gal_pinv, const int64_t *gal_2q, const int64_t *ql, const int64_t *qh, const int64_t *kl,
                        const int64_t *kh, int device, void *stream);
int lf_galois_batch(const int64_t *const *a, int64_t *const *dst, int count, int rows, int logN, int64_t p,
                    const int64_t *_2q, int device, void *stream);

/* lf_intt_mul (relaxed, tail 2) followed by lf_ks_digits of its result in ONE launch behind the tiled pass: the column thread of
 * the last inverse pass takes the columns of all `alpha` limbs of its digit, runs the Garner step on the canonical words it holds
 * and stores the digit state — the coefficient-domain product is never written (cc_mult: ckks_engine.py:1099-1101, 1129, 654-705).
 * `scratch` [batch][rows][N] receives the tiled pass's output; `state` [batch][rows][N]; desc / tab = lf_ks_digits' tables for
 * `nparts` digits of at most `max_alpha` limbs.  Applies to two-pass ring degrees with max_alpha * 2^(logN - 12) <= 32 (silver,
 * bronze; <= 64 at logN 16 — gold — when batch >= 2); otherwise returns LF_ERR_ARG with nothing launched and the caller takes the two calls.  Same words in `state`. */
int lf_intt_mul_digits(int64_t *scratch, const int64_t *a, int64_t a_stride, const int64_t *b, int64_t b_stride, int batch, int rows,
                       int logN, int64_t *state, int nparts, int max_alpha, const int64_t *desc, const int64_t *tab,
                       const int64_t *ipsi_br, const double *ipsi_dp, const int64_t *q_host, const int64_t *Ninv, int flags,
                       const int64_t *ql, const int64_t *qh, const int64_t *kl, const int64_t *kh, int device, void *stream);

/* cc_mult's opening (ckks_engine.py:1085-1093): rescale `count` (<= 8) polynomials and transform them, i.e.
 * lf_rescale_batch(in, row0, {x + i*rows*N}, ...) followed by lf_ntt(x, count, ...) with the same constants.
 * For two-pass ring degrees (logN 13..16) the rescale is evaluated inside the first NTT pass: no launch and no
 * trip through HBM of its own; other degrees run the two steps one after the other.  Same results either way. */
int lf_rescale_ntt(const int64_t *const *in, const int64_t *const *row0, int count, int64_t *x, int rows, int logN,
                   const int64_t *scales, int64_t round_at, const int64_t *psi_br, const double *psi_dp,
                   const int64_t *q_host, const int64_t *Rs, int flags, const int64_t *_2q, const int64_t *ql,
                   const int64_t *qh, const int64_t *kl, const int64_t *kh, int device, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Whole engine ops behind one entry each (csrc/ckks_ops.hip).  The reference's Python engine issues ~250 extension calls
 * per cc_mult + relinearize (ckks_engine.py:1072-1151 over 654-961); a binding that wants the host side of an op to be ONE
 * native call fills an lf_ks_plan once per (device, level) — everything that does not change from call to call — and
 * hands the operands and the key per call.  The entries enqueue exactly the steps above (lf_rescale_ntt, lf_intt_mul,
 * lf_ks_digits(_galois), lf_relin_core_batch / lf_ks_core, lf_ks_moddown_ws) on `stream` and return the first failure.
 * They apply when every limb of the level lives on this device (no exchange between the digits and their extension).
 * `rows` = ell + K limbs, ordinary first: the per-row vectors and twiddle tables are those of lf_ks_core.
 * ---------------------------------------------------------------------------------------------- */
typedef struct lf_ks_plan {
    int32_t logN, ell, K, nparts;        /* ring degree, ordinary limbs at the op's level, special primes, digits */
    int32_t dig_nparts, device;          /* digits lf_ks_digits builds here (= nparts on one device) */
    int32_t max_nct;                     /* ciphertexts per batched call the scratch below is sized for (1, 2 or 4) */
    int32_t md_consts;                   /* `count` md_ws was primed for by lf_ks_moddown_consts (the op entries run
                                            lf_ks_moddown_one only for exactly that many polynomials: 2 x the ciphertexts
                                            of the call); 0 = never primed: every entry takes lf_ks_moddown_ws, which
                                            needs no preparation.  A zero-filled struct is therefore always safe. */
    int64_t round_at;                    /* cc_mult: rescale rounding threshold q_l / 2 (lf_rescale) */
    int64_t md_ws_words;
    const int64_t *ql, *qh, *kl, *kh, *_2q, *Rs, *Ninv;      /* device, [rows] */
    const int64_t *q_host;                                    /* HOST, [rows] */
    const int64_t *psi, *ipsi;                                /* compact twiddle tables of the rows */
    const double *psi_dp, *ipsi_dp;                           /* their auxiliary tables (lf_twiddle_dp) */
    const int64_t *dig_desc, *dig_tab;                        /* lf_ks_digits */
    const int64_t *ext_desc, *E;                              /* lf_ks_core */
    const double *Ed;
    const int64_t *PiR;                                       /* lf_ks_moddown */
    const double *PiP;
    const uint8_t *own;                                       /* lf_relin_*: may be NULL */
    const int64_t *rescale_scales, *PR;                       /* cc_mult only: [ell] q_l^-1 R and P R mod q_r */
    int64_t *state, *ext, *sum, *md_ws;                       /* scratch: max_nct x ([ell][N], [nparts][rows][N], [2][rows][N]); md_ws_words for 2 max_nct polynomials */
    int64_t *x4, *d2;                                         /* cc_mult only: max_nct x ([4][ell][N], [ell][N]) */
} lf_ks_plan;

/* ckks_engine.cc_mult(a, b, evk) with relinearisation, level l -> l + 1 (ckks_engine.py:1072-1151): in[0..3] = first
 * SURVIVING row of a.c0, a.c1, b.c0, b.c1 (HOST array of device pointers, as lf_rescale_batch), row0[0..3] their dropped
 * rows; the plan describes level l + 1; key addressed as in lf_ks_inner, in `key_format`; out0 / out1 [ell][N] canonical. */
int lf_cc_mult_evk(const lf_ks_plan *plan, const int64_t *const *in, const int64_t *const *row0, const int64_t *ksk,
                   int64_t part_stride, int64_t comp_stride, int64_t row_off, int key_format, int64_t *out0, int64_t *out1,
                   void *stream);

/* ckks_engine.switch_key / rotate_single / conjugate of a coefficient-domain ciphertext (c0, c1), [ell][N] each
 * (ckks_engine.py:939-961, 1180-1206, 1718-1734): out = (c0(X^p) + ks_0, ks_1) with ks = key switch of c1(X^p);
 * gal_pinv = p^-1 mod 2N (0: no automorphism), gal_canonical != 0: rotate_single's make_unsigned + reduce_2q. */
int lf_switch_key(const lf_ks_plan *plan, const int64_t *c0, const int64_t *c1, int64_t gal_pinv, int gal_canonical,
                  const int64_t *ksk, int64_t part_stride, int64_t comp_stride, int64_t row_off, int key_format, int64_t *out0,
                  int64_t *out1, void *stream);

/* The same ops for nct = 1, 2 or 4 ciphertexts under ONE key (plan->max_nct >= nct): every launch covers all of them and the
 * inner product reads each key word once for the group (lf_ks_core_batch / lf_relin_core_batch).  c0 / c1 / out0 / out1: HOST
 * arrays of nct device pointers; in / row0: 4 per ciphertext pair, in the order of lf_cc_mult_evk.  Results equal nct single calls. */
int lf_switch_key_batch(const lf_ks_plan *plan, int nct, const int64_t *const *c0, const int64_t *const *c1, int64_t gal_pinv,
                        int gal_canonical, const int64_t *ksk, int64_t part_stride, int64_t comp_stride, int64_t row_off,
                        int key_format, int64_t *const *out0, int64_t *const *out1, void *stream);
int lf_cc_mult_evk_batch(const lf_ks_plan *plan, int nct, const int64_t *const *in, const int64_t *const *row0, const int64_t *ksk,
                         int64_t part_stride, int64_t comp_stride, int64_t row_off, int key_format, int64_t *const *out0,
                         int64_t *const *out1, void *stream);

/* The halves of an op around the digit exchange of a limb-sharded engine (one process per GPU; the reference gathers every
 * digit on every GPU through the host before it extends any, ckks_engine.py:778-829).  The plan describes THIS rank's rows
 * at the level (dig_nparts = the digits it owns, nparts = all digits, state = its own digit rows):
 *   lf_cc_mult_evk_pre / lf_switch_key_pre   everything up to the digits this rank owns, into plan->state;
 *   lf_ks_plan_fwd                           extension + forward NTT of digits first .. first + count - 1 of the gathered
 *                                            storage-order buffer `digits` (own digits while the others travel, foreign runs
 *                                            after the wait); relin != 0: inside cc_mult (own-limb pairs skipped);
 *   lf_cc_mult_evk_post / lf_switch_key_post inner product over ALL digits + inverse NTT + mod-down (+ c0(X^p)).
 * pre, fwd over every digit, post == lf_cc_mult_evk / lf_switch_key.
 * `which` (bit mask, 3 = the whole half) separates the launches whose ADDRESSES change from call to call from those that only
 * touch the plan's scratch, tables and the key — the second kind can be captured once into a HIP graph and replayed (the
 * engine does: a rank of a sharded gold cc_mult enqueues in ~55 us of host time instead of 140, profiles/r05_host_overhead.txt):
 *   lf_cc_mult_evk_pre   1 = the launch that reads in / row0 (rescale + column pass), 2 = the rest (tiled pass, x1 * y1,
 *                        inverse NTT, digits);  lf_switch_key_pre is one launch, it reads c1;
 *   lf_*_post            1 = inner product + inverse NTT, 2 = the mod-down (reads c0, writes out0 / out1). */
int lf_cc_mult_evk_pre(const lf_ks_plan *plan, const int64_t *const *in, const int64_t *const *row0, int which, void *stream);
int lf_switch_key_pre(const lf_ks_plan *plan, const int64_t *c1, int64_t gal_pinv, int gal_canonical, void *stream);
int lf_ks_plan_fwd(const lf_ks_plan *plan, const int64_t *digits, int first, int count, int relin, void *stream);
int lf_cc_mult_evk_post(const lf_ks_plan *plan, const int64_t *ksk, int64_t part_stride, int64_t comp_stride, int64_t row_off,
                        int key_format, int64_t *out0, int64_t *out1, int which, void *stream);
int lf_switch_key_post(const lf_ks_plan *plan, const int64_t *c0, int64_t gal_pinv, int gal_canonical, const int64_t *ksk,
                       int64_t part_stride, int64_t comp_stride, int64_t row_off, int key_format, int64_t *out0, int64_t *out1,
                       int which, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Samplers: the reference's csprng extensions, src/liberate/csprng/.
 * A ChaCha20 state is 16 int64 words holding 32-bit values (csprng.py:124-160); words 12/13 are
 * the 64-bit block counter, advanced by `step` after every draw.  All tables below are DEVICE
 * pointers except q_host / btree_host, which the reference also takes as host pointers.
 * Row counts need not be multiples of the block size (the reference silently drops the tail).
 * ---------------------------------------------------------------------------------------------- */

/* chacha20_cuda.chacha20 (chacha20.cpp:17-40, chacha20_cuda_kernel.cu:10-46): dest[i] = block(states[i])
 * for n states [n][16]; states' counters advance. */
int lf_chacha20(int64_t *states, int64_t *dest, int64_t n, uint64_t step, int device, void *stream);

/* randint_cuda.randint_fast (randint.cpp:38-52, randint_cuda_kernel.cu:22-101): fused ChaCha20 +
 * floor(q_c * X / 2^128) + shift, four samples per state; states [channels][L][16] -> dst [channels][4L]. */
int lf_randint_fast(int64_t *states, int64_t *dst, int channels, int64_t L, const uint64_t *q_host, int64_t shift,
                    uint64_t step, int device, void *stream);

/* randint_cuda.randint (randint.cpp:21-33, randint_cuda_kernel.cu:108-152): the same map in place on
 * random words rand_bytes [channels][n][16]; word 4j of each row receives the sample of words 4j..4j+3. */
int lf_randint(int64_t *rand_bytes, int channels, int64_t n, const uint64_t *q_host, int device, void *stream);

/* discrete_gaussian_cuda.discrete_gaussian_fast (discrete_gaussian_cuda_kernel.cu:28-109, 186-215): fused
 * ChaCha20 + CDT binary-tree walk, four samples per state; states [n][16] -> dst [4n].
 * btree_host: btree_size low words in level order, then btree_size high words
 * (discrete_gaussian_sampler.py:96-118); 2*btree_size <= 128. */
int lf_discrete_gaussian_fast(int64_t *states, int64_t *dst, int64_t n, const uint64_t *btree_host, int btree_size,
                              int depth, uint64_t step, int device, void *stream);

/* discrete_gaussian_cuda.discrete_gaussian (discrete_gaussian_cuda_kernel.cu:118-168, 222-240): in place on
 * rand_bytes [n][16]. */
int lf_discrete_gaussian(int64_t *rand_bytes, int64_t n, const uint64_t *btree_host, int btree_size, int depth,
                         int device, void *stream);

/* randround_cuda.randround (randround_cuda_kernel.cu:8-56): rand_bytes[i] = sign(c) * (floor|c| +
 * [rand_bytes[i] < rn(frac|c| * 2^32)]), c = coef[i]; rand_bytes holds 32-bit random words. */
int lf_randround(const double *coef, int64_t *rand_bytes, int64_t n, int device, void *stream);

/* ------------------------------------------------------------------------------------------------
 * The reference's 30-bit / int32 word mode of the same 15 functions (ckks_context.py:213-216 buffer_bit_length = 30:
 * R = 2^30, 15-bit halves ql / qh / kl / kh, 28-bit message primes; K.cu:141, 223, 339 dispatch the kernel templates for
 * int32 as well).  No preset, test or example of the reference selects it; it is served for the completeness of the
 * boundary (csrc/ckks_w30.hip: one plain launch per step of the reference's own chain, its exact lazy words), the fused
 * engine entries above are 62-bit only.  Arguments as their lf_* counterparts, words and per-row vectors int32;
 * psi_br / ipsi_br = the compact [rows][N] tables in Montgomery form (R = 2^30).
 *   mont_mult, mont_enter, mont_redc (ntt.cpp:120-163, 248-263); reduce_2q, make_signed, make_unsigned, tile_unsigned,
 *   mont_add, mont_sub (347-419); lf30_ntt: Rs = NULL ntt, Rs != NULL enter_ntt (166-216); lf30_intt: tail 0 intt, 1 intt_exit,
 *   2 intt_exit_reduce, 3 intt_exit_reduce_signed (219-345).
 * ---------------------------------------------------------------------------------------------- */
int lf30_mont_mult(const int32_t *a, const int32_t *b, int32_t *c, int rows, int64_t N, const int32_t *ql, const int32_t *qh,
                   const int32_t *kl, const int32_t *kh, int device, void *stream);
int lf30_mont_enter(int32_t *a, const int32_t *Rs, int rows, int64_t N, const int32_t *ql, const int32_t *qh, const int32_t *kl,
                    const int32_t *kh, int device, void *stream);
int lf30_mont_redc(int32_t *a, int rows, int64_t N, const int32_t *ql, const int32_t *qh, const int32_t *kl, const int32_t *kh,
                   int device, void *stream);
int lf30_reduce_2q(int32_t *a, int rows, int64_t N, const int32_t *_2q, int device, void *stream);
int lf30_make_signed(int32_t *a, int rows, int64_t N, const int32_t *_2q, int device, void *stream);
int lf30_make_unsigned(int32_t *a, int rows, int64_t N, const int32_t *_2q, int device, void *stream);
int lf30_tile_unsigned(const int32_t *a, int32_t *dst, int rows, int64_t N, const int32_t *_2q, int device, void *stream);
int lf30_mont_add(const int32_t *a, const int32_t *b, int32_t *c, int rows, int64_t N, const int32_t *_2q, int device, void *stream);
int lf30_mont_sub(const int32_t *a, const int32_t *b, int32_t *c, int rows, int64_t N, const int32_t *_2q, int device, void *stream);
int lf30_ntt(int32_t *a, int batch, int rows, int logN, const int32_t *psi_br, const int32_t *Rs, const int32_t *_2q,
             const int32_t *ql, const int32_t *qh, const int32_t *kl, const int32_t *kh, int device, void *stream);
int lf30_intt(int32_t *a, int batch, int rows, int logN, const int32_t *ipsi_br, const int32_t *Ninv, int tail, const int32_t *_2q,
              const int32_t *ql, const int32_t *qh, const int32_t *kl, const int32_t *kh, int device, void *stream);

#ifdef __cplusplus
}
#endif
#endif
