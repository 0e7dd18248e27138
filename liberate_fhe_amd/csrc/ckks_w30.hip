// ckks_w30.hip — the reference's 30-bit / int32 word mode of the ntt_cuda surface: lf30_*.
//
// The reference's kernels are templates over scalar_t and are dispatched for int32 as well as int64
// (K.cu:141, 223, 339 AT_DISPATCH_INTEGRAL_TYPES; ckks_context.py:213-216 buffer_bit_length = 30: R = 2^30, 15-bit halves,
// 28-bit message primes).  No preset, test or example of the reference selects that mode; it is served here for the
// completeness of the 15-function boundary, not for speed: every op is one plain launch per step of the reference's own
// chain (a transform = one launch per stage over a compact twiddle table, then the chain's elementwise tail), in the
// reference's exact lazy arithmetic — the literal half-word formulas on int32, wrapping like the CUDA device code does.
// The engine's fused ops (lf_ks_*, lf_cc_mult_evk, ..) exist for the 62-bit mode only.
#include "../../include/ckks_hip.h"
#include "ckks_common.h"

namespace {

typedef int32_t w30;
#define H30 15
#define LB30 ((w30)((1 << H30) - 1))
#define FB30 ((w30)((1 << 30) - 1))

// K.cu:12-59 with scalar_t = int32 (nbits 30, half 15); unsigned arithmetic where the CUDA code relies on wrap-around
__device__ __forceinline__ w30 mm30(w30 a, w30 b, w30 ql, w30 qh, w30 kl, w30 kh) {
    typedef uint32_t u;
    const w30 al = a & LB30, ah = a >> H30;
    const w30 bl = b & LB30, bh = b >> H30;
    const w30 alpha = (w30)((u)ah * (u)bh);
    const w30 beta = (w30)((u)ah * (u)bl + (u)al * (u)bh);
    const w30 gamma = (w30)((u)al * (u)bl);
    const w30 gammal = gamma & LB30, gammah = gamma >> H30;
    const w30 betal = beta & LB30, betah = beta >> H30;
    u upper = (u)gammal * (u)kh;
    upper = upper + (u)(gammah + betal) * (u)kl;
    upper = upper << H30;
    w30 s = (w30)(upper + (u)gammal * (u)kl);
    s = s & FB30;
    const w30 sl = s & LB30, sh = s >> H30;
    const w30 sqb = (w30)((u)sh * (u)ql + (u)sl * (u)qh);
    const w30 sqbl = sqb & LB30, sqbh = sqb >> H30;
    w30 carry = (w30)((u)gamma + (u)sl * (u)ql) >> H30;
    carry = (w30)((u)carry + (u)betal + (u)sqbl) >> H30;
    return (w30)((u)alpha + (u)betah + (u)sqbh + (u)carry + (u)sh * (u)qh);
}

// K.cu:587-606
__device__ __forceinline__ w30 redc30(w30 x, w30 ql, w30 qh, w30 kl, w30 kh) {
    typedef uint32_t u;
    const w30 xl = x & LB30, xh = x >> H30;
    const w30 xkb = (w30)((u)xh * (u)kl + (u)xl * (u)kh);
    w30 s = (w30)(((u)xkb << H30) + (u)xl * (u)kl);
    s = s & FB30;
    const w30 sl = s & LB30, sh = s >> H30;
    const w30 sqb = (w30)((u)sh * (u)ql + (u)sl * (u)qh);
    const w30 sqbl = sqb & LB30, sqbh = sqb >> H30;
    w30 carry = (w30)((u)x + (u)sl * (u)ql) >> H30;
    carry = (w30)((u)carry + (u)sqbl) >> H30;
    return (w30)((u)sqbh + (u)carry + (u)sh * (u)qh);
}

enum { W_MULT, W_ENTER, W_REDC, W_REDUCE, W_SIGNED, W_UNSIGNED, W_TILE, W_ADD, W_SUB };

// one word per thread; grid = (ceil(N / 256), rows).  b / c / per-row vectors as the op needs them.
template <int OP>
__global__ void __launch_bounds__(256) ew30_kernel(const w30 *a, const w30 *b, w30 *c, int64_t N,   /* c may be a: in-place ops */
                                                   const w30 *__restrict__ v0, const w30 *__restrict__ ql, const w30 *__restrict__ qh,
                                                   const w30 *__restrict__ kl, const w30 *__restrict__ kh) {
    const int r = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const int64_t at = (int64_t)r * N + j;
    if (OP == W_MULT) c[at] = mm30(a[at], b[at], ql[r], qh[r], kl[r], kh[r]);
    if (OP == W_ENTER) c[at] = mm30(a[at], v0[r], ql[r], qh[r], kl[r], kh[r]);       // v0 = Rs (or Ninv R)
    if (OP == W_REDC) c[at] = redc30(a[at], ql[r], qh[r], kl[r], kh[r]);
    if (OP == W_REDUCE) { const w30 q = v0[r] >> 1, v = a[at]; c[at] = v < q ? v : v - q; }            // K.cu:664-680 (ONE subtraction of q)
    if (OP == W_SIGNED) { const w30 q = v0[r] >> 1, v = a[at]; c[at] = v <= (q >> 1) ? v : v - q; }    // K.cu:682-699
    if (OP == W_UNSIGNED) c[at] = a[at] + (v0[r] >> 1);                                                // K.cu:980-995
    if (OP == W_TILE) c[at] = a[j] + (v0[r] >> 1);                                                     // K.cu:997-1014
    if (OP == W_ADD) { const w30 s = a[at] + b[at]; c[at] = s < v0[r] ? s : s - v0[r]; }              // K.cu:1016-1036
    if (OP == W_SUB) { const w30 s = a[at] + v0[r] - b[at]; c[at] = s < v0[r] ? s : s - v0[r]; }      // K.cu:1038-1058
}

// one stage of a transform, one butterfly per thread; grid = (N / 512, batch * rows).  Indices as the reference paints them
// (ckks_context.py:89-142): forward stage m = 2^s, t = N / 2m: block i < m pairs j, j + t for j in [2 i t, 2 i t + t),
// twiddle psi_br[m + i]; inverse stage t = 2^s, h = N / 2t: block i < h, twiddle ipsi_br[h + i].
template <bool INV>
__global__ void __launch_bounds__(256) stage30_kernel(w30 *__restrict__ a, const w30 *__restrict__ tw, int rows, int logN, int logt,
                                                      const w30 *__restrict__ _2q, const w30 *__restrict__ ql,
                                                      const w30 *__restrict__ qh, const w30 *__restrict__ kl,
                                                      const w30 *__restrict__ kh) {
    const int r = blockIdx.y % rows;
    const int64_t N = (int64_t)1 << logN;
    const int64_t bf = (int64_t)blockIdx.x * 256 + threadIdx.x;     // butterfly index of the row
    if (bf >= N / 2) return;
    const int64_t t = (int64_t)1 << logt;
    const int64_t i = bf >> logt, off = bf & (t - 1);
    const int64_t j = 2 * i * t + off;
    const int64_t blocks = (N / 2) >> logt;                          // m (forward) / h (inverse)
    w30 *x = a + (int64_t)blockIdx.y * N;
    const w30 S = tw[(int64_t)r * N + blocks + i];
    const w30 q2 = _2q[r];
    const w30 U = x[j], V = x[j + t];
    if (!INV) {   // K.cu:260-274
        const w30 W = mm30(S, V, ql[r], qh[r], kl[r], kh[r]);
        const w30 p = U + W, d = U + q2 - W;
        x[j] = p < q2 ? p : p - q2;
        x[j + t] = d < q2 ? d : d - q2;
    } else {      // K.cu:457-472
        const w30 d = U + q2 - V;
        const w30 O = d < q2 ? d : d - q2;
        x[j + t] = mm30(S, O, ql[r], qh[r], kl[r], kh[r]);
        const w30 p = U + V;
        x[j] = p < q2 ? p : p - q2;
    }
}

inline bool bad_ew(const void *a, int rows, int64_t N) { return !a || rows < 0 || N < 1; }

template <int OP>
int launch_ew(const w30 *a, const w30 *b, w30 *c, int rows, int64_t N, const w30 *v0, const w30 *ql, const w30 *qh, const w30 *kl,
              const w30 *kh, int device, void *stream) {
    if (rows == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    hipLaunchKernelGGL(ew30_kernel<OP>, dim3((unsigned)((N + 255) / 256), (unsigned)rows), dim3(256), 0, (hipStream_t)stream, a, b, c, N,
                       v0, ql, qh, kl, kh);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int lf30_mont_mult(const int32_t *a, const int32_t *b, int32_t *c, int rows, int64_t N, const int32_t *ql, const int32_t *qh,
                   const int32_t *kl, const int32_t *kh, int device, void *stream) {
    if (bad_ew(a, rows, N) || !b || !c || !ql || !qh || !kl || !kh) return LF_ERR_ARG;
    return launch_ew<W_MULT>(a, b, c, rows, N, nullptr, ql, qh, kl, kh, device, stream);
}
int lf30_mont_enter(int32_t *a, const int32_t *Rs, int rows, int64_t N, const int32_t *ql, const int32_t *qh, const int32_t *kl,
                    const int32_t *kh, int device, void *stream) {
    if (bad_ew(a, rows, N) || !Rs || !ql || !qh || !kl || !kh) return LF_ERR_ARG;
    return launch_ew<W_ENTER>(a, nullptr, a, rows, N, Rs, ql, qh, kl, kh, device, stream);
}
int lf30_mont_redc(int32_t *a, int rows, int64_t N, const int32_t *ql, const int32_t *qh, const int32_t *kl, const int32_t *kh,
                   int device, void *stream) {
    if (bad_ew(a, rows, N) || !ql || !qh || !kl || !kh) return LF_ERR_ARG;
    return launch_ew<W_REDC>(a, nullptr, a, rows, N, nullptr, ql, qh, kl, kh, device, stream);
}
int lf30_reduce_2q(int32_t *a, int rows, int64_t N, const int32_t *_2q, int device, void *stream) {
    if (bad_ew(a, rows, N) || !_2q) return LF_ERR_ARG;
    return launch_ew<W_REDUCE>(a, nullptr, a, rows, N, _2q, nullptr, nullptr, nullptr, nullptr, device, stream);
}
int lf30_make_signed(int32_t *a, int rows, int64_t N, const int32_t *_2q, int device, void *stream) {
    if (bad_ew(a, rows, N) || !_2q) return LF_ERR_ARG;
    return launch_ew<W_SIGNED>(a, nullptr, a, rows, N, _2q, nullptr, nullptr, nullptr, nullptr, device, stream);
}
int lf30_make_unsigned(int32_t *a, int rows, int64_t N, const int32_t *_2q, int device, void *stream) {
    if (bad_ew(a, rows, N) || !_2q) return LF_ERR_ARG;
    return launch_ew<W_UNSIGNED>(a, nullptr, a, rows, N, _2q, nullptr, nullptr, nullptr, nullptr, device, stream);
}
int lf30_tile_unsigned(const int32_t *a, int32_t *dst, int rows, int64_t N, const int32_t *_2q, int device, void *stream) {
    if (bad_ew(a, rows, N) || !dst || !_2q) return LF_ERR_ARG;
    return launch_ew<W_TILE>(a, nullptr, dst, rows, N, _2q, nullptr, nullptr, nullptr, nullptr, device, stream);
}
int lf30_mont_add(const int32_t *a, const int32_t *b, int32_t *c, int rows, int64_t N, const int32_t *_2q, int device, void *stream) {
    if (bad_ew(a, rows, N) || !b || !c || !_2q) return LF_ERR_ARG;
    return launch_ew<W_ADD>(a, b, c, rows, N, _2q, nullptr, nullptr, nullptr, nullptr, device, stream);
}
int lf30_mont_sub(const int32_t *a, const int32_t *b, int32_t *c, int rows, int64_t N, const int32_t *_2q, int device, void *stream) {
    if (bad_ew(a, rows, N) || !b || !c || !_2q) return LF_ERR_ARG;
    return launch_ew<W_SUB>(a, b, c, rows, N, _2q, nullptr, nullptr, nullptr, nullptr, device, stream);
}

/* ntt (Rs = NULL) / enter_ntt (Rs != NULL) of a [batch][rows][N] stack, in place (ntt.cpp:166-216). */
int lf30_ntt(int32_t *a, int batch, int rows, int logN, const int32_t *psi_br, const int32_t *Rs, const int32_t *_2q,
             const int32_t *ql, const int32_t *qh, const int32_t *kl, const int32_t *kh, int device, void *stream) {
    if (!a || batch < 0 || rows < 0 || logN < 1 || logN > 2 * NTT_TILE_LOG_MAX || !psi_br || !_2q || !ql || !qh || !kl || !kh)
        return LF_ERR_ARG;
    if (batch == 0 || rows == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    const int64_t N = (int64_t)1 << logN;
    hipStream_t st = (hipStream_t)stream;
    if (Rs)
        for (int b = 0; b < batch; ++b)
            hipLaunchKernelGGL(ew30_kernel<W_ENTER>, dim3((unsigned)((N + 255) / 256), (unsigned)rows), dim3(256), 0, st,
                               a + (int64_t)b * rows * N, nullptr, a + (int64_t)b * rows * N, N, Rs, ql, qh, kl, kh);
    const dim3 grid((unsigned)((N / 2 + 255) / 256), (unsigned)(batch * rows));
    for (int s = 0; s < logN; ++s)   // stage s: t = N / 2^(s + 1)
        hipLaunchKernelGGL(stage30_kernel<false>, grid, dim3(256), 0, st, a, psi_br, rows, logN, logN - 1 - s, _2q, ql, qh, kl, kh);
    return (int)hipGetLastError();
}

/* intt* of a [batch][rows][N] stack, in place: the inverse stages, x (N^-1 R) (K.cu:527-529), then the chain's tail —
 * tail 0 = intt, 1 = intt_exit (+ mont_redc), 2 = intt_exit_reduce (+ reduce_2q), 3 = .._signed (+ make_signed)
 * (ntt.cpp:219-345). */
int lf30_intt(int32_t *a, int batch, int rows, int logN, const int32_t *ipsi_br, const int32_t *Ninv, int tail, const int32_t *_2q,
              const int32_t *ql, const int32_t *qh, const int32_t *kl, const int32_t *kh, int device, void *stream) {
    if (!a || batch < 0 || rows < 0 || logN < 1 || logN > 2 * NTT_TILE_LOG_MAX || tail < 0 || tail > 3 || !ipsi_br || !Ninv || !_2q ||
        !ql || !qh || !kl || !kh)
        return LF_ERR_ARG;
    if (batch == 0 || rows == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    const int64_t N = (int64_t)1 << logN;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((N / 2 + 255) / 256), (unsigned)(batch * rows));
    for (int s = 0; s < logN; ++s)   // stage s: t = 2^s
        hipLaunchKernelGGL(stage30_kernel<true>, grid, dim3(256), 0, st, a, ipsi_br, rows, logN, s, _2q, ql, qh, kl, kh);
    const dim3 eg((unsigned)((N + 255) / 256), (unsigned)rows);
    for (int b = 0; b < batch; ++b) {
        w30 *x = a + (int64_t)b * rows * N;
        hipLaunchKernelGGL(ew30_kernel<W_ENTER>, eg, dim3(256), 0, st, x, nullptr, x, N, Ninv, ql, qh, kl, kh);
        if (tail >= 1) hipLaunchKernelGGL(ew30_kernel<W_REDC>, eg, dim3(256), 0, st, x, nullptr, x, N, nullptr, ql, qh, kl, kh);
        if (tail >= 2) hipLaunchKernelGGL(ew30_kernel<W_REDUCE>, eg, dim3(256), 0, st, x, nullptr, x, N, _2q, nullptr, nullptr, nullptr, nullptr);
        if (tail >= 3) hipLaunchKernelGGL(ew30_kernel<W_SIGNED>, eg, dim3(256), 0, st, x, nullptr, x, N, _2q, nullptr, nullptr, nullptr, nullptr);
    }
    return (int)hipGetLastError();
}

}  // extern "C"
