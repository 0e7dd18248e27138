// ckks_csprng.hip — samplers behind key generation and encryption.
//
// Replaces the reference's four CUDA extensions (src/liberate/csprng/):
//   chacha20_cuda_kernel.{h,cu}         ChaCha20 block function over an [n,16] table of states
//   randint_cuda_kernel.cu              uniform integers in [0,q) from 128 random bits, plain and fused
//   discrete_gaussian_cuda_kernel.cu    CDT binary-tree walk (sigma = 3.2, 128-bit table), plain and fused
//   randround_cuda_kernel.cu            stochastic rounding of an fp64 vector
//
// Data layout is the reference's: one ChaCha20 state per 16 int64 words, each word holding a 32-bit
// value (csprng.py:124-160); words 12/13 are the 64-bit block counter, stepped by `step` after every
// draw.  One thread owns one state.  The block function runs on 16 uint32 registers (the reference
// keeps int64 copies in shared memory and masks after every add/rotate); rotates are single
// v_alignbit instructions.
//
// Roofline (measured, MI355X, 43 limbs x 65536 coefficients = 704,512 states per launch): a draw moves
// 128 B in, 16 B (counter) + 32 B (four samples) out = 176 B per state, 124 MB per launch, against
// ~1,150 VALU instructions per state (80 quarter rounds x 12 + sampling).  At 4 cycles per wave64
// instruction that is ~20 us of issue time per launch, the same order as the ~19 us the bytes need at
// the 6.4 TB/s streaming floor; the kernels run in 32 us (3.8 TB/s algorithmic), i.e. co-bound with
// partial overlap.  Staging the states through LDS for fully coalesced loads was measured slower
// (35 us): the strided per-thread loads are not what limits it.
//
// Small per-launch tables (moduli, CDT tree) travel as kernel arguments instead of __constant__
// symbols, so concurrent streams / devices never race on them.
#include "ckks_common.h"

#define LF_ERR_ARG 10001
#define RNG_THREADS 256
#define RNG_TABLE 128  // reference LUT_SIZE (randint_cuda_kernel.cu:13, discrete_gaussian_cuda_kernel.cu:16)

struct RngTable {
    u64 w[RNG_TABLE];
};

static __device__ __forceinline__ uint32_t rotl32(uint32_t v, int s) { return __builtin_rotateleft32(v, s); }

#define LF_QR(a, b, c, d)     \
    a += b;                   \
    d = rotl32(d ^ a, 16);    \
    c += d;                   \
    b = rotl32(b ^ c, 12);    \
    a += b;                   \
    d = rotl32(d ^ a, 8);     \
    c += d;                   \
    b = rotl32(b ^ c, 7)

// Loads one state (low 32 bits of each int64 word), runs 10 double rounds, adds the input back
// (chacha20_cuda_kernel.cu:21-40) and steps the 64-bit counter in place (:42-45).
static __device__ __forceinline__ void chacha_draw(i64 *state, u64 step, uint32_t o[16]) {
    uint32_t s[16];
    const longlong2 *p = (const longlong2 *)state;
    longlong2 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = p[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        s[2 * i] = (uint32_t)v[i].x;
        s[2 * i + 1] = (uint32_t)v[i].y;
    }
    uint32_t x0 = s[0], x1 = s[1], x2 = s[2], x3 = s[3], x4 = s[4], x5 = s[5], x6 = s[6], x7 = s[7];
    uint32_t x8 = s[8], x9 = s[9], x10 = s[10], x11 = s[11], x12 = s[12], x13 = s[13], x14 = s[14], x15 = s[15];
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        LF_QR(x0, x4, x8, x12);
        LF_QR(x1, x5, x9, x13);
        LF_QR(x2, x6, x10, x14);
        LF_QR(x3, x7, x11, x15);
        LF_QR(x0, x5, x10, x15);
        LF_QR(x1, x6, x11, x12);
        LF_QR(x2, x7, x8, x13);
        LF_QR(x3, x4, x9, x14);
    }
    o[0] = x0 + s[0], o[1] = x1 + s[1], o[2] = x2 + s[2], o[3] = x3 + s[3];
    o[4] = x4 + s[4], o[5] = x5 + s[5], o[6] = x6 + s[6], o[7] = x7 + s[7];
    o[8] = x8 + s[8], o[9] = x9 + s[9], o[10] = x10 + s[10], o[11] = x11 + s[11];
    o[12] = x12 + s[12], o[13] = x13 + s[13], o[14] = x14 + s[14], o[15] = x15 + s[15];
    // Counter step, in the reference's int64 arithmetic on the stored words.
    i64 c12 = v[6].x + (i64)step;
    i64 c13 = v[6].y + (c12 >> 32);
    c12 &= 0xffffffffll;
    longlong2 c;
    c.x = c12, c.y = c13;
    ((longlong2 *)state)[6] = c;
}

// floor(p * X / 2^128), X = (w2 : w3 : w0 : w1) as 32-bit digits from most to least significant
// (randint_cuda_kernel.cu:60-97; the 32-bit carry chain there is the exact 64x128 product).
static __device__ __forceinline__ u64 scale128(u64 p, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    const u64 xl = ((u64)w0 << 32) | w1;
    const u64 xh = ((u64)w2 << 32) | w3;
    const u64 alpha = __umul64hi(p, xl);
    const u64 lo = p * xh;
    return __umul64hi(p, xh) + (u64)(lo + alpha < lo);
}

// The tree arrives as a kernel argument and is staged in LDS, where the per-lane walk indexes it.
static __device__ __forceinline__ void stage_tree(u64 *lut, const RngTable &tab, int size) {
    if ((int)threadIdx.x < 2 * size) lut[threadIdx.x] = tab.w[threadIdx.x];
    __syncthreads();
}

// CDT tree walk (discrete_gaussian_cuda_kernel.cu:61-106).  tab[0..size) = low words of the nodes
// in level order, tab[size..2*size) = high words.
static __device__ __forceinline__ i64 cdt_walk(const u64 *tab, int size, int depth, uint32_t w0, uint32_t w1,
                                               uint32_t w2, uint32_t w3) {
    const u64 xl = ((u64)w0 << 32) | w1;
    u64 xh = ((u64)w2 << 32) | w3;
    const i64 sign = (i64)(xh & 1);
    xh >>= 1;
    int jump = 1, current = 0, counter = 0;
    for (int j = 0; j < depth; ++j) {
        const u64 yh = tab[counter + current + size], yl = tab[counter + current];
        const int ge = (xh > yh) | ((xh == yh) & (xl >= yl));
        current = 2 * current + ge;
        counter += jump;
        jump *= 2;
    }
    return (sign * 2 - 1) * (i64)current;
}

__global__ void __launch_bounds__(RNG_THREADS) chacha20_kernel(i64 *states, i64 *dest, i64 n, u64 step) {
    const i64 idx = (i64)blockIdx.x * RNG_THREADS + threadIdx.x;
    if (idx >= n) return;
    uint32_t o[16];
    chacha_draw(states + idx * 16, step, o);
    longlong2 *d = (longlong2 *)(dest + idx * 16);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        longlong2 t;
        t.x = (i64)o[2 * i], t.y = (i64)o[2 * i + 1];
        d[i] = t;
    }
}

// states [channels, L, 16] -> dst [channels, 4L]; one modulus per channel (blockIdx.y).
__global__ void __launch_bounds__(RNG_THREADS)
    randint_fast_kernel(i64 *states, i64 *dst, i64 L, RngTable q, i64 shift, u64 step) {
    const i64 idx = (i64)blockIdx.x * RNG_THREADS + threadIdx.x;
    if (idx >= L) return;
    const int ch = blockIdx.y;
    uint32_t o[16];
    chacha_draw(states + ((i64)ch * L + idx) * 16, step, o);
    const u64 p = q.w[ch];
    longlong2 r0, r1;
    r0.x = (i64)scale128(p, o[0], o[1], o[2], o[3]) + shift;
    r0.y = (i64)scale128(p, o[4], o[5], o[6], o[7]) + shift;
    r1.x = (i64)scale128(p, o[8], o[9], o[10], o[11]) + shift;
    r1.y = (i64)scale128(p, o[12], o[13], o[14], o[15]) + shift;
    longlong2 *d = (longlong2 *)(dst + ((i64)ch * L + idx) * 4);
    d[0] = r0, d[1] = r1;
}

// states [n,16] -> dst [4n].
__global__ void __launch_bounds__(RNG_THREADS)
    gaussian_fast_kernel(i64 *states, i64 *dst, i64 n, RngTable tab, int size, int depth, u64 step) {
    __shared__ u64 lut[RNG_TABLE];
    stage_tree(lut, tab, size);
    const i64 idx = (i64)blockIdx.x * RNG_THREADS + threadIdx.x;
    if (idx >= n) return;
    uint32_t o[16];
    chacha_draw(states + idx * 16, step, o);
    longlong2 r0, r1;
    r0.x = cdt_walk(lut, size, depth, o[0], o[1], o[2], o[3]);
    r0.y = cdt_walk(lut, size, depth, o[4], o[5], o[6], o[7]);
    r1.x = cdt_walk(lut, size, depth, o[8], o[9], o[10], o[11]);
    r1.y = cdt_walk(lut, size, depth, o[12], o[13], o[14], o[15]);
    longlong2 *d = (longlong2 *)(dst + idx * 4);
    d[0] = r0, d[1] = r1;
}

// The unfused forms work in place on a table of random words [.., 16]: word 4j of each row is
// replaced by the sample made from words 4j..4j+3 (randint_cuda_kernel.cu:108-152,
// discrete_gaussian_cuda_kernel.cu:118-168).
__global__ void __launch_bounds__(RNG_THREADS) randint_inplace_kernel(i64 *rb, i64 n, RngTable q) {
    const i64 idx = (i64)blockIdx.x * RNG_THREADS + threadIdx.x;
    if (idx >= n * 4) return;
    const int ch = blockIdx.y;
    i64 *w = rb + ((i64)ch * n * 4 + idx) * 4;
    w[0] = (i64)scale128(q.w[ch], (uint32_t)w[0], (uint32_t)w[1], (uint32_t)w[2], (uint32_t)w[3]);
}

__global__ void __launch_bounds__(RNG_THREADS)
    gaussian_inplace_kernel(i64 *rb, i64 n, RngTable tab, int size, int depth) {
    __shared__ u64 lut[RNG_TABLE];
    stage_tree(lut, tab, size);
    const i64 idx = (i64)blockIdx.x * RNG_THREADS + threadIdx.x;
    if (idx >= n * 4) return;
    i64 *w = rb + idx * 4;
    w[0] = cdt_walk(lut, size, depth, (uint32_t)w[0], (uint32_t)w[1], (uint32_t)w[2], (uint32_t)w[3]);
}

// randround_cuda_kernel.cu:8-37: sign * (floor|x| + [r < rn(frac * 2^32)]), r a 32-bit random word.
__global__ void __launch_bounds__(RNG_THREADS) randround_kernel(const double *coef, i64 *rnd, i64 n) {
    const i64 idx = (i64)blockIdx.x * RNG_THREADS + threadIdx.x;
    if (idx >= n) return;
    const double c = coef[idx];
    const double a = fabs(c);
    const double ip = floor(a);
    const i64 ifrac = (i64)rint((a - ip) * 4294967296.0);
    const i64 up = rnd[idx] < ifrac;
    const i64 sign = signbit(c) ? -1 : 1;
    rnd[idx] = sign * ((i64)ip + up);
}

static inline unsigned rng_blocks(i64 n) { return (unsigned)((n + RNG_THREADS - 1) / RNG_THREADS); }

extern "C" {

int lf_chacha20(int64_t *states, int64_t *dest, int64_t n, uint64_t step, int device, void *stream) {
    if (n < 0) return LF_ERR_ARG;
    if (n == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    hipLaunchKernelGGL(chacha20_kernel, dim3(rng_blocks(n)), dim3(RNG_THREADS), 0, (hipStream_t)stream, (i64 *)states,
                       (i64 *)dest, (i64)n, (u64)step);
    return (int)hipGetLastError();
}

int lf_randint_fast(int64_t *states, int64_t *dst, int channels, int64_t L, const uint64_t *q_host, int64_t shift,
                    uint64_t step, int device, void *stream) {
    if (channels < 0 || L < 0) return LF_ERR_ARG;
    if (channels == 0 || L == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    // The reference holds at most 128 moduli per launch; longer lists go out in slices.
    for (int c0 = 0; c0 < channels; c0 += RNG_TABLE) {
        const int nc = channels - c0 < RNG_TABLE ? channels - c0 : RNG_TABLE;
        RngTable q;
        for (int i = 0; i < nc; ++i) q.w[i] = q_host[c0 + i];
        hipLaunchKernelGGL(randint_fast_kernel, dim3(rng_blocks(L), nc), dim3(RNG_THREADS), 0, (hipStream_t)stream,
                           (i64 *)states + (i64)c0 * L * 16, (i64 *)dst + (i64)c0 * L * 4, (i64)L, q, (i64)shift,
                           (u64)step);
    }
    return (int)hipGetLastError();
}

int lf_randint(int64_t *rand_bytes, int channels, int64_t n, const uint64_t *q_host, int device, void *stream) {
    if (channels < 0 || n < 0) return LF_ERR_ARG;
    if (channels == 0 || n == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    for (int c0 = 0; c0 < channels; c0 += RNG_TABLE) {
        const int nc = channels - c0 < RNG_TABLE ? channels - c0 : RNG_TABLE;
        RngTable q;
        for (int i = 0; i < nc; ++i) q.w[i] = q_host[c0 + i];
        hipLaunchKernelGGL(randint_inplace_kernel, dim3(rng_blocks(n * 4), nc), dim3(RNG_THREADS), 0,
                           (hipStream_t)stream, (i64 *)rand_bytes + (i64)c0 * n * 16, (i64)n, q);
    }
    return (int)hipGetLastError();
}

static int load_tree(RngTable &tab, const uint64_t *btree_host, int btree_size, int depth) {
    if (btree_size <= 0 || 2 * btree_size > RNG_TABLE || depth < 0 || ((1ll << depth) - 1) > btree_size)
        return LF_ERR_ARG;
    for (int i = 0; i < 2 * btree_size; ++i) tab.w[i] = btree_host[i];
    return 0;
}

int lf_discrete_gaussian_fast(int64_t *states, int64_t *dst, int64_t n, const uint64_t *btree_host, int btree_size,
                              int depth, uint64_t step, int device, void *stream) {
    RngTable tab;
    if (n < 0) return LF_ERR_ARG;
    if (int e = load_tree(tab, btree_host, btree_size, depth)) return e;
    if (n == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    hipLaunchKernelGGL(gaussian_fast_kernel, dim3(rng_blocks(n)), dim3(RNG_THREADS), 0, (hipStream_t)stream,
                       (i64 *)states, (i64 *)dst, (i64)n, tab, btree_size, depth, (u64)step);
    return (int)hipGetLastError();
}

int lf_discrete_gaussian(int64_t *rand_bytes, int64_t n, const uint64_t *btree_host, int btree_size, int depth,
                         int device, void *stream) {
    RngTable tab;
    if (n < 0) return LF_ERR_ARG;
    if (int e = load_tree(tab, btree_host, btree_size, depth)) return e;
    if (n == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    hipLaunchKernelGGL(gaussian_inplace_kernel, dim3(rng_blocks(n * 4)), dim3(RNG_THREADS), 0, (hipStream_t)stream,
                       (i64 *)rand_bytes, (i64)n, tab, btree_size, depth);
    return (int)hipGetLastError();
}

int lf_randround(const double *coef, int64_t *rand_bytes, int64_t n, int device, void *stream) {
    if (n < 0) return LF_ERR_ARG;
    if (n == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    hipLaunchKernelGGL(randround_kernel, dim3(rng_blocks(n)), dim3(RNG_THREADS), 0, (hipStream_t)stream, coef,
                       (i64 *)rand_bytes, (i64)n);
    return (int)hipGetLastError();
}

}  // extern "C"
