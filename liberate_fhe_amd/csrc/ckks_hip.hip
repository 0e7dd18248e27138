// ckks_hip.hip — gfx950 (MI355X / CDNA4) kernels and C ABI for RNS-CKKS limb arithmetic.
//
// Design notes (see DESIGN.md for the full account):
//  * Word model = the reference's 62-bit mode: int64 words, Montgomery radix R = 2^62, lazy values
//    in [0, 2q).  The reference builds REDC62 from 31-bit half-words (K.cu:12-59); here the same
//    function is evaluated in closed form with native 64x64->128 products:
//        mm(a,b) = hi62(a*b) + floor(4*s*q / 2^64) + [lo62(a*b) != 0],  s = lo62(a*b) * k mod 2^62
//    which is exactly (a*b + s*q) / 2^62 for all |a|,|b| < 2^62, i.e. bit-identical outputs.
//  * The NTT family lives in ckks_ntt.hip, the engine-level fused kernels in ckks_fused.hip.
//  * No MFMA: this is 64-bit integer modular arithmetic.  No CUDA shims; gfx950 only.
#include <mutex>
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

#define LF_BATCH_MAX 8
struct GaloisBatch {
    const i64 *a[LF_BATCH_MAX];
    i64 *dst[LF_BATCH_MAX];
};

__global__ void __launch_bounds__(256) gather_rows_kernel(GaloisBatch gb, i64 N) {
    const i64 j = ((i64)blockIdx.x * 256 + threadIdx.x) * 2;
    if (j < N) *reinterpret_cast<longlong2 *>(gb.dst[blockIdx.y] + j) = *reinterpret_cast<const longlong2 *>(gb.a[blockIdx.y] + j);
}

__global__ void __launch_bounds__(256) galois_kernel(GaloisBatch gb, int logN, i64 p, const i64 *__restrict__ _2q) {
    const i64 *__restrict__ a = gb.a[blockIdx.z];
    i64 *__restrict__ dst = gb.dst[blockIdx.z];
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

// one wave: core-clock cycles per stretch of >= `ticks` ticks of the constant 100 MHz counter (lf_clock_probe)
__global__ void clock_probe_kernel(unsigned long long *out, int samples, unsigned long long ticks) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < samples; ++i) {
        const unsigned long long r0 = wall_clock64(), c0 = clock64();
        unsigned long long r1 = r0;
        while (r1 - r0 < ticks) r1 = wall_clock64();
        const unsigned long long c1 = clock64();
        out[2 * i] = c1 - c0;
        out[2 * i + 1] = r1 - r0;
    }
}

}  // namespace

// ---- formats of scratch that crosses native calls (ckks_common.h) ------------------------------------------------------------
namespace {
struct FmtRange {
    uintptr_t lo, hi;
    int fmt;
    unsigned long long age;
    unsigned epoch;   // lf_tune generation the range was written under
};
std::mutex g_fmt_mutex;
FmtRange g_fmt[256];
int g_fmt_n = 0;
unsigned long long g_fmt_clock = 0;
unsigned g_fmt_epoch = 0;
}  // namespace

// lf_tune changed a format knob: ranges noted before are now "written under another setting"
void lf_fmt_epoch_bump() {
    std::lock_guard<std::mutex> lock(g_fmt_mutex);
    ++g_fmt_epoch;
}

void lf_fmt_note(const void *p, size_t bytes, int fmt) {
    if (!p || !bytes) return;
    const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
    std::lock_guard<std::mutex> lock(g_fmt_mutex);
    // what this write covers is forgotten (ranges it cuts keep the part it left; a range split in two keeps the lower part:
    // forgetting is always safe, unknown ranges are taken on trust)
    int n = 0;
    for (int i = 0; i < g_fmt_n; ++i) {
        FmtRange r = g_fmt[i];
        if (r.hi <= lo || r.lo >= hi) {
            g_fmt[n++] = r;
            continue;
        }
        if (r.lo < lo) {
            r.hi = lo;
            g_fmt[n++] = r;
        } else if (r.hi > hi) {
            r.lo = hi;
            g_fmt[n++] = r;
        }
    }
    g_fmt_n = n;
    if (g_fmt_n == 256) {   // oldest out
        int o = 0;
        for (int i = 1; i < 256; ++i)
            if (g_fmt[i].age < g_fmt[o].age) o = i;
        g_fmt[o] = g_fmt[--g_fmt_n];
    }
    g_fmt[g_fmt_n++] = FmtRange{lo, hi, fmt, ++g_fmt_clock, g_fmt_epoch};
}

int lf_fmt_expect(const void *p, size_t bytes, int fmt) {
    if (!p || !bytes) return 0;
    const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
    std::lock_guard<std::mutex> lock(g_fmt_mutex);
    // a mismatching note made under the CURRENT knob setting is not this hazard: it is a stale note on memory the caller's
    // allocator has recycled (nobody unregisters a freed buffer), or a caller that passes explicit format flags and owns them
    for (int i = 0; i < g_fmt_n; ++i)
        if (g_fmt[i].hi > lo && g_fmt[i].lo < hi && g_fmt[i].fmt != fmt && g_fmt[i].epoch != g_fmt_epoch) return LF_ERR_STATE;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int lf_abi_version(void) { return LF_ABI_VERSION; }

int lf_clock_probe(uint64_t *out, int samples, uint64_t ticks, int device, void *stream) {
    // (at most 10 s of spinning in all: the wave holds its CU slot for samples x ticks x 10 ns)
    if (!out || samples < 1 || samples > 4096 || ticks < 1 || ticks > 100000000ull || (uint64_t)samples * ticks > 1000000000ull)
        return LF_ERR_ARG;
    if (int e = lf_set_device(device)) return e;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long *)out, samples,
                       (unsigned long long)ticks);
    return (int)hipGetLastError();
}

// capacities compiled into the kernels (ckks_common.h)
int lf_limits(int which) {
    switch (which) {
        case LF_LIMIT_DIGIT_LIMBS: return KS_MAX_ALPHA;
        case LF_LIMIT_SPECIAL_PRIMES: return KS_MAX_K;
        case LF_LIMIT_ROWS: return MAX_LIST_ROWS;
        case LF_LIMIT_BATCH: return LF_BATCH_MAX;
        case LF_LIMIT_LOGN: return 2 * NTT_TILE_LOG_MAX;
        default: return -1;
    }
}

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

int lf_galois_batch(const int64_t *const *a, int64_t *const *dst, int count, int rows, int logN, int64_t p,
                    const int64_t *_2q, int device, void *stream) {
    if (count < 0 || count > LF_BATCH_MAX || rows < 0 || logN < 1 || logN > 30 || p < 1 || p >= ((int64_t)2 << logN) || !(p & 1))
        return LF_ERR_ARG;
    for (int i = 0; i < count; ++i)
        if (a[i] == dst[i]) return LF_ERR_ARG;   // a permutation cannot run in place
    if (count == 0 || rows == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    GaloisBatch gb;
    for (int i = 0; i < count; ++i) gb.a[i] = (const i64 *)a[i], gb.dst[i] = (i64 *)dst[i];
    const i64 N = (i64)1 << logN;
    dim3 grid((unsigned)((N + 255) / 256), (unsigned)rows, (unsigned)count);
    hipLaunchKernelGGL(galois_kernel, grid, dim3(256), 0, (hipStream_t)stream, gb, logN, (i64)p, (const i64 *)_2q);
    return (int)hipGetLastError();
}

// count (<= LF_BATCH_MAX) rows of N words from anywhere into consecutive rows of dst: ONE launch where a host loop of copies
// would be `count` (the owner of a dropped limb stages the rows its peers need: ckks_engine._rescale_operands)
int lf_gather_rows(const int64_t *const *src, int64_t *dst, int count, int64_t N, int device, void *stream) {
    if (count < 0 || count > LF_BATCH_MAX || N < 2 || (N & 1) || !src || !dst) return LF_ERR_ARG;
    for (int i = 0; i < count; ++i)
        if (!src[i] || ((uintptr_t)src[i] & 15)) return LF_ERR_ARG;
    if ((uintptr_t)dst & 15) return LF_ERR_ARG;
    if (count == 0) return 0;
    if (int e = lf_set_device(device)) return e;
    GaloisBatch gb;
    for (int i = 0; i < count; ++i) gb.a[i] = (const i64 *)src[i], gb.dst[i] = (i64 *)dst + (i64)i * N;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((N / 2 + 255) / 256), (unsigned)count), dim3(256), 0, (hipStream_t)stream, gb, (i64)N);
    return (int)hipGetLastError();
}

int lf_galois(const int64_t *a, int64_t *dst, int rows, int logN, int64_t p, const int64_t *_2q, int device, void *stream) {
    return lf_galois_batch(&a, &dst, 1, rows, logN, p, _2q, device, stream);
}

}  // extern "C"
