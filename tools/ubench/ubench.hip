// Instruction-rate micro-benchmark for gfx950 (MI355X).
// Measures the issue rate of the integer / fp64 instructions that a 62-bit
// Montgomery product is built from, relative to v_fma_f32 (2 cycles per
// wave64 on a SIMD-32).  Output: one line per op, "cycles per wave-instr per
// SIMD" assuming the fp32 FMA baseline = 2.0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define CLB : "vcc", "s10", "s11"
#define REP8_32(INS) \
  asm volatile(INS : "+v"(r0) : "v"(b), "v"(c) CLB); asm volatile(INS : "+v"(r1) : "v"(b), "v"(c) CLB); \
  asm volatile(INS : "+v"(r2) : "v"(b), "v"(c) CLB); asm volatile(INS : "+v"(r3) : "v"(b), "v"(c) CLB); \
  asm volatile(INS : "+v"(r4) : "v"(b), "v"(c) CLB); asm volatile(INS : "+v"(r5) : "v"(b), "v"(c) CLB); \
  asm volatile(INS : "+v"(r6) : "v"(b), "v"(c) CLB); asm volatile(INS : "+v"(r7) : "v"(b), "v"(c) CLB);

#define KERNEL32(NAME, INS) \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, int iters) { \
  uint32_t r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
  uint32_t b = blockIdx.x * 2654435761u + 12345u, c = threadIdx.x * 40503u + 7u; \
  for (int i = 0; i < iters; ++i) { REP8_32(INS) REP8_32(INS) } \
  out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7; }

#define REP8_64(INS) \
  asm volatile(INS : "+v"(r0) : "v"(b), "v"(c) : "vcc"); asm volatile(INS : "+v"(r1) : "v"(b), "v"(c) : "vcc"); \
  asm volatile(INS : "+v"(r2) : "v"(b), "v"(c) : "vcc"); asm volatile(INS : "+v"(r3) : "v"(b), "v"(c) : "vcc"); \
  asm volatile(INS : "+v"(r4) : "v"(b), "v"(c) : "vcc"); asm volatile(INS : "+v"(r5) : "v"(b), "v"(c) : "vcc"); \
  asm volatile(INS : "+v"(r6) : "v"(b), "v"(c) : "vcc"); asm volatile(INS : "+v"(r7) : "v"(b), "v"(c) : "vcc");

// 64-bit accumulator, 32-bit sources
#define KERNEL64(NAME, INS) \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, int iters) { \
  uint64_t r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
  uint32_t b = blockIdx.x * 2654435761u + 12345u, c = threadIdx.x * 40503u + 7u; \
  for (int i = 0; i < iters; ++i) { REP8_64(INS) REP8_64(INS) } \
  uint64_t x = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7; \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(x ^ (x >> 32)); }

// 64-bit accumulator, 64-bit sources (fp64 / u64)
#define KERNEL64D(NAME, INS) \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, int iters) { \
  double r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
  double b = 1.0 + 1e-9 * blockIdx.x, c = 1e-3 * threadIdx.x; \
  for (int i = 0; i < iters; ++i) { REP8_64(INS) REP8_64(INS) } \
  double x = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7; \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(__double_as_longlong(x)); }

#define KERNEL64U(NAME, INS) \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, int iters) { \
  uint64_t r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
  uint64_t b = blockIdx.x * 0x9E3779B97F4A7C15ull + 12345u, c = threadIdx.x * 40503ull + 7u; \
  for (int i = 0; i < iters; ++i) { REP8_64(INS) REP8_64(INS) } \
  uint64_t x = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7; \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(x ^ (x >> 32)); }

KERNEL32(k_fma_f32,      "v_fma_f32 %0, %0, %1, %2")
KERNEL32(k_add_u32,      "v_add_u32 %0, %0, %1")
KERNEL32(k_mul_lo_u32,   "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi_u32,   "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_mul_u32_u24,  "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_mad_u32_u24,  "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL32(k_mul_hi_u24,   "v_mul_hi_u32_u24 %0, %0, %1")
KERNEL32(k_lshl_or,      "v_lshl_or_b32 %0, %0, 3, %1")
KERNEL32(k_cndmask,      "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL64(k_mad_u64_u32,  "v_mad_u64_u32 %0, vcc, %1, %2, %0")
KERNEL64(k_mad_i64_i32,  "v_mad_i64_i32 %0, vcc, %1, %2, %0")
KERNEL64D(k_fma_f64,     "v_fma_f64 %0, %0, %1, %2")
KERNEL64D(k_mul_f64,     "v_mul_f64 %0, %0, %1")
KERNEL64D(k_add_f64,     "v_add_f64 %0, %0, %1")
KERNEL64D(k_rndne_f64,   "v_rndne_f64 %0, %0")
KERNEL64D(k_floor_f64,   "v_floor_f64 %0, %0")
KERNEL64D(k_trunc_f64,   "v_trunc_f64 %0, %0")
KERNEL64D(k_ldexp_f64,   "v_ldexp_f64 %0, %0, 1")
KERNEL64D(k_addclamp_f64,"v_add_f64 %0, -%0, %1 clamp")
KERNEL64D(k_max_f64,     "v_max_f64 %0, %0, %1")
KERNEL32(k_cndmask_s,    "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
KERNEL32(k_cmp_cnd,      "v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc")
KERNEL32(k_cmp_u32,      "v_cmp_lt_u32 vcc, %0, %1")
KERNEL32(k_cmp_s,        "v_cmp_lt_u32_e64 s[10:11], %0, %1")
KERNEL32(k_min_u32,      "v_min_u32 %0, %0, %1")
KERNEL32(k_ashr,         "v_ashrrev_i32 %0, 31, %0")
KERNEL32(k_addco,        "v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %2, vcc")
KERNEL32(k_and_or,       "v_and_or_b32 %0, %0, %1, %2")
KERNEL32(k_xor,          "v_xor_b32 %0, %0, %1")
KERNEL32(k_bfi,          "v_bfi_b32 %0, %1, %2, %0")
KERNEL32(k_cvt_f64_u32,  "v_cvt_f32_u32 %0, %0")
KERNEL64U(k_cmp_u64,     "v_cmp_lt_u64 vcc, %0, %1")
KERNEL64U(k_lshl_add_u64,"v_lshl_add_u64 %0, %0, 1, %1")
KERNEL64U(k_lshlrev_b64, "v_lshlrev_b64 %0, 1, %0")
KERNEL64U(k_lshrrev_b64, "v_lshrrev_b64 %0, 1, %0")

// ---- whole modular products, written the way the product kernels will use them
__device__ __forceinline__ int64_t mm62(int64_t a, int64_t b, uint64_t q, uint64_t k) {
  const uint64_t M62 = (1ull << 62) - 1;
  __int128 x = (__int128)a * (__int128)b;
  uint64_t lo = (uint64_t)x; int64_t hi = (int64_t)(x >> 64);
  uint64_t xl = lo & M62;
  int64_t xh = (int64_t)(((uint64_t)hi << 2) | (lo >> 62));
  uint64_t s = (xl * k) & M62;
  uint64_t sq = __umul64hi(s << 2, q);
  return xh + (int64_t)sq + (xl != 0);
}
__global__ void __launch_bounds__(256) k_mm62(uint32_t* out, int iters) {
  int64_t r0 = threadIdx.x + 11, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3;
  uint64_t q = 1152921504606584833ull, k = 0x2F0F0F0F0F0F0F01ull | 1;  // k value irrelevant for timing
  int64_t w = blockIdx.x * 7919 + 1234567891234567ll;
  for (int i = 0; i < iters; ++i) {
    r0 = mm62(r0, w, q, k); r1 = mm62(r1, w, q, k); r2 = mm62(r2, w, q, k); r3 = mm62(r3, w, q, k);
  }
  int64_t x = r0 ^ r1 ^ r2 ^ r3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(x ^ (x >> 32));
}
// fp64 path for primes below 2^41: canonical a*w mod q through fma (operands < 2^42 held as doubles)
__device__ __forceinline__ double mulmod_dp(double a, double w, double q, double qinv) {
  double hi = a * w;
  double lo = fma(a, w, -hi);
  double quo = rint(hi * qinv);
  double r = fma(-quo, q, hi) + lo;
  r = r < 0 ? r + q : r;
  return r;
}
__global__ void __launch_bounds__(256) k_mm_dp(uint32_t* out, int iters) {
  double r0 = threadIdx.x + 11, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3;
  double q = 1099510054913.0, qinv = 1.0 / q;
  double w = blockIdx.x * 7919 + 123456789123.0;
  for (int i = 0; i < iters; ++i) {
    r0 = mulmod_dp(r0, w, q, qinv); r1 = mulmod_dp(r1, w, q, qinv); r2 = mulmod_dp(r2, w, q, qinv); r3 = mulmod_dp(r3, w, q, qinv);
  }
  double x = r0 + r1 + r2 + r3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(__double_as_longlong(x));
}

// ---- HBM streaming: read+write 16 B per lane
__global__ void __launch_bounds__(256) k_copy(const ulonglong2* __restrict__ src, ulonglong2* __restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) { ulonglong2 v = src[i]; v.x += 1; dst[i] = v; }
}

typedef void (*kern_t)(uint32_t*, int);

int main() {
  int dev = 0; CHECK(hipSetDevice(dev));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, dev));
  printf("device %s CUs=%d clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  const int blocks = prop.multiProcessorCount * 8, threads = 256;   // 8 waves per SIMD
  uint32_t* out; CHECK(hipMalloc(&out, (size_t)blocks * threads * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  struct Item { const char* name; kern_t k; int per_iter; };
  std::vector<Item> items = {
    {"v_fma_f32", k_fma_f32, 16}, {"v_add_u32", k_add_u32, 16}, {"v_mul_lo_u32", k_mul_lo_u32, 16},
    {"v_mul_hi_u32", k_mul_hi_u32, 16}, {"v_mul_u32_u24", k_mul_u32_u24, 16}, {"v_mad_u32_u24", k_mad_u32_u24, 16},
    {"v_mul_hi_u32_u24", k_mul_hi_u24, 16}, {"v_lshl_or_b32", k_lshl_or, 16}, {"v_cndmask_b32", k_cndmask, 16},
    {"v_mad_u64_u32", k_mad_u64_u32, 16}, {"v_mad_i64_i32", k_mad_i64_i32, 16},
    {"v_fma_f64", k_fma_f64, 16}, {"v_mul_f64", k_mul_f64, 16}, {"v_add_f64", k_add_f64, 16}, {"v_rndne_f64", k_rndne_f64, 16},
    {"v_floor_f64", k_floor_f64, 16}, {"v_trunc_f64", k_trunc_f64, 16}, {"v_ldexp_f64", k_ldexp_f64, 16},
    {"v_add_f64 clamp", k_addclamp_f64, 16}, {"v_max_f64", k_max_f64, 16},
    {"v_cndmask_b32 (sgpr mask)", k_cndmask_s, 16}, {"v_cmp_lt_u32+v_cndmask (pair)", k_cmp_cnd, 16},
    {"v_cmp_lt_u32 vcc", k_cmp_u32, 16}, {"v_cmp_lt_u32 sgpr", k_cmp_s, 16}, {"v_min_u32", k_min_u32, 16},
    {"v_ashrrev_i32", k_ashr, 16}, {"v_add_co+v_addc (pair)", k_addco, 16}, {"v_and_or_b32", k_and_or, 16},
    {"v_xor_b32", k_xor, 16}, {"v_bfi_b32", k_bfi, 16}, {"v_cvt_f32_u32", k_cvt_f64_u32, 16},
    {"v_cmp_lt_u64", k_cmp_u64, 16},
    {"v_lshl_add_u64", k_lshl_add_u64, 16}, {"v_lshlrev_b64", k_lshlrev_b64, 16}, {"v_lshrrev_b64", k_lshrrev_b64, 16},
    {"mm62 (int closed form)", k_mm62, 4}, {"mulmod_dp (fp64, q<2^41)", k_mm_dp, 4},
  };
  double base_rate = 0;
  for (auto& it : items) {
    int iters = (it.per_iter == 4) ? 16384 : 65536;
    it.k<<<blocks, threads>>>(out, iters);  // warm (also ramps the clock)
    CHECK(hipDeviceSynchronize());
    float ms = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(e0));
      it.k<<<blocks, threads>>>(out, iters);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float m; CHECK(hipEventElapsedTime(&m, e0, e1)); if (m < ms) ms = m;
    }
    double wave_instr = (double)blocks * (threads / 64) * (double)iters * it.per_iter;
    double rate = wave_instr / (ms * 1e-3);             // wave-instr per second, chip
    if (base_rate == 0) base_rate = rate;
    double cyc = 2.0 * base_rate / rate;                // cycles per wave-instr per SIMD, fma_f32 := 2
    double per_simd_ghz = rate / (prop.multiProcessorCount * 4.0) * cyc / 1e9;  // implied clock
    printf("%-28s %8.3f ms  %10.3e wave-ops/s  ~%6.2f cyc/wave-op/SIMD (implied clk %.2f GHz)  lane-ops/s %.3e\n",
           it.name, ms, rate, cyc, per_simd_ghz, rate * 64);
  }
  // HBM copy
  size_t bytes = (size_t)2 << 30; size_t n = bytes / 16;
  ulonglong2 *src, *dst; CHECK(hipMalloc(&src, bytes)); CHECK(hipMalloc(&dst, bytes));
  CHECK(hipMemset(src, 1, bytes)); CHECK(hipMemset(dst, 0, bytes));
  for (int g : {2048, 4096, 8192, 16384}) {
    k_copy<<<g, 256>>>(src, dst, n); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 50; ++r) k_copy<<<g, 256>>>(src, dst, n);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("copy 2GiB->2GiB grid=%d: %.3f ms/iter  %.2f TB/s (r+w)\n", g, ms / 50, 2.0 * bytes / (ms / 50 * 1e-3) / 1e12);
  }
  return 0;
}
