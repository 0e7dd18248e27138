// A/B of the two ways a wave can run the four small-distance stages of an NTT on gfx950 (MI355X):
//
//   regs   every lane holds 16 consecutive words in registers: the partner of a butterfly is another REGISTER of the same
//          lane, a butterfly is one modular product + add + sub (what csrc/ckks_ntt_tile16.h does);
//   lanes  every lane holds ONE word of each of 16 independent rows: the partner of a butterfly at distance d is lane ^ d,
//          reached with DPP quad_perm (d = 1, 2) / ds_swizzle (d = 4, 8) — "wavefront-level shuffles for the small radix
//          stages".  Both lanes of a pair must hold the product, so the b-lane computes it and the halves are exchanged:
//          per word and stage  1 product (every lane executes it, half of them for nothing) + 2 selects + 2 lane
//          moves + 1 sign flip + 1 add.
// Same data per lane (16 fp64 words), same butterflies per wave (16 words x 4 stages / 2 per lane = 32 per lane), relaxed
// fp64 arithmetic of the fused key switch (8 instructions per butterfly in the regs form), everything in registers.
// Output: wave-level butterflies per second of both forms.   hipcc --offload-arch=gfx950 -O3 -o shuffle_ab shuffle_ab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Mod { double q, qinv; };

__device__ __forceinline__ double mulmod_bal(double a, double w, const Mod &m) {   // (a * w) mod q, balanced, |a| < 2^52
    const double hi = a * w;
    const double lo = __builtin_fma(a, w, -hi);
    const double quo = __builtin_rint(hi * m.qinv);
    return __builtin_fma(-quo, m.q, hi) + lo;
}
__device__ __forceinline__ double fold(double x, const Mod &m) { return __builtin_fma(-__builtin_rint(x * m.qinv), m.q, x); }

__global__ void __launch_bounds__(256) k_regs(double *out, int iters, Mod m, double w0) {
    double x[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = (double)((threadIdx.x * 16 + e) * 2654435761u % 1000003u);
    double w = w0 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int du = 8 >> u;
#pragma unroll
            for (int j = 0; j < (1 << u); ++j)
#pragma unroll
                for (int t = 0; t < du; ++t) {
                    const int a = (j << (4 - u)) + t, b = a + du;
                    const double U = x[a], V = mulmod_bal(x[b], w + (double)(u * 8 + j), m);
                    x[a] = U + V;
                    x[b] = U - V;
                }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) x[e] = fold(x[e], m);   // keep magnitudes bounded (both forms pay it)
    }
    double s = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += x[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// lane ^ d through DPP quad_perm (d = 1, 2) or ds_swizzle in bit-mask mode (d = 4, 8): two 32-bit moves per 64-bit word
template <int D>
__device__ __forceinline__ double lane_xor(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    if (D == 1) {
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false);
    } else if (D == 2) {
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xF, 0xF, false);
    } else {
        lo = __builtin_amdgcn_ds_swizzle(lo, 0x001F | (D << 10));           // and 0x1f, or 0, xor D
        hi = __builtin_amdgcn_ds_swizzle(hi, 0x001F | (D << 10));
    }
    return __hiloint2double(hi, lo);
}

template <int D>
__device__ __forceinline__ void lane_stage(double (&x)[16], double w, const Mod &m, bool is_b, int flip) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const double V = mulmod_bal(x[e], w + (double)e, m);          // needed in the b-lanes only, executed by all
        const double X = is_b ? V : x[e];                             // what the partner needs from this lane
        const double Y = lane_xor<D>(X);                              // a-lane receives V, b-lane receives U
        const double Z = __hiloint2double(__double2hiint(X) ^ flip, __double2loint(X));   // b-lane: -V, a-lane: U
        x[e] = Y + Z;                                                 // a: U + V      b: U - V
    }
}

__global__ void __launch_bounds__(256) k_lanes(double *out, int iters, Mod m, double w0) {
    double x[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = (double)((threadIdx.x * 16 + e) * 2654435761u % 1000003u);
    const double w = w0 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        lane_stage<8>(x, w, m, lane & 8, (lane & 8) ? (int)0x80000000 : 0);
        lane_stage<4>(x, w + 16.0, m, lane & 4, (lane & 4) ? (int)0x80000000 : 0);
        lane_stage<2>(x, w + 32.0, m, lane & 2, (lane & 2) ? (int)0x80000000 : 0);
        lane_stage<1>(x, w + 48.0, m, lane & 1, (lane & 1) ? (int)0x80000000 : 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) x[e] = fold(x[e], m);
    }
    double s = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += x[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 4, threads = 256, iters = 4096;   // 4 waves per SIMD, like the tiled pass
    double *out;
    CHECK(hipMalloc(&out, (size_t)blocks * threads * 8));
    const Mod m{1099511922689.0, 1.0 / 1099511922689.0};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const char *names[2] = {"regs  (16 words per lane, partner = register)", "lanes (DPP quad_perm / ds_swizzle, partner = lane ^ d)"};
    double rate[2];
    for (int k = 0; k < 2; ++k) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0));
            if (k == 0) k_regs<<<blocks, threads>>>(out, iters, m, 3.0);
            else k_lanes<<<blocks, threads>>>(out, iters, m, 3.0);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        // butterflies per lane and iteration: 16 words x 4 stages / 2 = 32 in both forms
        rate[k] = (double)blocks * threads * iters * 32.0 / (best * 1e-3);
        printf("%-58s %8.3f ms  %8.2f G butterflies/s\n", names[k], best, rate[k] / 1e9);
    }
    printf("lanes / regs = %.2f x the time per butterfly\n", rate[0] / rate[1]);
    return 0;
}
