// A/B of the two ways a wave can do the SECOND exchange of the 4096-word tile (csrc/ckks_ntt_tile16.h) on gfx950 (MI355X):
// a 16 x 16 transpose of 64-bit words among 16 lanes that hold 16 words each, between two radix-16 register steps.
//
//   lds    what the kernel ships: 16 ds_write_b64 at stride 17 words into the wave's own LDS span, wave-local wait,
//          16 ds_read_b64 (conflict-free padded layout, no block barrier needed);
//   regs   no LDS: four rounds of half-register swaps.  The 16 lanes of a group are chosen so that two of the four lane
//          bits are the ROW bits of the wave (lane bits 4, 5: v_permlane16_swap / v_permlane32_swap, new on gfx950 — one
//          instruction swaps a dword of the odd rows of one register with the even rows of another, exactly a transpose
//          step) and two are the QUAD bits (lane bits 0, 1: DPP quad_perm on v_cndmask, two instructions per dword pair).
//          Per 64-bit word pair: 2 instructions in a permlane round, 4 in a DPP round; 8 pairs per round:
//          2 x 16 + 2 x 32 = 96 VALU instructions per lane and transpose, no memory instruction at all.
// Both kernels run the same relaxed fp64 radix-16 step (32 butterflies per lane) between transposes, so the measured
// difference is the exchange alone at the instruction mix of the real kernel (4 waves per SIMD).
// Output: time per (step + transpose) of both forms.   hipcc --offload-arch=gfx950 -O3 -o transpose_ab transpose_ab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Mod { double q, qinv; };

__device__ __forceinline__ double mulmod_bal(double a, double w, const Mod &m) {
    const double hi = a * w;
    const double lo = __builtin_fma(a, w, -hi);
    const double quo = __builtin_rint(hi * m.qinv);
    return __builtin_fma(-quo, m.q, hi) + lo;
}
__device__ __forceinline__ double fold(double x, const Mod &m) { return __builtin_fma(-__builtin_rint(x * m.qinv), m.q, x); }

__device__ __forceinline__ void step16(double (&x)[16], double w, const Mod &m) {
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
    for (int e = 0; e < 16; ++e) x[e] = fold(x[e], m);
}

// ---- lds: the shipped exchange (wave-private span, one pad word per 16) ------------------------------------------------
__global__ void __launch_bounds__(256, 4) k_lds(double *out, int iters, Mod m, double w0) {
    __shared__ double sm[4096 + 256 + 1];
    const int w = threadIdx.x;
    double x[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = (double)((w * 16 + e) * 2654435761u % 1000003u);
    const int pB = ((w >> 4) << 8) | (w & 15);
    double *spB = sm + pB + (pB >> 4);      // stride-16 view: element e at + 17 e
    double *spC = sm + 17 * w;              // 16 consecutive words
    for (int it = 0; it < iters; ++it) {
        step16(x, w0 + w, m);
#pragma unroll
        for (int e = 0; e < 16; ++e) spB[e * 17] = x[e];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the 16 lanes of a 256-word block sit in one wave
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int e = 0; e < 16; ++e) x[e] = spC[e];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    double s = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += x[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- regs: 16 x 16 transpose over lane bits {0, 1, 4, 5} ---------------------------------------------------------------
// round over register bit RB and a ROW lane bit: lo-lane keeps X (reg bit 0), receives into Y (reg bit 1) the partner's X
template <int ROWS32>
__device__ __forceinline__ void swap_rows(double &X, double &Y) {
    unsigned xl = (unsigned)__double2loint(X), xh = (unsigned)__double2hiint(X);
    unsigned yl = (unsigned)__double2loint(Y), yh = (unsigned)__double2hiint(Y);
    // v_permlaneNN_swap vdst, vsrc: the upper half (odd rows) of vdst <-> the lower half (even rows) of vsrc
    if (ROWS32) {
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(xl), "+v"(yl));
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(xh), "+v"(yh));
    } else {
        asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(xl), "+v"(yl));
        asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(xh), "+v"(yh));
    }
    X = __hiloint2double((int)xh, (int)xl);
    Y = __hiloint2double((int)yh, (int)yl);
}

// round over a QUAD lane bit (D = 1: lane ^ 1, D = 2: lane ^ 2): even lane Y' = partner's X, odd lane X' = partner's Y
template <int D>
__device__ __forceinline__ void swap_quad(double &X, double &Y, bool odd) {
    const int xl = __double2loint(X), xh = __double2hiint(X), yl = __double2loint(Y), yh = __double2hiint(Y);
    constexpr int ctrl = D == 1 ? 0xB1 : 0x4E;   // quad_perm [1,0,3,2] / [2,3,0,1]
    const int pxl = __builtin_amdgcn_update_dpp(xl, xl, ctrl, 0xF, 0xF, false), pxh = __builtin_amdgcn_update_dpp(xh, xh, ctrl, 0xF, 0xF, false);
    const int pyl = __builtin_amdgcn_update_dpp(yl, yl, ctrl, 0xF, 0xF, false), pyh = __builtin_amdgcn_update_dpp(yh, yh, ctrl, 0xF, 0xF, false);
    // (the compiler folds each move into the select: v_cndmask_b32_dpp)
    X = __hiloint2double(odd ? pyh : xh, odd ? pyl : xl);
    Y = __hiloint2double(odd ? yh : pxh, odd ? yl : pxl);
}

__global__ void __launch_bounds__(256, 4) k_regs(double *out, int iters, Mod m, double w0) {
    const int w = threadIdx.x;
    const int lane = w & 63;
    double x[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = (double)((w * 16 + e) * 2654435761u % 1000003u);
    const bool odd1 = lane & 1, odd2 = lane & 2;
    for (int it = 0; it < iters; ++it) {
        step16(x, w0 + w, m);
        // register bit 3 <-> lane bit 5, bit 2 <-> lane bit 4, bit 1 <-> lane bit 1, bit 0 <-> lane bit 0
#pragma unroll
        for (int e = 0; e < 8; ++e) swap_rows<1>(x[e], x[e + 8]);
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (!(e & 4)) swap_rows<0>(x[e], x[e + 4]);
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (!(e & 2)) swap_quad<2>(x[e], x[e + 2], odd2);
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (!(e & 1)) swap_quad<1>(x[e], x[e + 1], odd1);
    }
    double s = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += x[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- the radix-16 step alone (what both pay besides the exchange) -------------------------------------------------------
__global__ void __launch_bounds__(256, 4) k_none(double *out, int iters, Mod m, double w0) {
    const int w = threadIdx.x;
    double x[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = (double)((w * 16 + e) * 2654435761u % 1000003u);
    for (int it = 0; it < iters; ++it) step16(x, w0 + w, m);
    double s = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += x[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// correctness of the register network: element (lane group position r, register c) ends up at (c, r)
__global__ void k_check(int *bad) {
    const int lane = threadIdx.x & 63;
    const int r = (lane & 3) | (((lane >> 4) & 3) << 2);      // position of the lane inside its group of 16 (bits 0, 1, 4, 5)
    const int grp = (lane >> 2) & 3;
    double x[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) x[c] = (double)(grp * 1000 + r * 16 + c);
    const bool odd1 = lane & 1, odd2 = lane & 2;
#pragma unroll
    for (int e = 0; e < 8; ++e) swap_rows<1>(x[e], x[e + 8]);
#pragma unroll
    for (int e = 0; e < 16; ++e)
        if (!(e & 4)) swap_rows<0>(x[e], x[e + 4]);
#pragma unroll
    for (int e = 0; e < 16; ++e)
        if (!(e & 2)) swap_quad<2>(x[e], x[e + 2], odd2);
#pragma unroll
    for (int e = 0; e < 16; ++e)
        if (!(e & 1)) swap_quad<1>(x[e], x[e + 1], odd1);
    int wrong = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) wrong += x[c] != (double)(grp * 1000 + c * 16 + r);
    if (wrong) atomicAdd(bad, wrong);
}

int main() {
    const int blocks = 1024, threads = 256, iters = 4096;
    double *out;
    int *bad, hbad = -1;
    CHECK(hipMalloc(&out, sizeof(double) * blocks * threads));
    CHECK(hipMalloc(&bad, sizeof(int)));
    CHECK(hipMemset(bad, 0, sizeof(int)));
    hipLaunchKernelGGL(k_check, dim3(4), dim3(256), 0, 0, bad);
    CHECK(hipMemcpy(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost));
    printf("register transpose network (permlane32_swap, permlane16_swap, DPP quad_perm x 2): %s\n", hbad == 0 ? "correct" : "WRONG");
    const Mod m{1099511590913.0, 1.0 / 1099511590913.0};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float ms[3] = {0, 0, 0};
    const char *names[3] = {"step alone (no exchange)", "step + LDS exchange (ds_write_b64 x16, ds_read_b64 x16)", "step + register transpose (96 VALU, no LDS)"};
    for (int rep = 0; rep < 3; ++rep) {
        for (int k = 0; k < 3; ++k) {
            CHECK(hipEventRecord(e0));
            if (k == 0) hipLaunchKernelGGL(k_none, dim3(blocks), dim3(threads), 0, 0, out, iters, m, 3.0);
            if (k == 1) hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(threads), 0, 0, out, iters, m, 3.0);
            if (k == 2) hipLaunchKernelGGL(k_regs, dim3(blocks), dim3(threads), 0, 0, out, iters, m, 3.0);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms[k], e0, e1));
        }
    }
    for (int k = 0; k < 3; ++k) printf("%-60s %8.3f ms   %7.2f ns per (wave step + exchange)\n", names[k], ms[k], ms[k] * 1e6 / ((double)iters * blocks * 4) * 1024 * 4 / 1024 / 4);
    printf("exchange alone: LDS %+.3f ms, registers %+.3f ms  ->  registers / LDS = %.2f\n", ms[1] - ms[0], ms[2] - ms[0], (ms[2] - ms[0]) / (ms[1] - ms[0]));
    return hbad == 0 ? 0 : 1;
}
