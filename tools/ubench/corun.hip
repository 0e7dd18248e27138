// Co-run probe (tools/corun_probe.py): a SLIM persistent streaming reader — `blocks` workgroups of 256 threads, all
// dispatched at once, each lane keeping UNROLL 16-byte nontemporal loads in flight over its share of a buffer — to be
// launched on a second stream beside a VALU-bound kernel of the library.  Question it answers: kernels of two streams do not
// interleave while the first still has workgroups to dispatch (LAB_NOTES rounds 1-4); does a kernel whose grid is fully
// resident from the start share the CUs with the next kernel's blocks, and at what rate does each then run?
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o corun.so corun.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef long long ll2_t __attribute__((ext_vector_type(2)));

template <int UNROLL>
__global__ void __launch_bounds__(256) stream_reader(const ll2_t *__restrict__ buf, uint64_t n16, long long *__restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    long long acc = 0;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        ll2_t v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(buf + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y;
    }
    for (; i < n16; i += stride) {
        const ll2_t v = __builtin_nontemporal_load(buf + i);
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x0123456789abcdefll) out[blockIdx.x] = acc;   // never true in practice: keeps the loads alive
}

// a slim persistent COPY (read + write, the column pass's traffic shape)
template <int UNROLL>
__global__ void __launch_bounds__(256) stream_copy(const ll2_t *__restrict__ src, ll2_t *__restrict__ dst, uint64_t n16) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        ll2_t v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) __builtin_nontemporal_store(v[u], dst + i + u * stride);
    }
}

extern "C" int corun_read(const void *buf, uint64_t bytes, void *out, int blocks, int unroll, void *stream) {
    const uint64_t n16 = bytes / 16;
    hipStream_t st = (hipStream_t)stream;
    switch (unroll) {
        case 2: hipLaunchKernelGGL(stream_reader<2>, dim3(blocks), dim3(256), 0, st, (const ll2_t *)buf, n16, (long long *)out); break;
        case 4: hipLaunchKernelGGL(stream_reader<4>, dim3(blocks), dim3(256), 0, st, (const ll2_t *)buf, n16, (long long *)out); break;
        case 8: hipLaunchKernelGGL(stream_reader<8>, dim3(blocks), dim3(256), 0, st, (const ll2_t *)buf, n16, (long long *)out); break;
        default: hipLaunchKernelGGL(stream_reader<16>, dim3(blocks), dim3(256), 0, st, (const ll2_t *)buf, n16, (long long *)out); break;
    }
    return (int)hipGetLastError();
}

extern "C" int corun_copy(const void *src, void *dst, uint64_t bytes, int blocks, int unroll, void *stream) {
    const uint64_t n16 = bytes / 16;
    hipStream_t st = (hipStream_t)stream;
    switch (unroll) {
        case 4: hipLaunchKernelGGL(stream_copy<4>, dim3(blocks), dim3(256), 0, st, (const ll2_t *)src, (ll2_t *)dst, n16); break;
        case 8: hipLaunchKernelGGL(stream_copy<8>, dim3(blocks), dim3(256), 0, st, (const ll2_t *)src, (ll2_t *)dst, n16); break;
        default: hipLaunchKernelGGL(stream_copy<16>, dim3(blocks), dim3(256), 0, st, (const ll2_t *)src, (ll2_t *)dst, n16); break;
    }
    return (int)hipGetLastError();
}
