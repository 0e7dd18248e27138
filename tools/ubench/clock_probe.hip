// Shader-clock probe: one wave spins for `spin_us` microseconds of wall time and reports how many core-clock cycles
// (s_memtime) passed per tick of the constant 100 MHz counter (s_memrealtime): the frequency the CUs actually run at while
// other kernels execute next to it.   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libclock_probe.so clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ void clock_probe_kernel(uint64_t *out, int samples, uint64_t ticks) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < samples; ++i) {
        const uint64_t r0 = wall_clock64(), c0 = clock64();
        uint64_t r1 = r0;
        while (r1 - r0 < ticks) r1 = wall_clock64();
        const uint64_t c1 = clock64();
        out[2 * i] = c1 - c0;
        out[2 * i + 1] = r1 - r0;
    }
}

extern "C" int clock_probe(uint64_t *out, int samples, uint64_t ticks, void *stream) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out, samples, ticks);
    return (int)hipGetLastError();
}
