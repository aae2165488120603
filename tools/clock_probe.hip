// Diagnostic: shader-clock frequency seen by a resident wave while other kernels run.
// s_memtime counts shader-engine clocks, s_memrealtime a constant 100 MHz reference; one wave spins for `ticks`
// reference ticks and reports both deltas.  Build: hipcc --offload-arch=gfx950 -shared -fPIC tools/clock_probe.hip -o tools/libclockprobe.so
#include <hip/hip_runtime.h>

__global__ void clock_probe_kernel(unsigned long long *out, unsigned long long ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r = r0;
    while (r - r0 < ticks) {
        __builtin_amdgcn_s_sleep(64);
        r = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    out[0] = c1 - c0;
    out[1] = r - r0;
}

extern "C" int clock_probe_launch(unsigned long long *out_dev, unsigned long long ticks, void *stream) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out_dev, ticks);
    return (int)hipGetLastError();
}
