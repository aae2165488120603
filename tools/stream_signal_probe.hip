// What does a cross-stream hand-over cost the PRODUCING stream?  N short kernels back to back on stream A, with after each of them
//   (0) nothing, (1) hipEventRecord + hipStreamWaitEvent on stream B, (2) hipStreamWriteValue32 on A + hipStreamWaitValue32 on B,
//   (3) hipEventRecord only (nobody waits);
// stream B runs one short kernel behind every wait.  Prints the time of stream A's N kernels (events around them) per mode.
// build: hipcc --offload-arch=gfx950 -O2 tools/stream_signal_probe.hip -o tools/probe_bin/stream_signal_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin(float *p, int iters) {
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    p[threadIdx.x] = v;
}

int main() {
    const int N = 200, ITERS = 20000;          // ~25 us per kernel
    hipStream_t a, b;
    CK(hipStreamCreate(&a));
    CK(hipStreamCreate(&b));
    float *da, *db;
    CK(hipMalloc(&da, 1024 * 4));
    CK(hipMalloc(&db, 1024 * 4));
    CK(hipMemset(da, 0, 4096));
    CK(hipMemset(db, 0, 4096));
    uint32_t *sig = nullptr;
    hipError_t se = hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory);
    if (se != hipSuccess) { printf("signal memory: %s\n", hipGetErrorString(se)); sig = nullptr; }
    else CK(hipMemset(sig, 0, 8));
    std::vector<hipEvent_t> ev(N);
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));
    uint32_t counter = 0;
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 4; ++mode) {
            if (mode == 2 && !sig) continue;
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(t0, a));
            for (int i = 0; i < N; ++i) {
                hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, a, da, ITERS);
                if (mode == 1) {
                    CK(hipEventRecord(ev[i], a));
                    CK(hipStreamWaitEvent(b, ev[i], 0));
                    hipLaunchKernelGGL(spin, dim3(8), dim3(256), 0, b, db, ITERS / 4);
                } else if (mode == 2) {
                    ++counter;
                    CK(hipStreamWriteValue32(a, sig, counter, 0));
                    CK(hipStreamWaitValue32(b, sig, counter, hipStreamWaitValueGte, 0xffffffffu));
                    hipLaunchKernelGGL(spin, dim3(8), dim3(256), 0, b, db, ITERS / 4);
                } else if (mode == 3) {
                    CK(hipEventRecord(ev[i], a));
                }
            }
            CK(hipEventRecord(t1, a));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, t0, t1));
            const char *names[4] = {"kernels only", "event record + wait", "write value + wait value", "event record, no waiter"};
            printf("%-26s %8.1f us per kernel (%d kernels)\n", names[mode], 1e3 * ms / N, N);
        }
    return 0;
}
