// Shared host-side helpers for libssm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <tuple>
#include <type_traits>
#include <utility>

#include "../../include/ssm_hip.h"

namespace ssm {

void set_error(const char *fmt, ...);

inline bool aligned16(const void *p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return SSM_E_LAUNCH;
    }
    return SSM_OK;
}

// Opt a kernel into more than 64 KiB of dynamic LDS.  The attribute belongs to the (kernel, DEVICE) pair, so the guard is a bit per
// device of the calling thread's current device, kept by the call site (one `static std::atomic<uint64_t>` per kernel instantiation):
// a process that drives several GPUs from several threads - torch.nn.DataParallel's replica threads, scripts/main.py:74-76 of the
// reference - opts every device in on its first launch there.  Two threads racing on one device both set the (idempotent) attribute.
inline hipError_t reserve_lds(std::atomic<uint64_t> &done, const void *kernel, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev > 63) return hipErrorInvalidDevice;
    const uint64_t bit = 1ull << dev;
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

// split-K switches of the two plan functions (ssm_wino_splitk_plan, ssm_conv_splitk_plan): initialised from $SSM_WINO_SPLITK / $SSM_CONV_SPLITK
// on first use, settable at run time through ssm_splitk_enable (tests A/B the reordered sums in one process)
std::atomic<int> &splitk_switch(int which);          // 0: Winograd form, 1: direct form; value -1 = not initialised yet

// ---- launch programs (ssm_program.cpp; include/ssm_hip.h "launch programs") --------------------------------------------------------
// Every kernel launch of the library goes through ssm::launch (SSM_LAUNCH): the kernel's arguments are converted to its parameter types,
// launched with hipLaunchKernel - what hipLaunchKernelGGL does - and, while a program records, appended to it as a node (host function,
// grid, block, dynamic LDS, stream slot, a copy of the argument values).  ssm_program_run replays the nodes with the same call.
struct Recorder;
extern std::atomic<Recorder *> g_recorder;          // process-wide: a program records one single-threaded warm-up pass
void record_kernel(Recorder *r, const void *fn, dim3 grid, dim3 block, unsigned lds, hipStream_t st, void *const *args, const size_t *sizes,
                   const size_t *aligns, int n);
void record_memset(Recorder *r, void *dst, int value, size_t bytes, hipStream_t st);

template <class... KArgs, class... Args, size_t... I>
inline void launch_impl(void (*kern)(KArgs...), dim3 grid, dim3 block, unsigned lds, hipStream_t st, std::index_sequence<I...>, Args &&...args) {
    std::tuple<std::decay_t<KArgs>...> vals{static_cast<std::decay_t<KArgs>>(std::forward<Args>(args))...};
    void *ptrs[sizeof...(KArgs) + 1] = {static_cast<void *>(&std::get<I>(vals))..., nullptr};
    (void)hipLaunchKernel(reinterpret_cast<const void *>(kern), grid, block, ptrs, lds, st);          // (errors: hipGetLastError in check_launch)
    if (Recorder *r = g_recorder.load(std::memory_order_acquire)) {
        static const size_t sizes[sizeof...(KArgs) + 1] = {sizeof(std::decay_t<KArgs>)..., 0};
        static const size_t aligns[sizeof...(KArgs) + 1] = {alignof(std::decay_t<KArgs>)..., 0};
        record_kernel(r, reinterpret_cast<const void *>(kern), grid, block, lds, st, ptrs, sizes, aligns, (int)sizeof...(KArgs));
    }
}
template <class... KArgs, class... Args>
inline void launch(void (*kern)(KArgs...), dim3 grid, dim3 block, unsigned lds, hipStream_t st, Args &&...args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel argument count");
    launch_impl(kern, grid, block, lds, st, std::index_sequence_for<KArgs...>{}, std::forward<Args>(args)...);
}
inline hipError_t memset_async(void *dst, int value, size_t bytes, hipStream_t st) {
    const hipError_t e = hipMemsetAsync(dst, value, bytes, st);
    if (Recorder *r = g_recorder.load(std::memory_order_acquire)) record_memset(r, dst, value, bytes, st);
    return e;
}
#define SSM_LAUNCH(...) ssm::launch(__VA_ARGS__)

#define SSM_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            ssm::set_error(__VA_ARGS__); \
            return SSM_E_ARG;           \
        }                               \
    } while (0)

}  // namespace ssm

// XCD-aware tile order.  The dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs (each with its own
// L2), so in launch order neighbouring tiles - which share input halos, and for Cout > BN the whole input patch - sit
// on different L2s.  This maps workgroup i to the logical tile index such that every XCD walks one contiguous
// 1/8 of the (cout block, x tile, y tile, batch) sequence.
#ifndef SSM_XCD_REMAP
#define SSM_XCD_REMAP 1
#endif
__device__ __forceinline__ int ssm_xcd_tile(int i, int n) {
#if SSM_XCD_REMAP
    const int xcd = i & 7, local = i >> 3;
    const int per = n >> 3, rem = n & 7;
    return xcd < rem ? xcd * (per + 1) + local : rem * (per + 1) + (xcd - rem) * per + local;
#else
    return i;
#endif
}
