// Shared host-side helpers for libssm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

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
