// Shared host-side helpers for libssm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
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

#define SSM_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            ssm::set_error(__VA_ARGS__); \
            return SSM_E_ARG;           \
        }                               \
    } while (0)

}  // namespace ssm
