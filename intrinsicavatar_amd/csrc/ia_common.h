// ia_common.h -- shared helpers for the gfx950 kernels of libia_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ia_amd.h"

#define IA_EXPORT extern "C" __attribute__((visibility("default")))

namespace ia {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return IA_ERR_LAUNCH;
    }
    return IA_OK;
}

constexpr int WAVE = 64;

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace ia

#define IA_REQUIRE(cond, msg)                    \
    do {                                         \
        if (!(cond)) {                           \
            ia::set_error("%s: %s", __func__, msg); \
            return IA_ERR_INVALID;               \
        }                                        \
    } while (0)
