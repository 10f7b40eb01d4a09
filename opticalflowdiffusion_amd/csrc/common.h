// Shared host-side helpers of libofd_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/ofd.h"

namespace ofd {

void set_error(const char* fmt, ...);

#define OFD_CHECK_ARG(cond, ...)                                   \
    do {                                                           \
        if (!(cond)) {                                             \
            ::ofd::set_error(__VA_ARGS__);                         \
            return OFD_ERR_ARG;                                    \
        }                                                          \
    } while (0)

#define OFD_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            ::ofd::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return OFD_ERR_HIP;                                                            \
        }                                                                                  \
    } while (0)

#define OFD_LAUNCH_CHECK() OFD_HIP(hipGetLastError())

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

typedef uint16_t bf16_t;   // raw bits

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even; NaN stays NaN (plain cast semantics)
__device__ __forceinline__ bf16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

}  // namespace ofd
