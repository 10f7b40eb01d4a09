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
// round-to-nearest-even, NaN stays NaN: the plain cast lowers to v_cvt_pk_bf16_f32 on gfx950
// (a hand-written integer rounding costs a divergent branch per value)
__device__ __forceinline__ bf16_t f2bf(float f) {
    const __bf16 h = (__bf16)f;
    return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ uint32_t f2bf2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    const f32x2_t v = {lo, hi};
    const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
    return __builtin_bit_cast(uint32_t, h);
}

}  // namespace ofd
