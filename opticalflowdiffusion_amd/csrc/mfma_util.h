// Small device helpers shared by the MFMA kernels (gfx950).
#pragma once
#include "common.h"

namespace ofd {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// Two transposing LDS reads (ds_read_b64_tr_b16) = one MFMA 32x32x16 operand fragment whose k index runs
// over LDS ROWS: rows are `row_stride_bytes` apart (pixels of an NHWC tile), the 32 operand rows/cols are
// 32 consecutive bf16 of a row.  Lane (m = lane & 31, half = lane >> 5) gets rows 8*half .. 8*half+7 of
// column m, starting at `base_row0`.
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* base_row0, int row_stride_bytes, int lane) {
    // lane -> (16-lane group: column block cb, k half h), (q, p) inside the group
    const int li = lane & 15, q = li >> 2, p = li & 3, cb = (lane >> 4) & 1, h = lane >> 5;
    const unsigned char* a = base_row0 + (size_t)(8 * h + q) * row_stride_bytes + (cb * 16 + 4 * p) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + 4 * row_stride_bytes));
    // (whole-vector reinterpretation: per-element bit_casts of the builtin's result are miscompiled by
    //  hipcc 7.2 into a splat of element 0 -- found with tools/probe/tr_probe2.hip)
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
}

}  // namespace ofd
