// Backward of the two attention cores (training path):
//   * LinearAttention core: la_core.hip (forward and backward);
//   * mid-block softmax attention, DD:256-268: flash-style recompute with MFMA 32x32x16 bf16, one
//     kernel that owns query blocks (dQ) and one that owns key blocks (dK, dV) -- no atomics.
// qkv / dqkv are [B][n][384] bf16 (q | k | v, head-major 4 x 32), attention outputs [B][n][128].
#include "blocks.h"

namespace ofd {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ void a_unpack8(const uint4& v, float* f) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[2 * j] = bf2f((bf16_t)(w[j] & 0xffffu));
        f[2 * j + 1] = bf2f((bf16_t)(w[j] >> 16));
    }
}
__device__ __forceinline__ uint4 a_pack8(const float* f) {
    return make_uint4(f2bf2(f[0], f[1]), f2bf2(f[2], f[3]), f2bf2(f[4], f[5]), f2bf2(f[6], f[7]));
}
__device__ __forceinline__ void load32(const bf16_t* p, float (&f)[32]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) a_unpack8(*(const uint4*)(p + u * 8), &f[u * 8]);
}
__device__ __forceinline__ void store32(bf16_t* p, const float (&f)[32]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) *(uint4*)(p + u * 8) = a_pack8(&f[u * 8]);
}

// =====================================================================================================
// softmax attention backward, d = 32
// D[bh][q] = sum_d dO[q][d] * O[q][d]
__global__ void __launch_bounds__(256) fa_bwd_delta_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout, float* __restrict__ delta, int n, size_t total) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {   // i = pixel*4 + h
        const size_t pix = i >> 2;
        const int h = (int)(i & 3);
        float a[32], g[32];
        load32(o + pix * 128 + h * 32, a);
        load32(dout + pix * 128 + h * 32, g);
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 32; ++j) s += a[j] * g[j];
        const size_t b = pix / n, q = pix % n;
        delta[(b * 4 + h) * n + q] = s;
    }
}

constexpr int FB_T = 64;      // tile of the streamed side
// permuted slot of element kk (0..31) inside a 32-block so that a 16-byte LDS read yields the 8 k-indices
// one lane of the next MFMA's accumulator-derived operand holds: kk = 16 s + 8 a + 4 hh + bb -> s*16 + hh*8 + a*4 + bb
__device__ __forceinline__ int perm_slot(int idx) {
    const int kb = idx >> 5, kk = idx & 31, s = kk >> 4, a = (kk >> 3) & 1, hh = (kk >> 2) & 1, bb = kk & 3;
    return kb * 32 + s * 16 + hh * 8 + a * 4 + bb;
}
__device__ __forceinline__ void stage_transposed(unsigned char* lds, int u, int slot, const uint4& v) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        *(bf16_t*)(lds + (u * 8 + 2 * j) * 144 + slot * 2) = (bf16_t)(w[j] & 0xffffu);
        *(bf16_t*)(lds + (u * 8 + 2 * j + 1) * 144 + slot * 2) = (bf16_t)(w[j] >> 16);
    }
}
__device__ __forceinline__ bf16x8 acc_to_bf16(const f32x16& a, int s) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (__bf16)a[8 * s + j];
    return f;
}

// dQ: workgroup = 4 waves x 32 queries of one (sample, head); streams key tiles.  grid (ceil(n/128), B*4)
__global__ void __launch_bounds__(256) fa_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                        const float* __restrict__ delta, bf16_t* __restrict__ dqkv, int n, float scale) {
    __shared__ __attribute__((aligned(16))) unsigned char k_lds[FB_T * 80], v_lds[FB_T * 80], kt_lds[32 * 144];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int bh = blockIdx.y, b = bh / 4, h = bh % 4;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const bf16_t* base = qkv + (size_t)b * n * 384;
    const int qi = min(q0 + l31, n - 1);
    bf16x8 qf[2], gf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        qf[s] = *(const bf16x8*)(base + (size_t)qi * 384 + h * 32 + s * 16 + half * 8);
        gf[s] = *(const bf16x8*)(dout + ((size_t)b * n + qi) * 128 + h * 32 + s * 16 + half * 8);
    }
    // P = exp(scale s - lse) as one fma + v_exp (c = scale log2 e, lse in log2 units); dS = P (dP - delta) scale as P * fma(dP, scale, -delta scale)
    const float c2 = scale * 1.4426950408889634f;
    const float lse_q = lse[(size_t)bh * n + qi] * 1.4426950408889634f, dl_q = delta[(size_t)bh * n + qi] * scale;
    f32x16 dq_acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) dq_acc[r] = 0.0f;
    for (int j0 = 0; j0 < n; j0 += FB_T) {
        __syncthreads();
        {
            const int key = tid >> 2, u = tid & 3;
            const int kg = min(j0 + key, n - 1);
            const uint4 kv = *(const uint4*)(base + (size_t)kg * 384 + 128 + h * 32 + u * 8);
            const uint4 vv = *(const uint4*)(base + (size_t)kg * 384 + 256 + h * 32 + u * 8);
            *(uint4*)(k_lds + key * 80 + u * 16) = kv;
            *(uint4*)(v_lds + key * 80 + u * 16) = vv;
            stage_transposed(kt_lds, u, perm_slot(key), kv);
        }
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 s_acc, p_acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s_acc[r] = 0.0f; p_acc[r] = 0.0f; }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 kf = *(const bf16x8*)(k_lds + (kb * 32 + l31) * 80 + (s * 16 + half * 8) * 2);
                const bf16x8 vf = *(const bf16x8*)(v_lds + (kb * 32 + l31) * 80 + (s * 16 + half * 8) * 2);
                s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], s_acc, 0, 0, 0);     // S^T[key][q]
                p_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, gf[s], p_acc, 0, 0, 0);     // dP^T[key][q]
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
                s_acc[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[r], c2, -lse_q)) * __builtin_fmaf(p_acc[r], scale, -dl_q);      // dS^T
            if (j0 + FB_T > n) {                     // keys past the end exist in the last tile only
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (j0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half >= n) s_acc[r] = 0.0f;
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 ktf = *(const bf16x8*)(kt_lds + l31 * 144 + (kb * 32 + s * 16 + half * 8) * 2);
                dq_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, acc_to_bf16(s_acc, s), dq_acc, 0, 0, 0);   // dQ^T[d][q]
            }
        }
    }
    const int q = q0 + l31;
    if (q < n) {
        bf16_t* dst = dqkv + ((size_t)b * n + q) * 384 + h * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *(uint2*)(dst + 8 * g + 4 * half) = make_uint2(f2bf2(dq_acc[4 * g], dq_acc[4 * g + 1]), f2bf2(dq_acc[4 * g + 2], dq_acc[4 * g + 3]));
    }
}

// dK, dV: workgroup = 4 waves x 32 keys of one (sample, head); streams query tiles.
__global__ void __launch_bounds__(256) fa_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                         const float* __restrict__ delta, bf16_t* __restrict__ dqkv, int n, float scale) {
    __shared__ __attribute__((aligned(16))) unsigned char q_lds[FB_T * 80], g_lds[FB_T * 80], qt_lds[32 * 144], gt_lds[32 * 144];
    __shared__ float lse_s[FB_T], dl_s[FB_T];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int bh = blockIdx.y, b = bh / 4, h = bh % 4;
    const int k0 = blockIdx.x * 128 + wave * 32;
    const bf16_t* base = qkv + (size_t)b * n * 384;
    const int ki = min(k0 + l31, n - 1);
    bf16x8 kf[2], vf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        kf[s] = *(const bf16x8*)(base + (size_t)ki * 384 + 128 + h * 32 + s * 16 + half * 8);
        vf[s] = *(const bf16x8*)(base + (size_t)ki * 384 + 256 + h * 32 + s * 16 + half * 8);
    }
    f32x16 dk_acc, dv_acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk_acc[r] = 0.0f; dv_acc[r] = 0.0f; }
    const float c2 = scale * 1.4426950408889634f;      // (as in the dQ kernel; lse_s / dl_s are staged in log2 / scaled units)
    for (int i0 = 0; i0 < n; i0 += FB_T) {
        __syncthreads();
        {
            const int qq = tid >> 2, u = tid & 3;
            const bool ok = i0 + qq < n;
            const int qg = min(i0 + qq, n - 1);
            const uint4 qv = *(const uint4*)(base + (size_t)qg * 384 + h * 32 + u * 8);
            uint4 gv = *(const uint4*)(dout + ((size_t)b * n + qg) * 128 + h * 32 + u * 8);
            if (!ok) gv = make_uint4(0u, 0u, 0u, 0u);                      // rows past n contribute nothing
            *(uint4*)(q_lds + qq * 80 + u * 16) = qv;
            *(uint4*)(g_lds + qq * 80 + u * 16) = gv;
            const int slot = perm_slot(qq);
            stage_transposed(qt_lds, u, slot, qv);
            stage_transposed(gt_lds, u, slot, gv);
            if (u == 0) {
                lse_s[qq] = ok ? lse[(size_t)bh * n + qg] * 1.4426950408889634f : 3.0e38f;      // (log2 units; rows past n: P = 0)
                dl_s[qq] = ok ? delta[(size_t)bh * n + qg] * scale : 0.0f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x16 s_acc, p_acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s_acc[r] = 0.0f; p_acc[r] = 0.0f; }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 qfr = *(const bf16x8*)(q_lds + (qb * 32 + l31) * 80 + (s * 16 + half * 8) * 2);
                const bf16x8 gfr = *(const bf16x8*)(g_lds + (qb * 32 + l31) * 80 + (s * 16 + half * 8) * 2);
                s_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr, kf[s], s_acc, 0, 0, 0);    // S[q][key]
                p_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gfr, vf[s], p_acc, 0, 0, 0);    // dP[q][key]
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qq = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[r], c2, -lse_s[qq]));
                s_acc[r] = p;
                p_acc[r] = p * __builtin_fmaf(p_acc[r], scale, -dl_s[qq]);                           // dS
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 gtf = *(const bf16x8*)(gt_lds + l31 * 144 + (qb * 32 + s * 16 + half * 8) * 2);
                const bf16x8 qtf = *(const bf16x8*)(qt_lds + l31 * 144 + (qb * 32 + s * 16 + half * 8) * 2);
                dv_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gtf, acc_to_bf16(s_acc, s), dv_acc, 0, 0, 0);   // dV^T[d][key]
                dk_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, acc_to_bf16(p_acc, s), dk_acc, 0, 0, 0);   // dK^T[d][key]
            }
        }
    }
    const int key = k0 + l31;
    if (key < n) {
        bf16_t* dst = dqkv + ((size_t)b * n + key) * 384 + h * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *(uint2*)(dst + 128 + 8 * g + 4 * half) = make_uint2(f2bf2(dk_acc[4 * g], dk_acc[4 * g + 1]), f2bf2(dk_acc[4 * g + 2], dk_acc[4 * g + 3]));
            *(uint2*)(dst + 256 + 8 * g + 4 * half) = make_uint2(f2bf2(dv_acc[4 * g], dv_acc[4 * g + 1]), f2bf2(dv_acc[4 * g + 2], dv_acc[4 * g + 3]));
        }
    }
}

int k_flash_attention_bwd(const bf16_t* qkv, const bf16_t* o, const bf16_t* dout, const float* lse, bf16_t* dqkv, float* delta, int B, int n,
                          hipStream_t s) {
    const size_t total = (size_t)B * n * 4;
    size_t gb = (total + 255) / 256;
    if (gb > 4096) gb = 4096;
    const float scale = 0.17677669529663687f;
    fa_bwd_delta_kernel<<<(int)gb, 256, 0, s>>>(o, dout, delta, n, total);
    fa_bwd_dq_kernel<<<dim3(cdiv(n, 128), B * 4), 256, 0, s>>>(qkv, dout, lse, delta, dqkv, n, scale);
    fa_bwd_dkv_kernel<<<dim3(cdiv(n, 128), B * 4), 256, 0, s>>>(qkv, dout, lse, delta, dqkv, n, scale);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

}  // namespace ofd
using namespace ofd;

extern "C" size_t ofd_la_workspace_floats(int B, int n) { return (size_t)B * 4 * (size_t)la_fwd_parts(B, n) * 1088; }
extern "C" size_t ofd_la_bwd_workspace_floats(int B, int n) { return la_bwd_workspace_floats(B, n); }
extern "C" int ofd_linear_attention_core(const void* qkv, void* out, float* ctx, float* ml, float* workspace, int B, int n, void* stream) {
    OFD_CHECK_ARG(qkv && out && ctx && ml && workspace && B > 0 && n > 0, "linear_attention_core: bad argument");
    return k_linear_attention_core((const bf16_t*)qkv, workspace, ctx, (bf16_t*)out, B, n, (hipStream_t)stream, ml);
}
extern "C" int ofd_linear_attention_core_backward(const void* qkv, const void* dout, const float* ctx, const float* ml, void* dqkv, float* workspace,
                                                  int B, int n, void* stream) {
    OFD_CHECK_ARG(qkv && dout && ctx && ml && dqkv && workspace && B > 0 && n > 0, "linear_attention_core_backward: bad argument");
    return k_linear_attention_core_bwd((const bf16_t*)qkv, (const bf16_t*)dout, ctx, ml, (bf16_t*)dqkv, workspace, B, n, (hipStream_t)stream);
}
extern "C" int ofd_flash_attention(const void* qkv, void* out, float* lse, int B, int n, void* stream) {
    OFD_CHECK_ARG(qkv && out && B > 0 && n > 0, "flash_attention: bad argument");
    return k_flash_attention((const bf16_t*)qkv, (bf16_t*)out, B, n, (hipStream_t)stream, lse);
}
extern "C" int ofd_flash_attention_backward(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* delta, int B,
                                            int n, void* stream) {
    OFD_CHECK_ARG(qkv && out && dout && lse && dqkv && delta && B > 0 && n > 0, "flash_attention_backward: bad argument");
    return k_flash_attention_bwd((const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, (bf16_t*)dqkv, delta, B, n, (hipStream_t)stream);
}
