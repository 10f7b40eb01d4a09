// Fused LinearAttention block for gfx950 (denoising_diffusion.py:216-244 inside
// Residual(PreNorm(.)), DD:81-87, DD:127-135), C in {64, 128}.
//
// The reference materialises LayerNorm(x), qkv (6x the activation), the head outputs and the
// to_out result: ~2.9 KB of HBM traffic per pixel.  Here the block is two streaming passes over x
// (384 B per pixel) with every contraction on MFMA 32x32x16 bf16 and no intermediate in HBM:
//   pass 1  x -> LN -> [k|v] = Wkv x^  -> online softmax of k over pixels -> ctx^T += v^T p
//           (the k/v accumulator tiles are used directly as the operands of the context MFMA:
//            accumulator rows = pixels = the contraction index, no LDS transpose)
//   pass 2  x -> LN -> q = Wq x^ -> softmax over d -> out = ctx^T q -> o = Wout out + b
//           -> LN -> + x   (each stage's accumulator tile is the next MFMA's B operand)
// The PreNorm gain g is folded into Wq / Wkv (W.diag(g)) when the weights are prepared.
#include <cstdlib>
#include "blocks.h"

namespace ofd {

int la_fused_blocks(int n, int B);

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ uint32_t la_pack2(float a, float b) { return f2bf2(a, b); }
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// registers 8*s2 .. 8*s2+7 of an accumulator tile as an MFMA operand fragment (k = tile row)
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& a, int s2) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (__bf16)a[8 * s2 + j];
    return f;
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
template <int C>
struct XRaw {
    u32x4 v[C / 16];
};

// raw x row of this lane's pixel: channels 16 s + 8 half .. + 7 for every k-step s
template <int C>
__device__ __forceinline__ void load_raw_x(const bf16_t* __restrict__ xrow, XRaw<C>& r, int half) {
#pragma unroll
    for (int s = 0; s < C / 16; ++s) r.v[s] = *(const u32x4*)(xrow + 16 * s + 8 * half);
}

// raw tile -> normalised (no gain) bf16 fragments (LayerNorm over channels, DD:121-125).  r03: the two sums run on the PACKED words
// (v_dot2_f32_bf16: x . (1, 1) and x . x, two channels per instruction, fp32 accumulate) and the variance is E[x^2] - mean^2 -- the
// inputs are bf16 activations, |mean| / std stays far below the 2^12 at which fp32 cancellation would reach bf16 resolution -- so an
// element costs unpack + one fma + convert instead of unpack + add + subtract + fma + multiply + convert (OFD_LA_TWO_PASS_LN=1 at
// build time restores the centred form).
#ifndef OFD_LA_TWO_PASS_LN
#define OFD_LA_TWO_PASS_LN 0
#endif
typedef __bf16 la_bf16x2 __attribute__((ext_vector_type(2)));
template <int C>
__device__ __forceinline__ void norm_x(const XRaw<C>& r, float eps, bf16x8 (&xs)[C / 16]) {
    constexpr int KS = C / 16;
#if OFD_LA_TWO_PASS_LN
    float v[KS][8];
    float sum = 0.0f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t w = r.v[s][j];
            v[s][2 * j] = bf2f((bf16_t)(w & 0xffffu));
            v[s][2 * j + 1] = bf2f((bf16_t)(w >> 16));
            sum += v[s][2 * j] + v[s][2 * j + 1];
        }
    }
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.0f / C);
    float q = 0.0f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[s][j] -= mean;
            q += v[s][j] * v[s][j];
        }
    q += __shfl_xor(q, 32, 64);
    const float rstd = rsqrtf(q * (1.0f / C) + eps);
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) xs[s][j] = (__bf16)(v[s][j] * rstd);
#else
    const la_bf16x2 one = __builtin_bit_cast(la_bf16x2, 0x3f803f80u);
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const la_bf16x2 w = __builtin_bit_cast(la_bf16x2, (uint32_t)r.v[s][j]);
            s1 = __builtin_amdgcn_fdot2_f32_bf16(w, one, s1, false);
            s2 = __builtin_amdgcn_fdot2_f32_bf16(w, w, s2, false);
        }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    const float mean = s1 * (1.0f / C);
    const float var = fmaxf(__builtin_fmaf(-mean, mean, s2 * (1.0f / C)), 0.0f);
    const float rstd = rsqrtf(var + eps), off = -mean * rstd;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t w = r.v[s][j];
            xs[s][2 * j] = (__bf16)__builtin_fmaf(bf2f((bf16_t)(w & 0xffffu)), rstd, off);
            xs[s][2 * j + 1] = (__bf16)__builtin_fmaf(bf2f((bf16_t)(w >> 16)), rstd, off);
        }
#endif
}

// Training form: the fragments ARE the LayerNorm output xn = x^ * g rounded to bf16 (what the unfused path stores and multiplies by the
// plain weights, and what the backward's q recompute / to_qkv weight gradient read) -- g: the PreNorm gain, fp32 [C] in LDS.
template <int C>
__device__ __forceinline__ void norm_x_gain(const XRaw<C>& r, float eps, const float* __restrict__ g, int half, bf16x8 (&xs)[C / 16]) {
    constexpr int KS = C / 16;
    const la_bf16x2 one = __builtin_bit_cast(la_bf16x2, 0x3f803f80u);
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const la_bf16x2 w = __builtin_bit_cast(la_bf16x2, (uint32_t)r.v[s][j]);
            s1 = __builtin_amdgcn_fdot2_f32_bf16(w, one, s1, false);
            s2 = __builtin_amdgcn_fdot2_f32_bf16(w, w, s2, false);
        }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    const float mean = s1 * (1.0f / C);
    const float var = fmaxf(__builtin_fmaf(-mean, mean, s2 * (1.0f / C)), 0.0f);
    const float rstd = rsqrtf(var + eps), off = -mean * rstd;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const float4 g0 = *(const float4*)(g + 16 * s + 8 * half), g1 = *(const float4*)(g + 16 * s + 8 * half + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t w = r.v[s][j];
            xs[s][2 * j] = (__bf16)(__builtin_fmaf(bf2f((bf16_t)(w & 0xffffu)), rstd, off) * gg[2 * j]);
            xs[s][2 * j + 1] = (__bf16)(__builtin_fmaf(bf2f((bf16_t)(w >> 16)), rstd, off) * gg[2 * j + 1]);
        }
    }
}

// accumulator tile [rows = 32 channels][col = this lane's pixel] -> bf16, 16-byte stores: pairs of register quads are exchanged between the
// half-waves (v_permlane32_swap) so that a lane stores 8 consecutive channels of its pixel; `row` = that pixel's first channel of the tile
__device__ __forceinline__ void store_acc_rows(const f32x16& a, bf16_t* __restrict__ row, int half, bool ok) {
#pragma unroll
    for (int g = 0; g < 4; g += 2) {
        const uint2 q0 = make_uint2(la_pack2(a[4 * g], a[4 * g + 1]), la_pack2(a[4 * g + 2], a[4 * g + 3]));
        const uint2 q1 = make_uint2(la_pack2(a[4 * g + 4], a[4 * g + 5]), la_pack2(a[4 * g + 6], a[4 * g + 7]));
        const auto rx = __builtin_amdgcn_permlane32_swap(q0.x, q1.x, false, false);
        const auto ry = __builtin_amdgcn_permlane32_swap(q0.y, q1.y, false, false);
        if (ok) *(uint4*)(row + 8 * g + 8 * half) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
    }
}

// ---- end of pass 1: the workgroup's four waves merge their online-softmax states (m, l, ctx relative to m) of one head through LDS and
// write ONE part: a quarter of the partial traffic, and a grid of 256+ workgroups (small batches) stays within the combine's 256 parts.
// Part format unchanged: [m 32 | l 32 | ctx[d][e] 32 x 32].  LDS: 4 x LA_PART_PITCH floats (rows padded to 36 floats).
constexpr int LA_PART_PITCH = 64 + 32 * 36;
constexpr float LOG2E = 1.4426950408889634f;
__device__ __forceinline__ void la_store_part(float* lds, float* __restrict__ o, float m, float l, const f32x16& ctxT, int tid, int wave, int l31,
                                              int half) {
    float* mine = lds + wave * LA_PART_PITCH;
    const float l_tot = l + __shfl_xor(l, 32, 64);
    if (half == 0) {
        mine[l31] = m;
        mine[32 + l31] = l_tot;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)   // lane = d (column), registers = e rows
        *(float4*)(mine + 64 + l31 * 36 + 8 * g + 4 * half) = make_float4(ctxT[4 * g], ctxT[4 * g + 1], ctxT[4 * g + 2], ctxT[4 * g + 3]);
    __syncthreads();
    const int d = tid >> 3, e0 = (tid & 7) * 4;
    float mw[4], M = -3.0e38f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        mw[w] = lds[w * LA_PART_PITCH + d];
        M = fmaxf(M, mw[w]);
    }
    float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float ls = 0.0f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float wt = __expf(mw[w] - M);
        const float4 v = *(const float4*)(lds + w * LA_PART_PITCH + 64 + d * 36 + e0);
        a.x += v.x * wt;
        a.y += v.y * wt;
        a.z += v.z * wt;
        a.w += v.w * wt;
        ls += lds[w * LA_PART_PITCH + 32 + d] * wt;
    }
    *(float4*)(o + 64 + d * 32 + e0) = a;
    if ((tid & 7) == 0) {
        o[d] = M;
        o[32 + d] = ls;
    }
    __syncthreads();
}

// ---- pass 1 ---------------------------------------------------------------------------------------
// TRAIN (the training forward, C = 64): the pass also WRITES what the backward reads -- xn (LayerNorm output with the gain, bf16) and k | v
// (channels 128 .. 383 of the [pixel][384] qkv tensor) -- so the LayerNorm kernel, the to_qkv conv and the read of k, v by la_ctx_stored_kernel
// disappear (11 -> 6 tensor passes of 64 channels).  Its operands are those of the unfused path: xn rounded to bf16 times the PLAIN weights
// (wkv: not folded with the gain), k rounded to bf16 before the softmax.  k, v leave through a second projection with the operands swapped
// (rows = channel, col = pixel: a lane owns whole 8-channel units of its pixel and stores 16 bytes); the MFMA pipe has the room.
template <int C, bool TRAIN>
__global__ void __launch_bounds__(256, 2) la_ctx_fused_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wkv,
                                                           float* __restrict__ partial, int n, float eps, float defer,
                                                           const float* __restrict__ g, bf16_t* __restrict__ xn_out, bf16_t* __restrict__ qkv_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KS = C / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    for (int i = tid; i < (C / 8) * 256; i += 256) ((uint4*)smem)[i] = ((const uint4*)wkv)[i];
    float* s_g = (float*)(smem + (C / 8) * 256 * 16);
    if constexpr (TRAIN)
        for (int i = tid; i < C; i += 256) s_g[i] = g[i];
    __syncthreads();
    const int b = blockIdx.y, wave_id = blockIdx.x * 4 + wave, nw = gridDim.x * 4, ntiles = (n + 31) / 32;

    f32x16 ctxT[4];
    float m[4], l[4];
#pragma unroll
    for (int hd = 0; hd < 4; ++hd) {
        m[hd] = -3.0e38f;
        l[hd] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) ctxT[hd][r] = 0.0f;
    }
    const bf16_t* xb = x + (size_t)b * n * C;
    XRaw<C> raw_next;
    load_raw_x<C>(xb + (size_t)min(wave_id * 32 + l31, n - 1) * C, raw_next, half);
    for (int tile = wave_id; tile < ntiles; tile += nw) {
        // the weight fragments are loop-invariant LDS reads: without this the compiler hoists all of
        // them into registers (C=128: 512 VGPRs + scratch spills) instead of re-reading LDS per tile
        asm volatile("" ::: "memory");
        const XRaw<C> raw = raw_next;
        // software prefetch of the next tile's rows (clamped: the last prefetch re-reads a valid row)
        load_raw_x<C>(xb + (size_t)min((tile + nw) * 32 + l31, n - 1) * C, raw_next, half);
        bf16x8 xs[KS];
        const int pix = tile * 32 + l31;
        if constexpr (TRAIN) {
            norm_x_gain<C>(raw, eps, s_g, half, xs);
            if (pix < n) {
                bf16_t* xrow = xn_out + ((size_t)b * n + pix) * C;
#pragma unroll
                for (int s = 0; s < KS; ++s) *(bf16x8*)(xrow + 16 * s + 8 * half) = xs[s];
            }
        } else {
            norm_x<C>(raw, eps, xs);
        }
        if constexpr (TRAIN) {
            // the stored k | v: the same products with rows = channel, col = pixel, one 32-channel block at a time (16 live accumulator registers)
            bf16_t* qrow = qkv_out + ((size_t)b * n + min(pix, n - 1)) * 384 + 128;
#pragma unroll
            for (int blk = 0; blk < 8; ++blk) {
                f32x16 t;
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = 0.0f;
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(smem + ((size_t)(2 * s + half) * 256 + blk * 32 + l31) * 16), xs[s], t, 0, 0, 0);
                store_acc_rows(t, qrow + blk * 32, half, pix < n);
            }
        }
#pragma unroll
        for (int hd = 0; hd < 4; ++hd) {
            f32x16 ka, va;
#pragma unroll
            for (int r = 0; r < 16; ++r) { ka[r] = 0.0f; va[r] = 0.0f; }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 wk = *(const bf16x8*)(smem + ((size_t)(2 * s + half) * 256 + hd * 32 + l31) * 16);
                const bf16x8 wv = *(const bf16x8*)(smem + ((size_t)(2 * s + half) * 256 + 128 + hd * 32 + l31) * 16);
                ka = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xs[s], wk, ka, 0, 0, 0);   // rows = pixels, col = d
                va = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xs[s], wv, va, 0, 0, 0);   // rows = pixels, col = e
            }
            if constexpr (TRAIN) {
#pragma unroll
                for (int r = 0; r < 16; ++r) ka[r] = bf2f(f2bf(ka[r]));      // the softmax sees the stored (bf16) k, as the backward will
            }
            float mt = -3.0e38f;
            if (tile * 32 + 32 > n) {            // (only a sample's last tile has rows past its end: 32 compares + selects per head otherwise)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (tile * 32 + acc_row(r, half) >= n) ka[r] = -3.0e38f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, ka[r]);
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            // deferred running maximum: the reference point m moves only when a tile's maximum exceeds it by more than `defer`
            // (natural-log units; 0 = always): p = exp(k - m) then stays below e^defer (2^8 by default: exact range for fp32 sums and
            // bf16 operands alike) and the rescaling of l and ctx^T -- 16 multiplies + an exp per head and tile -- runs only in the
            // tiles where some column's reference moves (wave-uniform skip).  (m, l, ctx) stay a consistent triple: the combine
            // kernel only needs sums relative to the m it is handed.
            const bool move = mt > m[hd] + defer;
            if (__builtin_amdgcn_ballot_w64(move)) {
                const float m_new = move ? mt : m[hd];
                const float f = __expf(m[hd] - m_new);
                m[hd] = m_new;
                l[hd] *= f;
#pragma unroll
                for (int r = 0; r < 16; ++r) ctxT[hd][r] *= f;
            }
            float ps = 0.0f;
            const float m2 = m[hd] * LOG2E;      // exp(k - m) as exp2(k log2e - m log2e): one fma + v_exp per element instead of sub, mul, v_exp
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(ka[r], LOG2E, -m2));
                ka[r] = p;
                ps += p;
            }
            l[hd] += ps;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)   // ctx^T[e][d] += sum_pix v[pix][e] p[pix][d]
                ctxT[hd] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(va, s2), acc_frag(ka, s2), ctxT[hd], 0, 0, 0);
        }
    }
    __syncthreads();                          // every wave is done with the weight fragments: the LDS becomes the combine buffer
#pragma unroll
    for (int hd = 0; hd < 4; ++hd)
        la_store_part((float*)smem, partial + ((size_t)(b * 4 + hd) * gridDim.x + blockIdx.x) * 1088, m[hd], l[hd], ctxT[hd], tid, wave, l31, half);
}

// ---- pass 1 on STORED k, v (the training forward: qkv is materialised for the backward): the online-softmax body of the kernel above,
// with the projection MFMAs replaced by a layout change -- a k (v) tile read as an A operand (lane = pixel, 8 consecutive channels: one
// 16-byte load) times the 32 x 32 identity gives the accumulator tile [rows = pixels][col = d] exactly (bf16 x 1.0, fp32 accumulate), which is
// what the context MFMA wants as operands.  Replaces lc_ctx_partial_kernel (la_core.hip), whose per-channel max / exp scans walk an LDS tile
// two bytes at a time: 1.3 ms -> 0.4 ms per full-resolution block.  Same partial format ([m 32 | l 32 | ctx 32 x 32] per (sample, head, part)).
__global__ void __launch_bounds__(256, 2) la_ctx_stored_kernel(const bf16_t* __restrict__ qkv, float* __restrict__ partial, int n) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.y, wave_id = blockIdx.x * 4 + wave, nw = gridDim.x * 4, ntiles = (n + 31) / 32;
    bf16x8 ident[2];                       // B operand of the identity: k index 16 s + 8 half + j, column l31
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 8; ++j) ident[s2][j] = (__bf16)((16 * s2 + 8 * half + j == l31) ? 1.0f : 0.0f);
    f32x16 ctxT[4];
    float m[4], l[4];
#pragma unroll
    for (int hd = 0; hd < 4; ++hd) {
        m[hd] = -3.0e38f;
        l[hd] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) ctxT[hd][r] = 0.0f;
    }
    const bf16_t* qb = qkv + (size_t)b * n * 384;
    // this lane's k (v) units of a tile: head hd, k-step s2 -> channels 128 (256) + 32 hd + 16 s2 + 8 half .. + 7 of pixel tile * 32 + l31.
    // No cross-tile prefetch: with it the kernel needs 265 registers (9 spilled); the second resident workgroup covers the load latency.
    for (int tile = wave_id; tile < ntiles; tile += nw) {
        u32x4 kr[8], vr[8];
        {
            const bf16_t* row = qb + (size_t)min(tile * 32 + l31, n - 1) * 384 + 128 + 8 * half;
#pragma unroll
            for (int i = 0; i < 8; ++i) kr[i] = *(const u32x4*)(row + 16 * i);
#pragma unroll
            for (int i = 0; i < 8; ++i) vr[i] = *(const u32x4*)(row + 128 + 16 * i);
        }
#pragma unroll
        for (int hd = 0; hd < 4; ++hd) {
            f32x16 ka, va;
#pragma unroll
            for (int r = 0; r < 16; ++r) { ka[r] = 0.0f; va[r] = 0.0f; }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                ka = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kr[hd * 2 + s2]), ident[s2], ka, 0, 0, 0);   // rows = pixels, col = d
                va = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vr[hd * 2 + s2]), ident[s2], va, 0, 0, 0);   // rows = pixels, col = e
            }
            float mt = -3.0e38f;
            if (tile * 32 + 32 > n) {            // (only a sample's last tile has rows past its end: 32 compares + selects per head otherwise)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (tile * 32 + acc_row(r, half) >= n) ka[r] = -3.0e38f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, ka[r]);
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            const float m_new = fmaxf(m[hd], mt);
            const float f = __expf(m[hd] - m_new);
            m[hd] = m_new;
            float ps = 0.0f;
            const float m2 = m_new * LOG2E;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = bf2f(f2bf(__builtin_amdgcn_exp2f(__builtin_fmaf(ka[r], LOG2E, -m2))));      // the value the context MFMA multiplies by: the normaliser sums the same
                ka[r] = p;
                ps += p;
            }
            l[hd] = l[hd] * f + ps;
#pragma unroll
            for (int r = 0; r < 16; ++r) ctxT[hd][r] *= f;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)   // ctx^T[e][d] += sum_pix v[pix][e] p[pix][d]
                ctxT[hd] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(va, s2), acc_frag(ka, s2), ctxT[hd], 0, 0, 0);
        }
    }
    __shared__ __attribute__((aligned(16))) float part_s[4 * LA_PART_PITCH];
#pragma unroll
    for (int hd = 0; hd < 4; ++hd)
        la_store_part(part_s, partial + ((size_t)(b * 4 + hd) * gridDim.x + blockIdx.x) * 1088, m[hd], l[hd], ctxT[hd], tid, wave, l31, half);
}

// pass 1 of the unfused (training) path on the stored qkv tensor: returns the number of parts written per (sample, head)
int launch_la_ctx_stored(const bf16_t* qkv, float* partial, int B, int n, hipStream_t s) {
    const int gx = la_fused_blocks(n, B);
    la_ctx_stored_kernel<<<dim3(gx, B), 256, 0, s>>>(qkv, partial, n);
    return gx;
}

// combine partials -> context as pass 2's A fragments:
// ctxfrag[bh][s2][lane][j] = ctx[d = 16 s2 + 8 (j>>2) + 4 (lane>>5) + (j&3)][e = lane & 31]
// grid (B*4, 4): each workgroup produces 256 of the 1024 fragment elements of one (sample, head);
// the per-part rescale weights exp(m_c - M) are computed once into LDS.
constexpr int LA_MAX_PARTS = 256;
__global__ void __launch_bounds__(256) la_ctx_combine_frag_kernel(const float* __restrict__ partial, bf16_t* __restrict__ ctxfrag,
                                                                  int nparts, float inv_n) {
    __shared__ float red[8][32], M[32], Linv[32], w_s[LA_MAX_PARTS][32];
    const int tid = threadIdx.x, bh = blockIdx.x, dd = tid & 31, grp = tid >> 5;
    const float* base = partial + (size_t)bh * nparts * 1088;
    float mx = -3.0e38f;
    // (four parts' loads in flight per thread in both scans: with few workgroups -- small batches -- these loops are latency chains)
    for (int c0 = grp; c0 < nparts; c0 += 32) {
        float t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = base[(size_t)min(c0 + 8 * k, nparts - 1) * 1088 + dd];     // (clamped: a repeated part does not change a max)
#pragma unroll
        for (int k = 0; k < 4; ++k) mx = fmaxf(mx, t[k]);
    }
    red[grp][dd] = mx;
    __syncthreads();
    if (tid < 32) {
        float m = red[0][tid];
#pragma unroll
        for (int g = 1; g < 8; ++g) m = fmaxf(m, red[g][tid]);
        M[tid] = m;
    }
    __syncthreads();
    const bool use_ws = nparts <= LA_MAX_PARTS;      // more parts (small batches: one part per wave of a grid sized for the chip): weights recomputed below
    float l = 0.0f;
    for (int c0 = grp; c0 < nparts; c0 += 32) {
        float tm[4], tl[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t o = (size_t)min(c0 + 8 * k, nparts - 1) * 1088;
            tm[k] = base[o + dd];
            tl[k] = base[o + 32 + dd];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (c0 + 8 * k < nparts) {
                const float w = __expf(tm[k] - M[dd]);
                if (use_ws) w_s[c0 + 8 * k][dd] = w;
                l += tl[k] * w;
            }
    }
    __syncthreads();
    red[grp][dd] = l;
    __syncthreads();
    if (tid < 32) {
        float t = red[0][tid];
#pragma unroll
        for (int g = 1; g < 8; ++g) t += red[g][tid];
        Linv[tid] = 1.0f / t;
    }
    __syncthreads();
    const int i = blockIdx.y * 256 + tid;
    const int j = i & 7, lane = (i >> 3) & 63, s2 = i >> 9;
    const int d = 16 * s2 + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3), e = lane & 31;
    // eight loads in flight per thread (one at a time this loop was a chain of nparts memory latencies: 54 us per launch); the
    // products are still added in part order
    float a = 0.0f;
    for (int c0 = 0; c0 < nparts; c0 += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = base[(size_t)min(c0 + k, nparts - 1) * 1088 + 64 + d * 32 + e];
        if (use_ws) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (c0 + k < nparts) a += v[k] * w_s[c0 + k][d];
        } else {
            float mc[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) mc[k] = base[(size_t)min(c0 + k, nparts - 1) * 1088 + d];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (c0 + k < nparts) a += v[k] * __expf(mc[k] - M[d]);
        }
    }
    ctxfrag[(size_t)bh * 1024 + i] = f2bf(a * Linv[d] * inv_n);
}

// ---- pass 2 ---------------------------------------------------------------------------------------
// TRAIN (the training forward, C = 64): operands as the unfused path has them (xn with the gain rounded to bf16 times the plain Wq, q rounded
// to bf16 before its softmax -- what the backward recomputes), and o2 = to_out.0's output is WRITTEN (bf16; the to_out.1 LayerNorm backward
// reads it) and the LayerNorm runs on those rounded values: lc_out_kernel + the LayerNorm kernel (5 tensor passes) become 3.
template <int C, bool TRAIN>
__global__ void __launch_bounds__(256, C == 64 ? 4 : 2) la_out_fused_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wq,
                                                           const bf16_t* __restrict__ woutp, const bf16_t* __restrict__ ctxfrag,
                                                           const float* __restrict__ bias, const float* __restrict__ g2,
                                                           bf16_t* __restrict__ y, int n, float eps_pre, float eps_post, float scale,
                                                           const float* __restrict__ g, bf16_t* __restrict__ o2_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KS = C / 16, RT = C / 32;
    constexpr int WQ_B = (C / 8) * 128 * 16, WO_B = 16 * C * 16, CF_B = 4 * 2 * 64 * 16;
    unsigned char* s_wq = smem;
    unsigned char* s_wo = smem + WQ_B;
    unsigned char* s_cf = s_wo + WO_B;
    float* s_bias = (float*)(s_cf + CF_B);
    float* s_g2 = s_bias + C;
    float* s_g = s_g2 + C;                 // TRAIN: the PreNorm gain
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.y;
    if constexpr (TRAIN)
        for (int i = tid; i < C; i += 256) s_g[i] = g[i];
    for (int i = tid; i < WQ_B / 16; i += 256) ((uint4*)s_wq)[i] = ((const uint4*)wq)[i];
    for (int i = tid; i < WO_B / 16; i += 256) ((uint4*)s_wo)[i] = ((const uint4*)woutp)[i];
    for (int i = tid; i < CF_B / 16; i += 256) ((uint4*)s_cf)[i] = ((const uint4*)(ctxfrag + (size_t)b * 4096))[i];
    for (int i = tid; i < C; i += 256) { s_bias[i] = bias[i]; s_g2[i] = g2[i]; }
    __syncthreads();
    const int wave_id = blockIdx.x * 4 + wave, nw = gridDim.x * 4, ntiles = (n + 31) / 32;

    const bf16_t* xb = x + (size_t)b * n * C;
    XRaw<C> raw_next;
    load_raw_x<C>(xb + (size_t)min(wave_id * 32 + l31, n - 1) * C, raw_next, half);
    for (int tile = wave_id; tile < ntiles; tile += nw) {
        asm volatile("" ::: "memory");   // keep the LDS weight reads inside the loop (see pass 1)
        const int pix = tile * 32 + l31;
        const int pid = min(pix, n - 1);
        const bf16_t* xrow = xb + (size_t)pid * C;
        const XRaw<C> raw = raw_next;
        load_raw_x<C>(xb + (size_t)min((tile + nw) * 32 + l31, n - 1) * C, raw_next, half);
        bf16x8 xs[KS];
        if constexpr (TRAIN) norm_x_gain<C>(raw, eps_pre, s_g, half, xs);
        else norm_x<C>(raw, eps_pre, xs);
        f32x16 acc_o[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_o[rt][r] = 0.0f;
#pragma unroll
        for (int hd = 0; hd < 4; ++hd) {
            f32x16 qa;
#pragma unroll
            for (int r = 0; r < 16; ++r) qa[r] = 0.0f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {   // rows = d, col = pixel
                const bf16x8 wf = *(const bf16x8*)(s_wq + ((size_t)(2 * s + half) * 128 + hd * 32 + l31) * 16);
                qa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xs[s], qa, 0, 0, 0);
            }
            if constexpr (TRAIN) {
#pragma unroll
                for (int r = 0; r < 16; ++r) qa[r] = bf2f(f2bf(qa[r]));
            }
            float mx = qa[0];                 // softmax over d (DD:234) then * scale (DD:237)
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, qa[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { qa[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(qa[r], LOG2E, -mx * LOG2E)); sum += qa[r]; }
            sum += __shfl_xor(sum, 32, 64);
            const float k = scale / sum;
#pragma unroll
            for (int r = 0; r < 16; ++r) qa[r] *= k;
            f32x16 oa;                        // out[e][pixel] = sum_d ctx[d][e] q~[d][pixel]   (DD:242)
#pragma unroll
            for (int r = 0; r < 16; ++r) oa[r] = 0.0f;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 cf = *(const bf16x8*)(s_cf + ((size_t)(hd * 2 + s2) * 64 + lane) * 16);
                oa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cf, acc_frag(qa, s2), oa, 0, 0, 0);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {  // o[c][pixel] += sum_e Wout[c][hd*32+e] out[e][pixel]  (to_out.0)
                const bf16x8 of = acc_frag(oa, s2);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const bf16x8 wf = *(const bf16x8*)(s_wo + ((size_t)((hd * 2 + s2) * 2 + half) * C + rt * 32 + l31) * 16);
                    acc_o[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, of, acc_o[rt], 0, 0, 0);
                }
            }
        }
        // bias, LayerNorm over channels (to_out.1, DD:226), + x (Residual)
        float sum = 0.0f;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *(const float4*)(s_bias + rt * 32 + 8 * g + 4 * half);
                acc_o[rt][4 * g] += bv.x; acc_o[rt][4 * g + 1] += bv.y; acc_o[rt][4 * g + 2] += bv.z; acc_o[rt][4 * g + 3] += bv.w;
                if constexpr (TRAIN) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc_o[rt][4 * g + k] = bf2f(f2bf(acc_o[rt][4 * g + k]));       // o2 as stored
                }
                sum += (acc_o[rt][4 * g] + acc_o[rt][4 * g + 1]) + (acc_o[rt][4 * g + 2] + acc_o[rt][4 * g + 3]);
            }
        if constexpr (TRAIN) {
            bf16_t* orow = o2_out + ((size_t)b * n + min(pix, n - 1)) * C;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) store_acc_rows(acc_o[rt], orow + rt * 32, half, pix < n);
        }
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / C);
        float q = 0.0f;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc_o[rt][r] -= mean; q += acc_o[rt][r] * acc_o[rt][r]; }
        q += __shfl_xor(q, 32, 64);
        const float rstd = rsqrtf(q * (1.0f / C) + eps_post);
        {
            // 16-byte accesses: pairs of register quads are exchanged between the half-waves (v_permlane32_swap) so that a
            // lane loads / stores 8 consecutive channels; every lane takes part, out-of-range pixels use a clamped row
            const bool ok = pix < n;
            bf16_t* yrow = y + ((size_t)b * n + min(pix, n - 1)) * C;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    // the residual x: the very words this lane loaded for the LayerNorm (channels 16 s + 8 half .. + 7 with s = 2 rt + g / 2) --
                    // kept in registers (r03; was a second global load per tile: 922 MB more through L2 at full resolution)
                    const u32x4 xw = raw.v[rt * 2 + (g >> 1)];
                    const uint4 xv = make_uint4(xw[0], xw[1], xw[2], xw[3]);
                    const auto sx = __builtin_amdgcn_permlane32_swap(xv.x, xv.z, false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(xv.y, xv.w, false, false);
                    const unsigned xq[2][2] = {{sx[0], sy[0]}, {sx[1], sy[1]}};      // residual of quad g, quad g+1
                    uint2 qo[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int c0 = rt * 32 + 8 * (g + k) + 4 * half;
                        const float4 gv = *(const float4*)(s_g2 + c0);
                        const float o0 = acc_o[rt][4 * (g + k)] * rstd * gv.x + bf2f((bf16_t)(xq[k][0] & 0xffffu));
                        const float o1 = acc_o[rt][4 * (g + k) + 1] * rstd * gv.y + bf2f((bf16_t)(xq[k][0] >> 16));
                        const float o2 = acc_o[rt][4 * (g + k) + 2] * rstd * gv.z + bf2f((bf16_t)(xq[k][1] & 0xffffu));
                        const float o3 = acc_o[rt][4 * (g + k) + 3] * rstd * gv.w + bf2f((bf16_t)(xq[k][1] >> 16));
                        qo[k] = make_uint2(la_pack2(o0, o1), la_pack2(o2, o3));
                    }
                    const auto rx = __builtin_amdgcn_permlane32_swap(qo[0].x, qo[1].x, false, false);
                    const auto ry = __builtin_amdgcn_permlane32_swap(qo[0].y, qo[1].y, false, false);
                    if (ok) *(uint4*)(yrow + rt * 32 + 8 * g + 8 * half) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
                }
        }
    }
}

// ---- weight preparation -----------------------------------------------------------------------------
// wqkv fp32 [384][C] (q | k | v rows), g fp32 [C], wout fp32 [C][128]  ->
//   wq   bf16 [C/8][128][8] = Wq  . diag(g)
//   wkv  bf16 [C/8][256][8] = Wkv . diag(g)
//   wout bf16 [hd*2+s2][half][C][8]: element j = Wout[c][hd*32 + 16 s2 + 8 (j>>2) + 4 half + (j&3)]
__global__ void __launch_bounds__(256) la_weight_prep_kernel(const float* __restrict__ wqkv, const float* __restrict__ g, const float* __restrict__ wout,
                                                             bf16_t* __restrict__ wq, bf16_t* __restrict__ wkv, bf16_t* __restrict__ woutp, int C) {
    const int total = 512 * C;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        if (i < 128 * C) {
            const int j = i & 7, o = (i >> 3) & 127, c8 = i >> 10, c = c8 * 8 + j;
            wq[i] = f2bf(wqkv[(size_t)o * C + c] * (g ? g[c] : 1.0f));
        } else if (i < 384 * C) {
            const int k = i - 128 * C;
            const int j = k & 7, o = (k >> 3) & 255, c8 = k >> 11, c = c8 * 8 + j;
            wkv[k] = f2bf(wqkv[(size_t)(128 + o) * C + c] * (g ? g[c] : 1.0f));
        } else if (woutp) {
            const int k = i - 384 * C;
            const int j = k & 7, c = (k >> 3) % C, rest = (k >> 3) / C;   // rest = (hd*2+s2)*2 + half
            const int hh = rest & 1, s2 = (rest >> 1) & 1, hd = rest >> 2;
            woutp[k] = f2bf(wout[(size_t)c * 128 + hd * 32 + 16 * s2 + 8 * (j >> 2) + 4 * hh + (j & 3)]);
        }
    }
}

static float la_defer() {
    static const float defer = getenv("OFD_LA_DEFER") ? (float)atof(getenv("OFD_LA_DEFER")) : 5.545177f;      // 8 ln 2; 0: the reference point follows every new maximum
    return defer;
}

template <int C>
static int launch_la(const bf16_t* x, const bf16_t* wq, const bf16_t* wkv, const bf16_t* woutp, const float* bias, const float* g2,
                     float* partial, bf16_t* ctxfrag, bf16_t* y, int B, int n, float eps_pre, float eps_post, hipStream_t s) {
    constexpr int LDS1 = (C / 8) * 256 * 16;
    constexpr int LDS2 = (C / 8) * 128 * 16 + 16 * C * 16 + 4 * 2 * 64 * 16 + 2 * C * 4;
    static bool attr = false;
    if (!attr) {
        OFD_HIP(hipFuncSetAttribute((const void*)la_ctx_fused_kernel<C, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS1));
        OFD_HIP(hipFuncSetAttribute((const void*)la_out_fused_kernel<C, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2));
        attr = true;
    }
    const int gx = la_fused_blocks(n, B);
    const float defer = la_defer();
    la_ctx_fused_kernel<C, false><<<dim3(gx, B), 256, LDS1, s>>>(x, wkv, partial, n, eps_pre, defer, nullptr, nullptr, nullptr);
    la_ctx_combine_frag_kernel<<<dim3(B * 4, 4), 256, 0, s>>>(partial, ctxfrag, gx, 1.0f / (float)n);
    int gx2 = cdiv(cdiv(n, 32), 4 * 4);     // >= 4 tiles per wave amortise the weight staging
    if (gx2 < 1) gx2 = 1;
    static int gx2_cap = -1;
    if (gx2_cap < 0) { const char* e = getenv("OFD_LA_GX2"); gx2_cap = e ? atoi(e) : 128; }
    const int cap2 = (gx2_cap * B < 1024) ? 1024 / B : gx2_cap;          // small batches: enough workgroups for the chip
    if (gx2 > cap2) gx2 = cap2;
    la_out_fused_kernel<C, false><<<dim3(gx2, B), 256, LDS2, s>>>(x, wq, woutp, ctxfrag, bias, g2, y, n, eps_pre, eps_post, 0.17677669529663687f, nullptr, nullptr);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

void launch_la_ctx_combine(const float* partial, float* ctx, int B, int nparts, float inv_n, float* ml_out, hipStream_t s);   // blocks.hip

// The training forward of a 64-channel block: the two fused passes, which also leave the tape (xn, k | v inside qkv, o2) and the fp32 context + softmax
// statistics (ctx, ml) of the unfused path for the backward (unet_train.hip linattn_backward).  wq / wkv: the PLAIN to_qkv weights in fragment layout.
int k_linear_attention_fused_train(const bf16_t* x, const bf16_t* wq, const bf16_t* wkv, const bf16_t* woutp, const float* bias, const float* g_pre,
                                   const float* g2, float* partial, bf16_t* ctxfrag, float* ctx, float* ml, bf16_t* xn, bf16_t* qkv, bf16_t* o2,
                                   bf16_t* y, int B, int n, int C, float eps_pre, float eps_post, hipStream_t s) {
    if (C != 64) { set_error("linear_attention_fused_train: C=%d unsupported", C); return OFD_ERR_ARG; }
    constexpr int LDS1 = (64 / 8) * 256 * 16 + 64 * 4;
    constexpr int LDS2 = (64 / 8) * 128 * 16 + 16 * 64 * 16 + 4 * 2 * 64 * 16 + 3 * 64 * 4;
    static bool attr = false;
    if (!attr) {
        OFD_HIP(hipFuncSetAttribute((const void*)la_ctx_fused_kernel<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS1));
        OFD_HIP(hipFuncSetAttribute((const void*)la_out_fused_kernel<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2));
        attr = true;
    }
    const int gx = la_fused_blocks(n, B);
    la_ctx_fused_kernel<64, true><<<dim3(gx, B), 256, LDS1, s>>>(x, wkv, partial, n, eps_pre, la_defer(), g_pre, xn, qkv);
    la_ctx_combine_frag_kernel<<<dim3(B * 4, 4), 256, 0, s>>>(partial, ctxfrag, gx, 1.0f / (float)n);
    launch_la_ctx_combine(partial, ctx, B, gx, 1.0f / (float)n, ml, s);
    int gx2 = cdiv(cdiv(n, 32), 4 * 4);
    if (gx2 < 1) gx2 = 1;
    const int cap2 = (128 * B < 1024) ? 1024 / B : 128;
    if (gx2 > cap2) gx2 = cap2;
    la_out_fused_kernel<64, true><<<dim3(gx2, B), 256, LDS2, s>>>(x, wq, woutp, ctxfrag, bias, g2, y, n, eps_pre, eps_post, 0.17677669529663687f, g_pre, o2);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

// workgroups per sample of the first pass (= parts per (sample, head): each workgroup writes one): 64 at the batch sizes that fill the chip by
// themselves, up to OFD_LA_GX1_TOTAL (default 256: one per CU; 128 / 192 / 384 / 512 measured slower at B = 1 1080p --
// beyond 256 the combine's chain over the parts costs more than the first pass gains) over the batch for small ones -- at B = 1 (1080p: BASELINE configs[4] per
// GPU) 64 workgroups were a quarter of the CUs
int la_fused_blocks(int n, int B) {
    static const int total = getenv("OFD_LA_GX1_TOTAL") ? atoi(getenv("OFD_LA_GX1_TOTAL")) : 256;
    int gx = cdiv(cdiv(n, 32), 8);
    int cap = total / (B < 1 ? 1 : B);
    if (cap < 64) cap = 64;
    if (gx < 1) gx = 1;
    if (gx > cap) gx = cap;
    return gx;
}

int k_la_weight_prep(const float* wqkv, const float* g, const float* wout, bf16_t* wq, bf16_t* wkv, bf16_t* woutp, int C, hipStream_t s) {
    la_weight_prep_kernel<<<cdiv((woutp ? 512 : 384) * C, 256), 256, 0, s>>>(wqkv, g, wout, wq, wkv, woutp, C);      // g == nullptr: plain weights; woutp == nullptr: q | k | v only
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

int k_linear_attention_fused(const bf16_t* x, const bf16_t* wq, const bf16_t* wkv, const bf16_t* woutp, const float* bias, const float* g2,
                             float* partial, bf16_t* ctxfrag, bf16_t* y, int B, int n, int C, float eps_pre, float eps_post, hipStream_t s) {
    if (C == 64) return launch_la<64>(x, wq, wkv, woutp, bias, g2, partial, ctxfrag, y, B, n, eps_pre, eps_post, s);
    if (C == 128) return launch_la<128>(x, wq, wkv, woutp, bias, g2, partial, ctxfrag, y, B, n, eps_pre, eps_post, s);
    set_error("linear_attention_fused: C=%d unsupported", C);
    return OFD_ERR_ARG;
}

}  // namespace ofd
