// LinearAttention core on a materialised qkv tensor (training path; inference uses la_fused.hip),
// DD:229-242: q softmax over d (* scale), k softmax over the pixels, ctx = k . v^T / n, out = ctx^T q.
// Forward and backward, MFMA 32x32x16 bf16, two passes over the pixels each:
//   pass 1 (reduction over pixels, K = pixel): context-shaped 32x32 sums per (sample, head); both operands
//           are pixel-major in LDS and read with the transposing LDS read;
//   pass 2 (per pixel): the PIXEL sits on the MFMA column (= lane), so every per-pixel softmax reduction is
//           in-lane plus one cross-half shuffle, and the 32x32 context matrices are the A operands.
// qkv / dqkv: [B][n][384] bf16 (q | k | v, 4 heads x 32); out / dout: [B][n][128].
#include <cstdlib>
#include "blocks.h"
#include "det.h"
#include "mfma_util.h"

namespace ofd {

constexpr int LC_CH = 128;            // pixels per staged chunk
constexpr float LC_SCALE = 0.17677669529663687f;

__device__ __forceinline__ void lc_unpack8(const u32x4& v, float* f) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[2 * j] = bf2f((bf16_t)(v[j] & 0xffffu));
        f[2 * j + 1] = bf2f((bf16_t)(v[j] >> 16));
    }
}
__device__ __forceinline__ u32x4 lc_pack8(const float* f) {
    u32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = f2bf2(f[2 * j], f[2 * j + 1]);
    return v;
}
__device__ __forceinline__ bf16x8 lc_frag(const float* f) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)f[j];
    return r;
}

constexpr int FC_CH = 96;             // forward pass 1 chunk (two 24 KB tiles + scan scratch within the 64 KB static LDS)
// 32 channels of one head for this lane's pixel: accumulator register 4g+j holds channel 8g + 4*half + j.  One
// v_permlane32_swap per dword pairs the half-waves so that every lane stores 16 bytes (8 consecutive channels):
// half as many store instructions, 32 contiguous bytes per pixel and instruction.  All lanes must call it.
__device__ __forceinline__ void lc_store_head(bf16_t* dst, const uint2 (&q)[4], int half, bool ok) {
#pragma unroll
    for (int g = 0; g < 4; g += 2) {
        const auto rx = __builtin_amdgcn_permlane32_swap(q[g].x, q[g + 1].x, false, false);
        const auto ry = __builtin_amdgcn_permlane32_swap(q[g].y, q[g + 1].y, false, false);
        if (ok) *(uint4*)(dst + 8 * g + 8 * half) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
    }
}

// inverse of lc_store_head: two 16-byte loads + permlane swaps give this lane its 16 accumulator-layout channels
// (register 4g+j <-> channel 8g + 4*half + j) of a 32-channel head.  All lanes must call it (clamped address).
__device__ __forceinline__ void lc_load_head(const bf16_t* src, int half, float (&f)[16]) {
#pragma unroll
    for (int g = 0; g < 4; g += 2) {
        const uint4 v = *(const uint4*)(src + 8 * g + 8 * half);
        const auto rx = __builtin_amdgcn_permlane32_swap(v.x, v.z, false, false);
        const auto ry = __builtin_amdgcn_permlane32_swap(v.y, v.w, false, false);
        const unsigned w4[4] = {rx[0], ry[0], rx[1], ry[1]};     // quad g: (x, y), quad g+1: (x, y)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[4 * g + 2 * j] = bf2f((bf16_t)(w4[j] & 0xffffu));
            f[4 * g + 2 * j + 1] = bf2f((bf16_t)(w4[j] >> 16));
        }
    }
}

// ---- forward pass 1: partial {m[32], l[32], ctx[32][32]} per (sample*head, part); grid (nparts, B) ----
// workgroup = 4 waves = 4 heads; online max over chunks of 96 pixels (accumulators rescaled per row d)
__global__ void __launch_bounds__(256) lc_ctx_partial_kernel(const bf16_t* __restrict__ qkv, float* __restrict__ partial, int n, int span, int nparts) {
    __shared__ __attribute__((aligned(16))) unsigned char ks[FC_CH * 256];     // [pixel][128 k channels] bf16 (raw, then exp(k - m))
    __shared__ __attribute__((aligned(16))) unsigned char vs[FC_CH * 256];     // [pixel][128 v channels]
    __shared__ float mx2[2][128], m_s[128], f_s[128], l2[2][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int part = blockIdx.x, b = blockIdx.y;
    const int n_begin = part * span, n_end = min(n, n_begin + span);
    const int ch = tid & 127, ph = tid >> 7;           // channel scans: thread -> (channel, pixel parity half)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    float l_run = 0.0f;                                // threads < 128: running sum of channel tid
    if (tid < 128) m_s[tid] = -3.0e38f;
    for (int c0 = n_begin; c0 < n_end; c0 += FC_CH) {
        const int cnt = min(FC_CH, n_end - c0);
        __syncthreads();
        // stage k and v: 128 px x (16 + 16) 16-byte units; pixels past the end: k = -inf-like, v = 0
#pragma unroll
        for (int i = 0; i < FC_CH / 8; ++i) {
            const int id = tid + i * 256, p = id >> 5, u = id & 31;
            const bool ok = p < cnt;
            const bf16_t* row = qkv + ((size_t)b * n + c0 + min(p, cnt - 1)) * 384 + 128;
            u32x4 v = *(const u32x4*)(row + u * 8);
            if (!ok) {
                const unsigned fill = (u < 16) ? 0xff7fff7fu : 0u;      // bf16 -3.4e38 | 0
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fill;
            }
            if (u < 16) *(u32x4*)(ks + p * 256 + u * 16) = v;
            else *(u32x4*)(vs + p * 256 + (u - 16) * 16) = v;
        }
        __syncthreads();
        {   // chunk max per channel
            float mx = -3.0e38f;
            for (int p = ph; p < FC_CH; p += 2) mx = fmaxf(mx, bf2f(*(const bf16_t*)(ks + p * 256 + ch * 2)));
            mx2[ph][ch] = mx;
        }
        __syncthreads();
        if (tid < 128) {
            const float m_old = m_s[tid], m_new = fmaxf(m_old, fmaxf(mx2[0][tid], mx2[1][tid]));
            f_s[tid] = __expf(m_old - m_new);
            m_s[tid] = m_new;
        }
        __syncthreads();
        {   // exponentiate in place (bf16), row sums of the values as the MFMA will see them
            const float m = m_s[ch];
            float sum = 0.0f;
            for (int p = ph; p < FC_CH; p += 2) {
                bf16_t* a = (bf16_t*)(ks + p * 256 + ch * 2);
                const bf16_t e = f2bf(__expf(bf2f(*a) - m));
                *a = e;
                sum += bf2f(e);
            }
            l2[ph][ch] = sum;
        }
        __syncthreads();
        if (tid < 128) l_run = l_run * f_s[tid] + l2[0][tid] + l2[1][tid];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] *= f_s[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
#pragma unroll
        for (int sl = 0; sl < FC_CH / 16; ++sl) {
            const bf16x8 kf = tr_frag(ks + sl * 16 * 256 + wave * 64, 256, lane);      // rows = d
            const bf16x8 vf = tr_frag(vs + sl * 16 * 256 + wave * 64, 256, lane);      // cols = e
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, vf, acc, 0, 0, 0);
        }
    }
    __syncthreads();
    float* o = partial + ((size_t)(b * 4 + wave) * nparts + part) * 1088;
    if (tid < 128) {
        float* oh = partial + ((size_t)(b * 4 + (tid >> 5)) * nparts + part) * 1088;
        oh[tid & 31] = m_s[tid];
        oh[32 + (tid & 31)] = l_run;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) o[64 + ((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[r];
}

// context-shaped matrices of one sample as bf16 A operands in LDS: [4 heads][32 rows][40] (80-byte rows)
__device__ __forceinline__ void lc_stage_matrix(unsigned char* lds, const float* __restrict__ src, bool transpose, int tid) {
    for (int i = tid; i < 4096; i += 256) {
        const int h = i >> 10, r = (i >> 5) & 31, c = i & 31;
        const float v = src[i];
        const int rr = transpose ? c : r, cc = transpose ? r : c;
        *(bf16_t*)(lds + ((h * 32 + rr) * 40 + cc) * 2) = f2bf(v);
    }
}
__device__ __forceinline__ bf16x8 lc_afrag(const unsigned char* lds, int h, int s, int l31, int half) {
    return *(const bf16x8*)(lds + ((h * 32 + l31) * 40 + s * 16 + half * 8) * 2);
}

// ---- forward pass 2: out[n][e] = sum_d ctx[d][e] * softmax_d(q)[n][d] * scale; grid (gx, B) -----------------
// All global traffic of the per-pixel passes goes through an LDS tile of 32 pixels: whole pixel rows are fetched with the
// direct global->LDS path (one 1 KB / 256 B instruction per pixel row, no staging registers) and results leave as
// consecutive 16-byte units of consecutive pixels.  Reading operand fragments straight from global memory (32 bytes per
// pixel and instruction) ran at ~2.8 TB/s; a padded row pitch keeps the lane-per-pixel LDS reads conflict-free.
constexpr int LO_PITCH = 256 + 128 + 16;  // q part of a pixel row (128 channels) | xn row (64 channels, q recomputed from it) | pad
// wo != nullptr (C = 64 or 128): the to_out.0 1x1 conv (DD:225) rides on the head-output tile while it is in LDS -- o2 = Wo ao + bo, wave w
// owns the 32-channel block w of o2 and keeps its eight A fragments (128 input channels) in registers for the whole launch.  The
// separate conv launch and its read of ao (1.85 GB at full resolution) go away; ao itself is still written (the backward's to_out
// weight gradient reads it).
// wq != nullptr (C = 64): q is RECOMPUTED from xn (q = Wq xn, four MFMAs per head and tile with the head's A fragments in registers for
// the launch; rounded to bf16 like the stored tensor) -- the to_qkv conv need not write q and this pass reads 64 channels instead of 128.
__global__ void __launch_bounds__(256) lc_out_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ ctx, bf16_t* __restrict__ out, int n,
                                                     const bf16_t* __restrict__ wo, const float* __restrict__ bo, bf16_t* __restrict__ o2, int C,
                                                     const bf16_t* __restrict__ xn, const bf16_t* __restrict__ wq) {
    __shared__ __attribute__((aligned(16))) unsigned char ct[4 * 32 * 80];      // ctx^T: rows e, k = d
    __shared__ __attribute__((aligned(16))) unsigned char st[32 * LO_PITCH];    // q rows in, out rows out
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5, b = blockIdx.y;
    lc_stage_matrix(ct, ctx + (size_t)b * 4096, true, tid);
    const bool proj = wo != nullptr && wave * 32 < C;       // this wave computes a block of o2
    bf16x8 wof[8];
    float4 bo4[4];
    if (proj) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) wof[ks] = *(const bf16x8*)(wo + ((size_t)(ks * 2 + half) * C + wave * 32 + l31) * 8);
#pragma unroll
        for (int g = 0; g < 4; ++g) bo4[g] = bo ? *(const float4*)(bo + wave * 32 + 8 * g + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int h = wave;                    // wave = head; the four waves of a workgroup share a 32-pixel tile
    bf16x8 wqf[4];                         // recompute: A fragments of Wq for this head (rows d, k = the 64 channels of xn)
    if (wq) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wqf[ks] = *(const bf16x8*)(wq + ((size_t)(ks * 2 + half) * 384 + h * 32 + l31) * 8);
    }
    for (int p0 = blockIdx.x * 32; p0 < n; p0 += gridDim.x * 32) {
        __syncthreads();                   // previous tile's copy-out is done (and, first time, ct is staged)
        if (wq) {   // 32 pixels x 8 units of xn: one per thread
            const int px = tid >> 3, u = tid & 7;
            *(u32x4*)(st + px * LO_PITCH + 256 + u * 16) = *(const u32x4*)(xn + ((size_t)b * n + min(p0 + px, n - 1)) * 64 + u * 8);
        } else {    // 32 pixels x 16 units of q: 512 units, two per thread, consecutive lanes = consecutive units of a pixel row
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int id = tid + k * 256, px = id >> 4, u = id & 15;
                const int p = min(p0 + px, n - 1);
                *(u32x4*)(st + px * LO_PITCH + u * 16) = *(const u32x4*)(qkv + ((size_t)b * n + p) * 384 + u * 8);
            }
        }
        __syncthreads();
        const unsigned char* row = st + l31 * LO_PITCH;
        float q[16];
        if (wq) {                          // operand-layout q (channel 16 s + 8 half + j of the head) from accumulator-layout MFMA output
            f32x16 qa;
#pragma unroll
            for (int r = 0; r < 16; ++r) qa[r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wqf[ks], *(const bf16x8*)(row + 256 + (ks * 16 + half * 8) * 2), qa, 0, 0, 0);
            // the softmax below is layout-agnostic (all 32 d of a pixel sit in its two half-lanes either way); the MFMA operand built from
            // q afterwards (lc_frag) wants channel 8 half + j (+16): go through the tile like the stored tensor would
            uint2 qq[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) qq[g] = make_uint2(f2bf2(qa[4 * g], qa[4 * g + 1]), f2bf2(qa[4 * g + 2], qa[4 * g + 3]));
            lc_store_head((bf16_t*)(st + l31 * LO_PITCH) + h * 32, qq, half, true);
        }
        lc_unpack8(*(const u32x4*)(row + (h * 32 + half * 8) * 2), &q[0]);
        lc_unpack8(*(const u32x4*)(row + (h * 32 + 16 + half * 8) * 2), &q[8]);
        float mx = q[0];
#pragma unroll
        for (int j = 1; j < 16; ++j) mx = fmaxf(mx, q[j]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j) { q[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(q[j], 1.4426950408889634f, -mx * 1.4426950408889634f)); sum += q[j]; }
        sum += __shfl_xor(sum, 32, 64);
        const float k = LC_SCALE * __builtin_amdgcn_rcpf(sum);
#pragma unroll
        for (int j = 0; j < 16; ++j) q[j] *= k;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lc_afrag(ct, h, 0, l31, half), lc_frag(&q[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lc_afrag(ct, h, 1, l31, half), lc_frag(&q[8]), acc, 0, 0, 0);
        {   // results over this wave's own q slice of the tile (nobody else reads or writes those 64 bytes per pixel)
            uint2 qo[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) qo[g] = make_uint2(f2bf2(acc[4 * g], acc[4 * g + 1]), f2bf2(acc[4 * g + 2], acc[4 * g + 3]));
            lc_store_head((bf16_t*)(st + l31 * LO_PITCH) + h * 32, qo, half, true);
        }
        __syncthreads();
        if (out) {                         // (nullptr: training with the fused backward -- nothing reads the head outputs again)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int id = tid + k * 256, px = id >> 4, u = id & 15;
                if (p0 + px < n) *(u32x4*)(out + ((size_t)b * n + p0 + px) * 128 + u * 8) = *(const u32x4*)(st + px * LO_PITCH + u * 16);
            }
        }
        if (proj) {                        // o2 block of this wave: rows = channel, cols = pixel, k = the 128 head-output channels
            f32x16 pa;
#pragma unroll
            for (int r = 0; r < 16; ++r) pa[r] = 0.0f;
            const unsigned char* arow = st + l31 * LO_PITCH + half * 16;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) pa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wof[ks], *(const bf16x8*)(arow + ks * 32), pa, 0, 0, 0);
            uint2 po[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                po[g] = make_uint2(f2bf2(pa[4 * g] + bo4[g].x, pa[4 * g + 1] + bo4[g].y), f2bf2(pa[4 * g + 2] + bo4[g].z, pa[4 * g + 3] + bo4[g].w));
            lc_store_head(o2 + ((size_t)b * n + min(p0 + l31, n - 1)) * C + wave * 32, po, half, p0 + l31 < n);
        }
    }
}

// ---- backward pass 1: dctx[d][e] = sum_n qs[n][d] * dout[n][e]; partial per (sample*head, part); grid (nparts, B)
// FD (C = 64): `dout` is do2 [pixel][64], the gradient of the to_out.0 OUTPUT -- dout = Wo^T do2 is never formed: since
// dctx = sum_n qs (x) (Wo^T do2) = (sum_n qs (x) do2) Wo, the pixel reduction runs on do2 (partial [32 d][64 c] per head and part) and the
// 64 x 32 product with Wo is left to the combine kernel.
template <bool FD>
__global__ void __launch_bounds__(256) lc_dctx_partial_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout, float* __restrict__ partial,
                                                              int n, int span, int nparts, float* __restrict__ dbias) {
    __shared__ __attribute__((aligned(16))) unsigned char qs[LC_CH * 256];     // softmax_d(q) * scale, bf16, pixel-major
    __shared__ __attribute__((aligned(16))) unsigned char gs[LC_CH * (FD ? 128 : 256)];     // dout | do2
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int part = blockIdx.x, b = blockIdx.y;
    const int n_begin = part * span, n_end = min(n, n_begin + span);
    f32x16 acc, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.0f; acc1[r] = 0.0f; }
    float bsum = 0.0f;                     // FD: column sums of do2 = the to_out.0 bias gradient (thread -> channel tid & 63, pixels (tid >> 6) * 32 .. + 31)
    for (int c0 = n_begin; c0 < n_end; c0 += LC_CH) {
        const int cnt = min(LC_CH, n_end - c0);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // item -> (pixel, head, half of d): 16 q values, softmax completed with the neighbouring lane
            const int id = tid + i * 256, p = id >> 3, hh = (id >> 1) & 3, hf = id & 1;
            const bool ok = p < cnt;
            const size_t pix = (size_t)b * n + c0 + min(p, cnt - 1);
            float q[16];
            lc_unpack8(*(const u32x4*)(qkv + pix * 384 + hh * 32 + hf * 16), &q[0]);
            lc_unpack8(*(const u32x4*)(qkv + pix * 384 + hh * 32 + hf * 16 + 8), &q[8]);
            u32x4 g0, g1;
            if constexpr (FD) {
                g0 = *(const u32x4*)(dout + pix * 64 + (hh * 2 + hf) * 8);       // 8 items per pixel: one 16-byte unit of the do2 row each
                g1 = g0;
            } else {
                g0 = *(const u32x4*)(dout + pix * 128 + hh * 32 + hf * 16);
                g1 = *(const u32x4*)(dout + pix * 128 + hh * 32 + hf * 16 + 8);
            }
            float mx = q[0];
#pragma unroll
            for (int j = 1; j < 16; ++j) mx = fmaxf(mx, q[j]);
            mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
            float sum = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) { q[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(q[j], 1.4426950408889634f, -mx * 1.4426950408889634f)); sum += q[j]; }
            sum += __shfl_xor(sum, 1, 64);
            const float k = ok ? LC_SCALE * __builtin_amdgcn_rcpf(sum) : 0.0f;      // pixels past the end contribute nothing
#pragma unroll
            for (int j = 0; j < 16; ++j) q[j] *= k;
            if (!ok) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { g0[j] = 0u; g1[j] = 0u; }
            }
            *(u32x4*)(qs + p * 256 + hh * 64 + hf * 32) = lc_pack8(&q[0]);
            *(u32x4*)(qs + p * 256 + hh * 64 + hf * 32 + 16) = lc_pack8(&q[8]);
            if constexpr (FD) {
                *(u32x4*)(gs + p * 128 + (hh * 2 + hf) * 16) = g0;
            } else {
                *(u32x4*)(gs + p * 256 + hh * 64 + hf * 32) = g0;
                *(u32x4*)(gs + p * 256 + hh * 64 + hf * 32 + 16) = g1;
            }
        }
        __syncthreads();
        if constexpr (FD) {
            if (dbias) {
#pragma unroll 8
                for (int p = (tid >> 6) * 32; p < (tid >> 6) * 32 + 32; ++p) bsum += bf2f(*(const bf16_t*)(gs + p * 128 + (tid & 63) * 2));
            }
        }
#pragma unroll
        for (int sl = 0; sl < LC_CH / 16; ++sl) {
            const bf16x8 qf = tr_frag(qs + sl * 16 * 256 + wave * 64, 256, lane);      // rows = d
            if constexpr (FD) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, tr_frag(gs + sl * 16 * 128, 128, lane), acc, 0, 0, 0);          // cols = c 0..31
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, tr_frag(gs + sl * 16 * 128 + 64, 128, lane), acc1, 0, 0, 0);   // cols = c 32..63
            } else {
                const bf16x8 gf = tr_frag(gs + sl * 16 * 256 + wave * 64, 256, lane);      // cols = e
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, gf, acc, 0, 0, 0);
            }
        }
    }
    if constexpr (FD) {
        if (dbias) gacc_add(dbias + (tid & 63), bsum);
        float* o = partial + ((size_t)(b * 4 + wave) * nparts + part) * 2048;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o[((r & 3) + 8 * (r >> 2) + 4 * half) * 64 + l31] = acc[r];
            o[((r & 3) + 8 * (r >> 2) + 4 * half) * 64 + 32 + l31] = acc1[r];
        }
    } else {
        float* o = partial + ((size_t)(b * 4 + wave) * nparts + part) * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[r];
    }
}

// ---- backward pass 1 with q RECOMPUTED from xn and dout in its do2 form (C = 64, every training fusion on): the chunk's xn and do2 rows
// are staged (128 pixels x 128 B each), a wave (= head) forms q = Wq xn for the four 32-pixel tiles (A fragments in registers for the
// launch), the softmax over d runs in the accumulator layout, and the scaled result goes to the pixel-major `qs` tile the transposing
// reads want.  Output: the [32 d][64 c] partials of lc_dctx_partial_kernel<true>.
constexpr int RQ_CH = 64;            // pixels per chunk of the recompute pass (32 KB of LDS: four workgroups per CU)
__global__ void __launch_bounds__(256) lc_dctx_partial_rq_kernel(const bf16_t* __restrict__ xn, const bf16_t* __restrict__ wq, const bf16_t* __restrict__ do2,
                                                                 float* __restrict__ partial, int n, int span, int nparts, float* __restrict__ dbias) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dq_smem[];
    unsigned char* qs = dq_smem;                      // [128 pixels][256 B] softmax_d(q) * scale
    unsigned char* gs = qs + RQ_CH * 256;             // [128 pixels][128 B] do2
    unsigned char* xs = gs + RQ_CH * 128;             // [128 pixels][128 B] xn
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5, h = wave;
    const int part = blockIdx.x, b = blockIdx.y;
    const int n_begin = part * span, n_end = min(n, n_begin + span);
    bf16x8 wqf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wqf[ks] = *(const bf16x8*)(wq + ((size_t)(ks * 2 + half) * 384 + h * 32 + l31) * 8);
    f32x16 acc, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.0f; acc1[r] = 0.0f; }
    float bsum = 0.0f;
    for (int c0 = n_begin; c0 < n_end; c0 += RQ_CH) {
        const int cnt = min(RQ_CH, n_end - c0);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RQ_CH / 32; ++i) {        // RQ_CH pixels x 8 units of each tensor
            const int id = tid + i * 256, p = id >> 3, u = id & 7;
            const size_t pix = (size_t)b * n + c0 + min(p, cnt - 1);
            u32x4 g = *(const u32x4*)(do2 + pix * 64 + u * 8);
            if (p >= cnt) g = u32x4{0u, 0u, 0u, 0u};  // pixels past the end contribute nothing
            *(u32x4*)(gs + p * 128 + u * 16) = g;
            *(u32x4*)(xs + p * 128 + u * 16) = *(const u32x4*)(xn + pix * 64 + u * 8);
        }
        __syncthreads();
        if (dbias) {
#pragma unroll 8
            for (int p = (tid >> 6) * (RQ_CH / 4); p < (tid >> 6) * (RQ_CH / 4) + RQ_CH / 4; ++p) bsum += bf2f(*(const bf16_t*)(gs + p * 128 + (tid & 63) * 2));
        }
#pragma unroll
        for (int pt = 0; pt < RQ_CH / 32; ++pt) {
            const int p = pt * 32 + l31;
            f32x16 qa;
#pragma unroll
            for (int r = 0; r < 16; ++r) qa[r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wqf[ks], *(const bf16x8*)(xs + p * 128 + (ks * 16 + half * 8) * 2), qa, 0, 0, 0);
            float q[16];
            float mx = -3.0e38f;
#pragma unroll
            for (int j = 0; j < 16; ++j) { q[j] = bf2f(f2bf(qa[j])); mx = fmaxf(mx, q[j]); }      // (bf16: what the stored q tensor held)
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) { q[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(q[j], 1.4426950408889634f, -mx * 1.4426950408889634f)); sum += q[j]; }
            sum += __shfl_xor(sum, 32, 64);
            const float k = p < cnt ? LC_SCALE * __builtin_amdgcn_rcpf(sum) : 0.0f;
            uint2 qq[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) qq[g] = make_uint2(f2bf2(q[4 * g] * k, q[4 * g + 1] * k), f2bf2(q[4 * g + 2] * k, q[4 * g + 3] * k));
            lc_store_head((bf16_t*)(qs + p * 256) + h * 32, qq, half, true);
        }
        __syncthreads();
#pragma unroll
        for (int sl = 0; sl < RQ_CH / 16; ++sl) {
            const bf16x8 qf = tr_frag(qs + sl * 16 * 256 + wave * 64, 256, lane);      // rows = d
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, tr_frag(gs + sl * 16 * 128, 128, lane), acc, 0, 0, 0);          // cols = c 0..31
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, tr_frag(gs + sl * 16 * 128 + 64, 128, lane), acc1, 0, 0, 0);   // cols = c 32..63
        }
    }
    if (dbias) gacc_add(dbias + (tid & 63), bsum);
    float* o = partial + ((size_t)(b * 4 + wave) * nparts + part) * 2048;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        o[((r & 3) + 8 * (r >> 2) + 4 * half) * 64 + l31] = acc[r];
        o[((r & 3) + 8 * (r >> 2) + 4 * half) * 64 + 32 + l31] = acc1[r];
    }
}

// ---- backward pass 2: per pixel; grid (gx, B) -----------------------------------------------------------
//   dq_raw = sm(q) * (dq - <sm(q), dq>), dq = scale * ctx . dout          (MFMA rows d, A = ctx)
//   dk_raw = k * (dctx . v / n - S),    k = exp(k_raw - M) / L            (MFMA rows d, A = dctx)
//   dv     = dctx^T . k / n                                               (MFMA rows e, A = dctx^T)
constexpr int LB_PITCH = 1024 + 16;       // qkv row (768 B) | dout row (256 B) | pad
constexpr int LB_LDS = 3 * 4 * 32 * 80 + 3 * 128 * 4 + 32 * LB_PITCH;
constexpr int LB_LDS_FUSE = LB_LDS + 32 * 128 + 2 * 16 * 64 * 4;      // + the xn tile [32 pixels][64 channels] + the data-gradient partial tiles of waves 2, 3
// FUSE (C = 64): the backward of the to_qkv 1x1 conv (DD:222) is done here, on the dqkv tile while it is in LDS -- dxn = W^T dqkv
// (waves 0, 1: one 32-channel block each, 24 k-steps over the 384 qkv channels; A fragments = the data-gradient weights [48][64][8]
// from L2) and dW[ci][co] += xn^T dqkv (all waves: 64 ci x 96 co each, contraction over the 32 pixels through transposing LDS reads,
// accumulated in registers over the whole launch, added to `dw` with atomics at the end).  The 5.5 GB dqkv tensor of a full-resolution
// block is then never written, nor read twice by the two 1x1 backward kernels: 25.8 GB -> 9.2 GB of traffic for the three steps.
// RQ (with FUSE): q is recomputed from the xn tile (q = Wq xn, wq: the prepared to_qkv weights [8][384][8]); the q part of the qkv rows is not read
template <bool FUSE, bool FD, bool RQ>      // FD (with FUSE): `dout` is do2 [pixel][64]; dout = Wo^T do2 is formed per tile and head (wot: [8][128][8])
__global__ void __launch_bounds__(256, 2) lc_bwd_apply_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout, const float* __restrict__ ctx,
                                                           const float* __restrict__ dctx, const float* __restrict__ ml, const float* __restrict__ S,
                                                           bf16_t* __restrict__ dqkv, int n, const bf16_t* __restrict__ xn, const bf16_t* __restrict__ wt,
                                                           float* __restrict__ dw, bf16_t* __restrict__ dxn, const bf16_t* __restrict__ wot, const bf16_t* __restrict__ wq) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lb_smem[];
    unsigned char* cA = lb_smem;
    unsigned char* dA = cA + 4 * 32 * 80;
    unsigned char* dT = dA + 4 * 32 * 80;
    float* Ms = (float*)(dT + 4 * 32 * 80);
    float* Li = Ms + 128;
    float* Ss = Li + 128;
    unsigned char* st = (unsigned char*)(Ss + 128);    // [32 pixels][qkv 768 B | dout 256 B | pad]; dqkv overwrites qkv in place
    unsigned char* xt = st + 32 * LB_PITCH;            // FUSE: [32 pixels][128 B] xn tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5, b = blockIdx.y;
    f32x16 wacc[FUSE ? 2 : 1][FUSE ? 3 : 1];           // FUSE: dW tiles (ci block a) x (co block wave * 96 + 32 j)
    if constexpr (FUSE) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) wacc[a][j][r] = 0.0f;
    }
    lc_stage_matrix(cA, ctx + (size_t)b * 4096, false, tid);
    lc_stage_matrix(dA, dctx + (size_t)b * 4096, false, tid);
    lc_stage_matrix(dT, dctx + (size_t)b * 4096, true, tid);
    if (tid < 128) {
        Ms[tid] = ml[((size_t)b * 4 + (tid >> 5)) * 64 + (tid & 31)];
        Li[tid] = ml[((size_t)b * 4 + (tid >> 5)) * 64 + 32 + (tid & 31)];
        Ss[tid] = S[((size_t)b * 4 + (tid >> 5)) * 32 + (tid & 31)];
    }
    const float inv_n = 1.0f / (float)n;
    const int h = wave;                    // wave = head; the four waves of a workgroup share a 32-pixel tile
    unsigned char* d2t = xt + 32 * 128;    // FD: do2 tile [32 pixels][128 B] (the region the data-gradient partials use at the END of a tile)
    for (int p0 = blockIdx.x * 32; p0 < n; p0 += gridDim.x * 32) {
        // FD: this head's A fragments of Wo^T (rows e = 32 h .., k = the 64 channels of do2): re-read from L2 per tile (64 bytes per lane, in
        // flight with the tile's DMA) -- held for the whole launch they push the kernel past 256 registers, i.e. to one wave per SIMD
        bf16x8 wof[FD ? 4 : 1];
        if constexpr (FD) {
            asm volatile("" ::: "memory");     // (keeps the loop-invariant loads below inside the loop)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wof[ks] = *(const bf16x8*)(wot + ((size_t)(ks * 2 + half) * 128 + h * 32 + l31) * 8);
        }
        __syncthreads();                   // previous tile's copy-out is done (and, first time, the matrices are staged)
        // one direct global->LDS instruction per pixel row: lanes 0..47 fetch the qkv row, lanes 48..63 the dout row
        for (int px = wave; px < 32; px += 4) {
            const size_t p = (size_t)b * n + min(p0 + px, n - 1);
            const bf16_t* src;
            if constexpr (RQ) src = qkv + p * 384 + (lane < 16 || lane >= 48 ? 128 + (lane & 15) * 8 : lane * 8);  // (q part and dout part: filler = the k units)
            else if constexpr (FD) src = lane < 48 ? qkv + p * 384 + lane * 8 : qkv + p * 384 + (lane - 48) * 8;     // (lanes 48..63: filler, overwritten by dout below)
            else src = lane < 48 ? qkv + p * 384 + lane * 8 : dout + p * 128 + (lane - 48) * 8;
            __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(st + px * LB_PITCH), 16, 0, 0);
        }
        if constexpr (FD) {                // do2 rows 8 wave .. 8 wave + 7: lane -> row lane >> 3, 16-byte unit lane & 7
            const size_t p = (size_t)b * n + min(p0 + wave * 8 + (lane >> 3), n - 1);
            __builtin_amdgcn_global_load_lds(dout + p * 64 + (lane & 7) * 8, (__attribute__((address_space(3))) void*)(d2t + wave * 1024), 16, 0, 0);
        }
        if constexpr (FUSE) {              // xn rows 8 wave .. 8 wave + 7 of the tile: lane -> row lane >> 3, 16-byte unit lane & 7
            const size_t p = (size_t)b * n + min(p0 + wave * 8 + (lane >> 3), n - 1);
            __builtin_amdgcn_global_load_lds(xn + p * 64 + (lane & 7) * 8, (__attribute__((address_space(3))) void*)(xt + wave * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned char* row = st + l31 * LB_PITCH;
        if constexpr (FD) {                // dout of this head = Wo^T do2: rows e, cols pixel; written to the tile where the loaded dout would be
            f32x16 da;
#pragma unroll
            for (int r = 0; r < 16; ++r) da[r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wof[ks], *(const bf16x8*)(d2t + l31 * 128 + (ks * 16 + half * 8) * 2), da, 0, 0, 0);
            uint2 dq4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) dq4[g] = make_uint2(f2bf2(da[4 * g], da[4 * g + 1]), f2bf2(da[4 * g + 2], da[4 * g + 3]));
            lc_store_head((bf16_t*)row + 384 + h * 32, dq4, half, true);        // (read back below by the same wave: LDS is in order per wave)
        }
        // everything this wave needs of its head, before any of it is overwritten
        const bf16x8 g0 = *(const bf16x8*)(row + 768 + (h * 32 + half * 8) * 2), g1 = *(const bf16x8*)(row + 768 + (h * 32 + 16 + half * 8) * 2);
        const bf16x8 v0 = *(const bf16x8*)(row + 512 + (h * 32 + half * 8) * 2), v1 = *(const bf16x8*)(row + 512 + (h * 32 + 16 + half * 8) * 2);
        float q[16], kc[16], kk[16];
        if constexpr (RQ) {                // q of this head from the xn tile: the MFMA output IS the accumulator layout; bf16-rounded like the stored tensor
            f32x16 qa;
#pragma unroll
            for (int r = 0; r < 16; ++r) qa[r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                qa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(wq + ((size_t)(ks * 2 + half) * 384 + h * 32 + l31) * 8),
                                                             *(const bf16x8*)(xt + l31 * 128 + (ks * 16 + half * 8) * 2), qa, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 16; ++j) q[j] = bf2f(f2bf(qa[j]));
        } else {
            lc_load_head((const bf16_t*)row + h * 32, half, q);           // accumulator layout
        }
        lc_load_head((const bf16_t*)row + 128 + h * 32, half, kc);
        lc_unpack8(*(const u32x4*)(row + 256 + (h * 32 + half * 8) * 2), &kk[0]);      // operand layout
        lc_unpack8(*(const u32x4*)(row + 256 + (h * 32 + 16 + half * 8) * 2), &kk[8]);
        f32x16 acc;
        uint2 qo[4];
        // ---- dq
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lc_afrag(cA, h, 0, l31, half), g0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lc_afrag(cA, h, 1, l31, half), g1, acc, 0, 0, 0);
        {
            float mx = q[0];
#pragma unroll
            for (int j = 1; j < 16; ++j) mx = fmaxf(mx, q[j]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) { q[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(q[j], 1.4426950408889634f, -mx * 1.4426950408889634f)); sum += q[j]; }
            sum += __shfl_xor(sum, 32, 64);
            const float rs = __builtin_amdgcn_rcpf(sum);
            float t = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) { q[j] *= rs; acc[j] *= LC_SCALE; t += q[j] * acc[j]; }
            t += __shfl_xor(t, 32, 64);
#pragma unroll
            for (int g = 0; g < 4; ++g)
                qo[g] = make_uint2(f2bf2(q[4 * g] * (acc[4 * g] - t), q[4 * g + 1] * (acc[4 * g + 1] - t)),
                                   f2bf2(q[4 * g + 2] * (acc[4 * g + 2] - t), q[4 * g + 3] * (acc[4 * g + 3] - t)));
            lc_store_head((bf16_t*)row + h * 32, qo, half, true);
        }
        // ---- dk
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lc_afrag(dA, h, 0, l31, half), v0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lc_afrag(dA, h, 1, l31, half), v1, acc, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = h * 32 + 8 * g + 4 * half;
            const float4 m4 = *(const float4*)&Ms[d0], l4 = *(const float4*)&Li[d0], s4 = *(const float4*)&Ss[d0];
            const float k0 = __expf(kc[4 * g] - m4.x) * l4.x, k1 = __expf(kc[4 * g + 1] - m4.y) * l4.y;
            const float k2 = __expf(kc[4 * g + 2] - m4.z) * l4.z, k3 = __expf(kc[4 * g + 3] - m4.w) * l4.w;
            qo[g] = make_uint2(f2bf2(k0 * (acc[4 * g] * inv_n - s4.x), k1 * (acc[4 * g + 1] * inv_n - s4.y)),
                               f2bf2(k2 * (acc[4 * g + 2] * inv_n - s4.z), k3 * (acc[4 * g + 3] * inv_n - s4.w)));
        }
        lc_store_head((bf16_t*)row + 128 + h * 32, qo, half, true);
        // ---- dv
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int d0 = h * 32 + 16 * s2 + 8 * half;
#pragma unroll
            for (int j = 0; j < 8; ++j) kk[8 * s2 + j] = __expf(kk[8 * s2 + j] - Ms[d0 + j]) * Li[d0 + j];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lc_afrag(dT, h, 0, l31, half), lc_frag(&kk[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lc_afrag(dT, h, 1, l31, half), lc_frag(&kk[8]), acc, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g)
            qo[g] = make_uint2(f2bf2(acc[4 * g] * inv_n, acc[4 * g + 1] * inv_n), f2bf2(acc[4 * g + 2] * inv_n, acc[4 * g + 3] * inv_n));
        lc_store_head((bf16_t*)row + 256 + h * 32, qo, half, true);
        __syncthreads();
        if constexpr (!FUSE) {
            // copy-out: 32 pixels x 48 units of dqkv, consecutive lanes = consecutive 16-byte units (pixel rows are contiguous)
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int id = tid + k * 256, px = id / 48, u = id - px * 48;
                if (p0 + px < n) *(u32x4*)(dqkv + ((size_t)b * n + p0 + px) * 384 + u * 8) = *(const u32x4*)(st + px * LB_PITCH + u * 16);
            }
        } else {
            // pixels past the end of the sample carry the clamped last row: they must not reach dW (dxn rows are simply not stored)
            if (p0 + 32 > n) {
                for (int id = tid; id < 32 * 48; id += 256) {
                    const int px = id / 48, u = id - px * 48;
                    if (p0 + px >= n) *(u32x4*)(st + px * LB_PITCH + u * 16) = u32x4{0u, 0u, 0u, 0u};
                }
                __syncthreads();
            }
            // ---- dW += xn^T dqkv: rows = ci, cols = co; k = pixel (two k-steps of 16)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 xf[2], yf[3];
#pragma unroll
                for (int a = 0; a < 2; ++a) xf[a] = tr_frag(xt + (ks * 16) * 128 + a * 64, 128, lane);
#pragma unroll
                for (int j = 0; j < 3; ++j) yf[j] = tr_frag(st + (ks * 16) * LB_PITCH + (wave * 96 + j * 32) * 2, LB_PITCH, lane);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int j = 0; j < 3; ++j) wacc[a][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[a], yf[j], wacc[a][j], 0, 0, 0);
            }
            // ---- dxn = W^T dqkv: rows = ci (block wave & 1), cols = pixel; k = the 384 qkv channels, split between the wave pairs
            //      (waves 2, 3 take k-steps 12..23 and hand their partial tile to waves 0, 1 through LDS)
            {
                const int mt = wave & 1, kh = wave >> 1;
                f32x16 dacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) dacc[r] = 0.0f;
                const bf16_t* wrow = wt + ((size_t)(kh * 24 + half) * 64 + mt * 32 + l31) * 8;
                const unsigned char* drow = st + l31 * LB_PITCH + half * 16 + kh * 12 * 32;
#pragma unroll
                for (int ks = 0; ks < 12; ++ks) {
                    const bf16x8 wf = *(const bf16x8*)(wrow + (size_t)ks * 2 * 64 * 8);
                    const bf16x8 df = *(const bf16x8*)(drow + ks * 32);
                    dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, df, dacc, 0, 0, 0);
                }
                float* part = (float*)(xt + 32 * 128) + (size_t)mt * 16 * 64;        // [m-tile][register][lane]
                if (kh == 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) part[r * 64 + lane] = dacc[r];
                }
                __syncthreads();
                if (kh == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) dacc[r] += part[r * 64 + lane];
                    uint2 qd[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) qd[g] = make_uint2(f2bf2(dacc[4 * g], dacc[4 * g + 1]), f2bf2(dacc[4 * g + 2], dacc[4 * g + 3]));
                    lc_store_head(dxn + ((size_t)b * n + min(p0 + l31, n - 1)) * 64 + mt * 32, qd, half, p0 + l31 < n);
                }
            }
        }
    }
    if constexpr (FUSE) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float* d = dw + ((size_t)a * 32) * 384 + wave * 96 + j * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ci = (r & 3) + 8 * (r >> 2) + 4 * half;
                    gacc_add(d + (size_t)ci * 384, wacc[a][j][r]);
                }
            }
    }
}

// combine of backward pass 1 (also used by the forward's la_ctx_combine in blocks.hip for the ctx partials)
// wo != nullptr (lc_dctx_partial_kernel<true>): the partials are QD[32 d][64 c] = sum_n qs (x) do2; dctx[d][e] = sum_c QD[d][c] Wo[c][h 32 + e]
// with Wo the prepared to_out.0 weights ([16 e-octets][64 c][8], bf16 -- the values the separate data-gradient conv multiplied by)
__global__ void __launch_bounds__(256) lc_bwd_combine_kernel(const float* __restrict__ partial, const float* __restrict__ ctx, float* __restrict__ dctx,
                                                             float* __restrict__ S, int nparts, const bf16_t* __restrict__ wo, float* __restrict__ dwo) {
    __shared__ float prod[1024];
    __shared__ float qd[2048];
    const int tid = threadIdx.x, bh = blockIdx.x, h = bh & 3;
    const int per = wo ? 2048 : 1024;
    for (int i = tid; i < per; i += 256) {
        float a = 0.0f;
        for (int c0 = 0; c0 < nparts; c0 += 8) {        // eight loads in flight; the sum stays in part order
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = partial[((size_t)bh * nparts + min(c0 + k, nparts - 1)) * per + i];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (c0 + k < nparts) a += v[k];
        }
        if (wo) {
            qd[i] = a;
        } else {
            dctx[(size_t)bh * 1024 + i] = a;
            prod[i] = a * ctx[(size_t)bh * 1024 + i];
        }
    }
    if (wo) {
        __syncthreads();
        // the to_out.0 WEIGHT gradient from the same partials: dWo[c][e] = sum_n do2[n][c] ao[n][e] with ao = ctx^T qs (what the forward's
        // second pass computed), i.e. sum_d QD[d][c] ctx[d][e] -- no pass over do2 and ao, and the forward need not keep ao at all.
        // dwo: the fp32 accumulator [ci = e (128)][co = c (64)] of the to_out.0 conv; every (sample, head) adds its 32 x 64 block.
        if (dwo) {
            for (int i = tid; i < 2048; i += 256) {
                const int c = i & 63, e = i >> 6;
                float a = 0.0f;
                for (int d = 0; d < 32; ++d) a += qd[d * 64 + c] * ctx[(size_t)bh * 1024 + d * 32 + e];
                gacc_add(dwo + (size_t)(h * 32 + e) * 64 + c, a);
            }
        }
        for (int i = tid; i < 1024; i += 256) {
            const int d = i >> 5, e = h * 32 + (i & 31);
            float a = 0.0f;
            for (int c = 0; c < 64; ++c) a += qd[d * 64 + c] * bf2f(wo[((size_t)(e >> 3) * 64 + c) * 8 + (e & 7)]);
            dctx[(size_t)bh * 1024 + i] = a;
            prod[i] = a * ctx[(size_t)bh * 1024 + i];
        }
    }
    __syncthreads();
    if (tid < 32) {
        float s = 0.0f;
        for (int e = 0; e < 32; ++e) s += prod[tid * 32 + e];
        S[(size_t)bh * 32 + tid] = s;
    }
}

static inline void lc_parts(int B, int n, int& nparts, int& span) {
    // enough workgroups to fill the chip in the reduction passes; spans are multiples of the 128-pixel chunk
    nparts = cdiv(768, B);
    const int maxp = cdiv(n, 512);
    if (nparts > maxp) nparts = maxp;
    if (nparts < 1) nparts = 1;
    span = cdiv(cdiv(n, nparts), LC_CH) * LC_CH;
    nparts = cdiv(n, span);
}
int la_parts(int B, int n) { int np, sp; lc_parts(B, n, np, sp); return np; }
// parts of the forward's first pass (whichever of its two kernels runs): sizes the partial buffer
int la_fwd_parts(int B, int n) { const int a = la_parts(B, n), f = la_fused_blocks(n, B); return a > f ? a : f; }

void launch_la_ctx_combine(const float* partial, float* ctx, int B, int nparts, float inv_n, float* ml_out, hipStream_t s);   // blocks.hip
int launch_la_ctx_stored(const bf16_t* qkv, float* partial, int B, int n, hipStream_t s);                                       // la_fused.hip

int k_linear_attention_core(const bf16_t* qkv, float* partial, float* ctx, bf16_t* out, int B, int n, hipStream_t s, float* ml_out,
                            const bf16_t* wo, const float* bo, bf16_t* o2, int C, const bf16_t* xn, const bf16_t* wq) {
    int nparts, span;
    lc_parts(B, n, nparts, span);
    // pass 1: the accumulator-layout kernel on the stored k, v (la_fused.hip la_ctx_stored_kernel); OFD_LA_CTX_STORED=0: the LDS-scan kernel above
    static const int stored = getenv("OFD_LA_CTX_STORED") ? atoi(getenv("OFD_LA_CTX_STORED")) : 1;
    if (stored) nparts = launch_la_ctx_stored(qkv, partial, B, n, s);
    else lc_ctx_partial_kernel<<<dim3(nparts, B), 256, 0, s>>>(qkv, partial, n, span, nparts);
    launch_la_ctx_combine(partial, ctx, B, nparts, 1.0f / (float)n, ml_out, s);
    int gx = cdiv(n, 32);
    if (gx > 2048) gx = 2048;
    lc_out_kernel<<<dim3(gx, B), 256, 0, s>>>(qkv, ctx, out, n, wo, bo, o2, C, xn, wq);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

size_t la_bwd_workspace_floats(int B, int n) { return (size_t)B * 4 * ((size_t)la_parts(B, n) * 2048 + 1024 + 32); }      // (2048: the [32][64] partials of the do2 form)

int k_linear_attention_core_bwd(const bf16_t* qkv, const bf16_t* dout, const float* ctx, const float* ml, bf16_t* dqkv, float* workspace, int B, int n,
                                hipStream_t s, const bf16_t* xn, const bf16_t* wt, float* dw, bf16_t* dxn, const bf16_t* wo_fwd, const bf16_t* wo_t,
                                float* dwo, float* dbo, const bf16_t* wq) {
    // wo_fwd / wo_t != nullptr (only with xn): `dout` is do2 [pixel][64], the gradient of the to_out.0 output (see lc_dctx_partial_kernel<true>)
    const bool fd = wo_fwd != nullptr && wo_t != nullptr && xn != nullptr;
    int nparts, span;
    lc_parts(B, n, nparts, span);
    float* partial = workspace;
    float* dctx = partial + (size_t)B * 4 * nparts * (fd ? 2048 : 1024);
    float* S = dctx + (size_t)B * 4 * 1024;
    if (fd && wq) {
        constexpr int DQ_LDS = RQ_CH * (256 + 128 + 128);
        static bool a2 = false;
        if (!a2) { OFD_HIP(hipFuncSetAttribute((const void*)lc_dctx_partial_rq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DQ_LDS)); a2 = true; }
        lc_dctx_partial_rq_kernel<<<dim3(nparts, B), 256, DQ_LDS, s>>>(xn, wq, dout, partial, n, span, nparts, dbo);
    } else if (fd) lc_dctx_partial_kernel<true><<<dim3(nparts, B), 256, 0, s>>>(qkv, dout, partial, n, span, nparts, dbo);
    else lc_dctx_partial_kernel<false><<<dim3(nparts, B), 256, 0, s>>>(qkv, dout, partial, n, span, nparts, nullptr);
    lc_bwd_combine_kernel<<<B * 4, 256, 0, s>>>(partial, ctx, dctx, S, nparts, fd ? wo_fwd : nullptr, fd ? dwo : nullptr);
    int gx = cdiv(n, 32);
    if (gx > 2048) gx = 2048;
    static bool attr = false;
    if (!attr) {
        OFD_HIP(hipFuncSetAttribute((const void*)lc_bwd_apply_kernel<false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LB_LDS));
        OFD_HIP(hipFuncSetAttribute((const void*)lc_bwd_apply_kernel<true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LB_LDS_FUSE));
        OFD_HIP(hipFuncSetAttribute((const void*)lc_bwd_apply_kernel<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LB_LDS_FUSE));
        OFD_HIP(hipFuncSetAttribute((const void*)lc_bwd_apply_kernel<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LB_LDS_FUSE));
        attr = true;
    }
    if (xn) {       // 64-channel block: the to_qkv backward fused in; fewer, longer-lived workgroups (each adds a 64 x 384 dW tile with atomics)
        if (gx > 512) gx = 512;
        if (fd && wq) lc_bwd_apply_kernel<true, true, true><<<dim3(gx, B), 256, LB_LDS_FUSE, s>>>(qkv, dout, ctx, dctx, ml, S, nullptr, n, xn, wt, dw, dxn, wo_t, wq);
        else if (fd) lc_bwd_apply_kernel<true, true, false><<<dim3(gx, B), 256, LB_LDS_FUSE, s>>>(qkv, dout, ctx, dctx, ml, S, nullptr, n, xn, wt, dw, dxn, wo_t, nullptr);
        else lc_bwd_apply_kernel<true, false, false><<<dim3(gx, B), 256, LB_LDS_FUSE, s>>>(qkv, dout, ctx, dctx, ml, S, nullptr, n, xn, wt, dw, dxn, nullptr, nullptr);
    } else {
        lc_bwd_apply_kernel<false, false, false><<<dim3(gx, B), 256, LB_LDS, s>>>(qkv, dout, ctx, dctx, ml, S, dqkv, n, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    }
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

OFD_DET_DEFINE_SETTER(det_set_ctx_la_core)

}  // namespace ofd
