// The non-convolution kernels of the UNet forward (denoising_diffusion.py), NHWC bf16 activations,
// fp32 arithmetic: input packing, time MLP, GroupNorm finalisation, ResnetBlock output,
// channel LayerNorm, LinearAttention (context pass + output pass), flash attention for the mid
// block and the final 1x1 convolution.
#include <cstdlib>
#include "blocks.h"

namespace ofd {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ float silu_f(float y) { return y * __builtin_amdgcn_rcpf(1.0f + __expf(-y)); }
__device__ __forceinline__ uint32_t pack2(float a, float b) { return f2bf2(a, b); }
__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[2 * j] = bf2f((bf16_t)(w[j] & 0xffffu));
        f[2 * j + 1] = bf2f((bf16_t)(w[j] >> 16));
    }
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    return make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
}

// ---- DD:368: cat(x, cond) -> NHWC bf16 padded to CP = 8 or 16 channels (input of the 7x7 init_conv) --------
template <int CP>
__global__ void __launch_bounds__(256) pack_input_kernel(const float* __restrict__ x, int Cx, const float* __restrict__ cond, int Cc,
                                                         bf16_t* __restrict__ out, int B, size_t plane) {
    const size_t total = (size_t)B * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / plane, pix = i % plane;
        float v[CP];
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            float t = 0.0f;
            if (c < Cx) t = x[(n * Cx + c) * plane + pix];
            else if (c < Cx + Cc) t = cond[(n * Cc + (c - Cx)) * plane + pix];
            v[c] = t;
        }
        uint4* o = (uint4*)(out + i * CP);
#pragma unroll
        for (int k = 0; k < CP / 8; ++k)
            o[k] = make_uint4(pack2(v[8 * k], v[8 * k + 1]), pack2(v[8 * k + 2], v[8 * k + 3]), pack2(v[8 * k + 4], v[8 * k + 5]), pack2(v[8 * k + 6], v[8 * k + 7]));
    }
}

// ---- DD:139-151 + DD:319-324: sinusoidal embedding -> Linear -> GELU(erf) -> Linear; output SiLU(temb)
// too, which is what every ResnetBlock's mlp consumes (DD:193-196).  One workgroup per sample.
__global__ void __launch_bounds__(256) time_mlp_kernel(const int64_t* __restrict__ t, const float* __restrict__ w1, const float* __restrict__ b1,
                                                       const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ temb,
                                                       float* __restrict__ temb_silu, int dim) {
    __shared__ float emb[256], h1[1024];
    const int b = blockIdx.x, tid = threadIdx.x, tdim = dim * 4, half_dim = dim / 2;
    const float tv = (float)t[b];
    for (int i = tid; i < half_dim; i += 256) {
        const float k = (float)(9.210340371976184 / (double)(half_dim - 1));   // math.log(10000) / (half_dim - 1), DD:147
        const float f = expf((float)i * -k);
        emb[i] = sinf(tv * f);
        emb[half_dim + i] = cosf(tv * f);
    }
    __syncthreads();
    for (int j = tid; j < tdim; j += 256) {
        float a = b1[j];
        for (int k = 0; k < dim; ++k) a += w1[(size_t)j * dim + k] * emb[k];
        h1[j] = 0.5f * a * (1.0f + erff(a * 0.70710678118654752f));
    }
    __syncthreads();
    for (int j = tid; j < tdim; j += 256) {
        float a = b2[j];
        for (int k = 0; k < tdim; ++k) a += w2[(size_t)j * tdim + k] * h1[k];
        temb[(size_t)b * tdim + j] = a;
        temb_silu[(size_t)b * tdim + j] = a / (1.0f + expf(-a));
    }
}

// ss[b][off + j] = Linear(SiLU(temb))[j] for every ResnetBlock (DD:205-208); grid (B, n_blocks)
__global__ void __launch_bounds__(256) block_mlp_kernel(const float* __restrict__ temb_silu, const MlpDesc* __restrict__ descs,
                                                        float* __restrict__ ss, int tdim, int ss_stride) {
    __shared__ float e[1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    const MlpDesc d = descs[blockIdx.y];
    for (int k = tid; k < tdim; k += 256) e[k] = temb_silu[(size_t)b * tdim + k];
    __syncthreads();
    for (int j = tid; j < d.n_out; j += 256) {
        float a = d.bias[j];
        const float* wr = d.weight + (size_t)j * tdim;
        if ((tdim & 3) == 0) {               // (a row is 16-byte aligned: 4 weights per load, 8 loads in flight; same summation order)
#pragma unroll 8
            for (int k = 0; k < tdim; k += 4) {
                const float4 w4 = *(const float4*)(wr + k);
                a += w4.x * e[k];
                a += w4.y * e[k + 1];
                a += w4.z * e[k + 2];
                a += w4.w * e[k + 3];
            }
        } else {
            for (int k = 0; k < tdim; ++k) a += wr[k] * e[k];
        }
        ss[(size_t)b * ss_stride + d.offset + j] = a;
    }
}

// ---- GroupNorm(8) statistics -> per-(sample, channel) affine (DD:181-185) -------------------------
// partial: [B][8 groups][slots = tiles * 4 waves][C/64][2] from the conv epilogue (gn_partial_index, conv_params.h).  y = x*a + s with
//   a = gamma*rstd*(scale+1), s = (beta - mean*rstd*gamma)*(scale+1) + shift.   grid (B, 8)
constexpr int GNF_NT = 1024;      // threads per (sample, group): the 7040 entries of a full-resolution group are one round of loads
__global__ void __launch_bounds__(GNF_NT) gn_finalize_kernel(const float* __restrict__ partial, int tiles, int C, double count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ ss, int ss_stride, int ss_offset,
                                                          float* __restrict__ a_out, float* __restrict__ s_out, float* __restrict__ stats_out) {
    const int b = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
    const int gs = C / 8, octs = gs / 8;
    // the (sum, sum of squares) pairs of this (sample, group) are one contiguous run; GU independent loads per thread are in flight together
    // (walked one load at a time the run is a latency chain)
    constexpr int GU = 8;
    double s1 = 0.0, s2 = 0.0;
    const int total = tiles * octs;
    const float2* run = (const float2*)partial + ((size_t)b * 8 + g) * total;
    for (int i0 = tid; i0 < total; i0 += GNF_NT * GU) {
        float2 v[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) v[u] = run[min(i0 + u * GNF_NT, total - 1)];
#pragma unroll
        for (int u = 0; u < GU; ++u)
            if (i0 + u * GNF_NT < total) { s1 += (double)v[u].x; s2 += (double)v[u].y; }
    }
    // wave sums first (the order of the additions is fixed: deterministic), then one value per wave through LDS
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    __shared__ double r1[GNF_NT / 64], r2[GNF_NT / 64];
    if ((tid & 63) == 0) { r1[tid >> 6] = s1; r2[tid >> 6] = s2; }
    __syncthreads();
    for (int k = GNF_NT / 128; k > 0; k >>= 1) {
        if (tid < k) { r1[tid] += r1[tid + k]; r2[tid] += r2[tid + k]; }
        __syncthreads();
    }
    const double mean_d = r1[0] / count;
    double var_d = r2[0] / count - mean_d * mean_d;
    if (var_d < 0.0) var_d = 0.0;
    const float mean = (float)mean_d, rstd = rsqrtf((float)var_d + 1e-5f);
    if (stats_out && tid == 0) {          // kept by the training forward for the GroupNorm backward
        stats_out[((size_t)b * 8 + g) * 2] = mean;
        stats_out[((size_t)b * 8 + g) * 2 + 1] = rstd;
    }
    for (int c = g * gs + tid; c < (g + 1) * gs; c += GNF_NT) {
        float sc = 0.0f, sh = 0.0f;
        if (ss) {
            sc = ss[(size_t)b * ss_stride + ss_offset + c];
            sh = ss[(size_t)b * ss_stride + ss_offset + C + c];
        }
        const float ga = gamma[c] * rstd;
        a_out[(size_t)b * C + c] = ga * (sc + 1.0f);
        s_out[(size_t)b * C + c] = (beta[c] - mean * ga) * (sc + 1.0f) + sh;
    }
}

// ---- DD:214 with identity res_conv: out = SiLU(h*a + s) + x ---------------------------------------
__global__ void __launch_bounds__(256) resblock_out_kernel(const bf16_t* __restrict__ h, const float* __restrict__ a, const float* __restrict__ s,
                                                           const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int C, size_t pix_per_sample, size_t total_units) {
    const int c8n = C / 8;
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < total_units; u += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = u / c8n;
        const int c = (int)(u % c8n) * 8;
        const size_t b = pix / pix_per_sample;
        float hv[8], xv[8], av[8], sv[8];
        unpack8(*(const uint4*)(h + u * 8), hv);
        unpack8(*(const uint4*)(x + u * 8), xv);
        *(float4*)&av[0] = *(const float4*)(a + b * C + c);
        *(float4*)&av[4] = *(const float4*)(a + b * C + c + 4);
        *(float4*)&sv[0] = *(const float4*)(s + b * C + c);
        *(float4*)&sv[4] = *(const float4*)(s + b * C + c + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) hv[j] = silu_f(hv[j] * av[j] + sv[j]) + xv[j];
        *(uint4*)(out + u * 8) = pack8(hv);
    }
}

// ---- DD:116-125 channel LayerNorm (gain only), optional "+ residual" (Residual of DD:81-87) ------
// C/8 lanes cooperate on one pixel (each holds 8 channels); a wave covers 512/C pixels at a time.
__global__ void __launch_bounds__(256) layernorm_c_kernel(const bf16_t* __restrict__ x, const float* __restrict__ g, const bf16_t* __restrict__ res,
                                                          bf16_t* __restrict__ out, int C, float eps, size_t npix) {
    const int lpp = C / 8;                         // lanes per pixel: 8,16,32,64
    const int lane = threadIdx.x & 63;
    const int sub = lane % lpp, slot = lane / lpp, ppw = 64 / lpp;
    const size_t wave_global = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    float gv[8];
    *(float4*)&gv[0] = *(const float4*)(g + sub * 8);
    *(float4*)&gv[4] = *(const float4*)(g + sub * 8 + 4);
    const float inv_c = 1.0f / (float)C;
    for (size_t p0 = wave_global * ppw; p0 < npix; p0 += nwaves * ppw) {
        const size_t p = p0 + slot;
        const bool ok = p < npix;
        float v[8];
        if (ok) unpack8(*(const uint4*)(x + p * C + sub * 8), v);
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.0f;
        }
        float s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        for (int o = 1; o < lpp; o <<= 1) s += __shfl_xor(s, o, 64);
        const float mean = s * inv_c;
        float q = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] -= mean; q += v[j] * v[j]; }
        for (int o = 1; o < lpp; o <<= 1) q += __shfl_xor(q, o, 64);
        const float rstd = rsqrtf(q * inv_c + eps);
        if (ok) {
            float r[8];
            if (res) unpack8(*(const uint4*)(res + p * C + sub * 8), r);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] * rstd * gv[j] + (res ? r[j] : 0.0f);
            *(uint4*)(out + p * C + sub * 8) = pack8(v);
        }
    }
}

// combine the partials: ctx[bh][d][e] = sum_c exp(m_c - M) ctx_c / (sum_c exp(m_c - M) l_c) / n
__global__ void __launch_bounds__(256) la_ctx_combine_kernel(const float* __restrict__ partial, float* __restrict__ ctx, int nparts, float inv_n, float* __restrict__ ml_out) {
    // all 256 threads on the maxima / normalisers (thread -> d = tid & 31, parts tid >> 5, + 8, ...), the per-part weights exp(m_c - M)
    // once into LDS, eight loads in flight in the accumulation: one thread per d and one load at a time made this 57 us per launch
    __shared__ float M[32], Linv[32], red[8][32];
    extern __shared__ float w_s[];                      // [nparts][32]
    const int tid = threadIdx.x, bh = blockIdx.x, dd = tid & 31, grp = tid >> 5;
    const float* base = partial + (size_t)bh * nparts * 1088;
    float mx = -3.0e38f;
    for (int c = grp; c < nparts; c += 8) mx = fmaxf(mx, base[(size_t)c * 1088 + dd]);
    red[grp][dd] = mx;
    __syncthreads();
    if (tid < 32) {
        float m = red[0][tid];
#pragma unroll
        for (int g = 1; g < 8; ++g) m = fmaxf(m, red[g][tid]);
        M[tid] = m;
    }
    __syncthreads();
    const bool use_ws = nparts <= 256;               // (the launcher sizes w_s for at most 256 parts; beyond that the weights are recomputed)
    float l = 0.0f;
    for (int c = grp; c < nparts; c += 8) {
        const float w = __expf(base[(size_t)c * 1088 + dd] - M[dd]);
        if (use_ws) w_s[c * 32 + dd] = w;
        l += base[(size_t)c * 1088 + 32 + dd] * w;
    }
    __syncthreads();
    red[grp][dd] = l;
    __syncthreads();
    if (tid < 32) {
        float t = red[0][tid];
#pragma unroll
        for (int g = 1; g < 8; ++g) t += red[g][tid];
        Linv[tid] = 1.0f / t;
        if (ml_out && blockIdx.y == 0) {  // softmax-over-pixels normalisers, kept for the backward
            ml_out[(size_t)bh * 64 + tid] = M[tid];
            ml_out[(size_t)bh * 64 + 32 + tid] = 1.0f / t;
        }
    }
    __syncthreads();
    {   // grid (B * 4, 4): this workgroup's quarter of the 32 x 32 context (the normalisers above are recomputed by each of the four)
        const int i = blockIdx.y * 256 + tid, d = i >> 5;
        float a = 0.0f;
        for (int c0 = 0; c0 < nparts; c0 += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = base[(size_t)min(c0 + k, nparts - 1) * 1088 + 64 + i];
            if (use_ws) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (c0 + k < nparts) a += v[k] * w_s[(c0 + k) * 32 + d];
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (c0 + k < nparts) a += v[k] * __expf(base[(size_t)(c0 + k) * 1088 + d] - M[d]);
            }
        }
        ctx[(size_t)bh * 1024 + i] = a * Linv[d] * inv_n;
    }
}

// ---- mid-block attention (DD:256-268), flash style, d = 32, 4 heads, MFMA 32x32x16 bf16 -----------
// workgroup = 4 waves x 32 queries of one (sample, head); K/V tiles of 64 keys shared through LDS.
// S^T = K.Q^T puts the query on the lane, so row max / sum are lane-local (+1 cross-half shuffle)
// and the exponentiated accumulator is directly the B operand of O^T += V^T.P^T.
constexpr int FA_KT = 64;
__global__ void __launch_bounds__(256) flash_attn_d32_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int n, float scale, float* __restrict__ lse) {
    __shared__ __attribute__((aligned(16))) unsigned char k_lds[FA_KT * 80];      // [key][32 d], rows padded to 80 B
    __shared__ __attribute__((aligned(16))) unsigned char vt_lds[32 * 144];       // [d][64 keys permuted], rows padded to 144 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int bh = blockIdx.y, b = bh / 4, h = bh % 4;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const bf16_t* base = qkv + (size_t)b * n * 384;

    bf16x8 qf[2];
    {
        const int q = min(q0 + l31, n - 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) qf[s] = *(const bf16x8*)(base + (size_t)q * 384 + h * 32 + s * 16 + half * 8);
    }
    f32x16 o_acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[r] = 0.0f;
    float m_run = -3.0e38f, l_run = 0.0f;

    for (int j0 = 0; j0 < n; j0 += FA_KT) {
        __syncthreads();     // previous tile fully consumed
        {
            // stage K [key][d] and V^T [d][slot]: thread -> key tid/4, 16-B unit tid%4 (8 d's)
            const int key = tid >> 2, u = tid & 3;
            const int kg = min(j0 + key, n - 1);
            const uint4 kv = *(const uint4*)(base + (size_t)kg * 384 + 128 + h * 32 + u * 8);
            *(uint4*)(k_lds + key * 80 + u * 16) = kv;
            const uint4 vv = *(const uint4*)(base + (size_t)kg * 384 + 256 + h * 32 + u * 8);
            // key = kb*32 + 16 s + 8 a + 4 hh + bb  ->  slot = kb*32 + s*16 + hh*8 + a*4 + bb
            const int kb = key >> 5, kk = key & 31, s = kk >> 4, a = (kk >> 3) & 1, hh = (kk >> 2) & 1, bb = kk & 3;
            const int slot = kb * 32 + s * 16 + hh * 8 + a * 4 + bb;
            const uint32_t w[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                *(bf16_t*)(vt_lds + (u * 8 + 2 * j) * 144 + slot * 2) = (bf16_t)(w[j] & 0xffffu);
                *(bf16_t*)(vt_lds + (u * 8 + 2 * j + 1) * 144 + slot * 2) = (bf16_t)(w[j] >> 16);
            }
        }
        __syncthreads();
        f32x16 s_acc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_acc[kb][r] = 0.0f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 kf = *(const bf16x8*)(k_lds + (kb * 32 + l31) * 80 + (s * 16 + half * 8) * 2);
                s_acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], s_acc[kb], 0, 0, 0);
            }
        }
        // online softmax; key index of register r: j0 + kb*32 + (r&3) + 8*(r>>2) + 4*half.  The running maximum is kept on the RAW scores
        // (scale > 0) and exp(scale (s - m)) is one fma + v_exp per element, c = scale log2(e); keys past the end exist in the last tile only.
        // (the kernel is VALU-bound: 8 MFMAs against 32 score elements per lane and tile -- 9 -> 5 VALU instructions per element)
        if (j0 + FA_KT > n) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (j0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half >= n) s_acc[kb][r] = -3.0e38f;
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s_acc[kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float c = scale * 1.4426950408889634f, m2 = m_new * c;
        const float alpha = __builtin_amdgcn_exp2f(__builtin_fmaf(m_run, c, -m2));
        m_run = m_new;
        float psum = 0.0f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[kb][r], c, -m2));
                s_acc[kb][r] = p;
                psum += p;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[r] *= alpha;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (__bf16)s_acc[kb][8 * s + j];
                const bf16x8 vf = *(const bf16x8*)(vt_lds + l31 * 144 + (kb * 32 + s * 16 + half * 8) * 2);
                o_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o_acc, 0, 0, 0);
            }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int q = q0 + l31;
    if (lse && q < n && half == 0) lse[(size_t)bh * n + q] = m_run * scale + __logf(l_tot);     // kept for the backward (m_run: raw scores)
    if (q < n) {
        bf16_t* dst = out + ((size_t)b * n + q) * 128 + h * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint2 pk = make_uint2(pack2(o_acc[4 * g] * inv, o_acc[4 * g + 1] * inv), pack2(o_acc[4 * g + 2] * inv, o_acc[4 * g + 3] * inv));
            *(uint2*)(dst + 8 * g + 4 * half) = pk;
        }
    }
}

// ---- DD:361,417: final 1x1 conv (fp32 weights) -> NCHW fp32 -------------------------------------
// C/8 lanes cooperate on one pixel (16 B each, so a wave reads whole 128-B lines); shuffle-reduce.
__global__ void __launch_bounds__(256) final_conv_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ out, int C, int out_dim, size_t plane, size_t total) {
    const int lpp = C / 8, lane = threadIdx.x & 63, sub = lane % lpp, slot = lane / lpp, ppw = 64 / lpp;
    float wv[4][8];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[o][j] = (o < out_dim) ? w[o * C + sub * 8 + j] : 0.0f;
    const size_t wave_global = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t p0 = wave_global * ppw; p0 < total; p0 += nwaves * ppw) {
        const size_t i = min(p0 + slot, total - 1);
        float f[8];
        unpack8(*(const uint4*)(x + i * C + sub * 8), f);
        float acc[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            acc[o] = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[o] += f[j] * wv[o][j];
            for (int k = 1; k < lpp; k <<= 1) acc[o] += __shfl_xor(acc[o], k, 64);
        }
        if (sub < out_dim && p0 + slot < total) {
            const size_t n = i / plane, pix = i % plane;
            float r = acc[0];
            if (sub == 1) r = acc[1];
            if (sub == 2) r = acc[2];
            if (sub == 3) r = acc[3];
            out[(n * out_dim + sub) * plane + pix] = r + bias[sub];
        }
    }
}

// debug: NHWC bf16 -> NCHW fp32
__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, int C, size_t plane, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pixc = i / C;
        const int c = (int)(i % C);
        const size_t n = pixc / plane, pix = pixc % plane;
        out[(n * C + c) * plane + pix] = bf2f(x[i]);
    }
}
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int C, size_t plane, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pixc = i / C;
        const int c = (int)(i % C);
        const size_t n = pixc / plane, pix = pixc % plane;
        out[i] = f2bf(x[(n * C + c) * plane + pix]);
    }
}

// cap = 0: one pass per thread.  With a capped, grid-strided launch the resident waves touch addresses spread over the whole tensor;
// uncapped, the blocks in flight form one moving window of a few MB and DRAM pages are streamed through once -- same-box A/B on
// resblock_out: 1.77 ms (cap 4096) -> 1.43 ms per denoise step; GroupNorm backward 16.4 -> 15.2 ms per training step.  Kernels with a
// per-thread preamble (final_conv, pack_input, the layout converters) lose (misc 0.46 -> 0.97 ms) and keep the cap.
// OFD_GRID_CAP restores a cap on the uncapped ones for A/B runs.
static inline int sgrid(size_t total, int block = 256, int cap = 4096) {
    static const int env_cap = getenv("OFD_GRID_CAP") ? atoi(getenv("OFD_GRID_CAP")) : (1 << 22);
    if (cap <= 0) cap = env_cap;
    size_t b = (total + block - 1) / block;
    return (int)(b < 1 ? 1 : (b > (size_t)cap ? cap : b));
}

// ------------------------------------------------------------------------------------ launchers
int k_pack_input(const float* x, int Cx, const float* cond, int Cc, bf16_t* out, int B, int H, int W, hipStream_t s, int cpad) {
    OFD_CHECK_ARG((cpad == 8 || cpad == 16) && Cx + Cc <= cpad && Cx > 0, "pack_input: %d+%d channels into %d", Cx, Cc, cpad);
    if (cpad == 8) pack_input_kernel<8><<<sgrid((size_t)B * H * W), 256, 0, s>>>(x, Cx, cond, Cc, out, B, (size_t)H * W);
    else pack_input_kernel<16><<<sgrid((size_t)B * H * W), 256, 0, s>>>(x, Cx, cond, Cc, out, B, (size_t)H * W);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}
int k_time_mlp(const int64_t* t, const float* w1, const float* b1, const float* w2, const float* b2, float* temb, float* temb_silu, int B, int dim, hipStream_t s) {
    OFD_CHECK_ARG(dim * 4 <= 1024 && dim <= 256, "time_mlp: dim %d too large", dim);
    time_mlp_kernel<<<B, 256, 0, s>>>(t, w1, b1, w2, b2, temb, temb_silu, dim);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}
int k_block_mlp(const float* temb_silu, const MlpDesc* descs, int n_desc, float* ss, int B, int tdim, int ss_stride, hipStream_t s) {
    block_mlp_kernel<<<dim3(B, n_desc), 256, 0, s>>>(temb_silu, descs, ss, tdim, ss_stride);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}
int k_gn_finalize(const float* partial, int B, int H, int W, int C, const float* gamma, const float* beta, const float* ss, int ss_stride,
                  int ss_offset, float* a_out, float* s_out, hipStream_t s, float* stats_out) {
    OFD_CHECK_ARG(C % 64 == 0, "gn_finalize: C=%d", C);
    const int tiles = cdiv(H, 8) * cdiv(W, 32) * 4;    // one partial per (tile, wave) from the conv epilogue
    gn_finalize_kernel<<<dim3(B, 8), GNF_NT, 0, s>>>(partial, tiles, C, (double)H * W * (C / 8), gamma, beta, ss, ss_stride, ss_offset, a_out, s_out, stats_out);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}
int k_resblock_out(const bf16_t* h, const float* a, const float* sft, const bf16_t* x, bf16_t* out, int B, int H, int W, int C, hipStream_t s) {
    const size_t units = (size_t)B * H * W * (C / 8);
    resblock_out_kernel<<<sgrid(units, 256, 0), 256, 0, s>>>(h, a, sft, x, out, C, (size_t)H * W, units);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}
int k_layernorm_c(const bf16_t* x, const float* g, const bf16_t* res, bf16_t* out, size_t npix, int C, float eps, hipStream_t s) {
    OFD_CHECK_ARG(C == 64 || C == 128 || C == 256 || C == 512, "layernorm_c: C=%d unsupported", C);
    const size_t waves = (npix + (512 / C) - 1) / (512 / C);
    layernorm_c_kernel<<<sgrid(waves * 64, 256, 0), 256, 0, s>>>(x, g, res, out, C, eps, npix);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}
void launch_la_ctx_combine(const float* partial, float* ctx, int B, int nparts, float inv_n, float* ml_out, hipStream_t s) {
    const size_t lds = (size_t)(nparts <= 256 ? nparts : 1) * 32 * sizeof(float);
    la_ctx_combine_kernel<<<dim3(B * 4, 4), 256, lds, s>>>(partial, ctx, nparts, inv_n, ml_out);
}
int k_flash_attention(const bf16_t* qkv, bf16_t* out, int B, int n, hipStream_t s, float* lse) {
    flash_attn_d32_kernel<<<dim3(cdiv(n, 128), B * 4), 256, 0, s>>>(qkv, out, n, 0.17677669529663687f, lse);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}
int k_final_conv(const bf16_t* x, const float* w, const float* bias, float* out, int B, int H, int W, int C, int out_dim, hipStream_t s) {
    OFD_CHECK_ARG(out_dim >= 1 && out_dim <= 4 && C % 8 == 0, "final_conv: out_dim=%d C=%d", out_dim, C);
    OFD_CHECK_ARG(C == 64 || C == 128 || C == 256 || C == 512, "final_conv: C=%d unsupported", C);
    final_conv_kernel<<<sgrid(((size_t)B * H * W + (512 / C) - 1) / (512 / C) * 64), 256, 0, s>>>(x, w, bias, out, C, out_dim, (size_t)H * W, (size_t)B * H * W);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}
int k_nhwc_to_nchw(const bf16_t* x, float* out, int B, int H, int W, int C, hipStream_t s) {
    const size_t total = (size_t)B * H * W * C;
    nhwc_to_nchw_kernel<<<sgrid(total), 256, 0, s>>>(x, out, C, (size_t)H * W, total);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}
int k_nchw_to_nhwc(const float* x, bf16_t* out, int B, int H, int W, int C, hipStream_t s) {
    const size_t total = (size_t)B * H * W * C;
    nchw_to_nhwc_kernel<<<sgrid(total), 256, 0, s>>>(x, out, C, (size_t)H * W, total);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

}  // namespace ofd
