// Training-path kernels that are not convolutions (backward of DD:172-214 Block / ResnetBlock,
// DD:116-125 LayerNorm, DD:361 final conv) plus the un-fused forward pieces the training forward
// keeps (the inference path fuses them into conv loaders / epilogues).  NHWC bf16 activations and
// activation gradients, fp32 statistics and parameter gradients.
#include <cstdlib>
#include "blocks.h"
#include "det.h"

namespace ofd {

__device__ __forceinline__ void t_unpack8(const uint4& v, float (&f)[8]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[2 * j] = bf2f((bf16_t)(w[j] & 0xffffu));
        f[2 * j + 1] = bf2f((bf16_t)(w[j] >> 16));
    }
}
__device__ __forceinline__ uint4 t_pack8(const float (&f)[8]) {
    return make_uint4(f2bf2(f[0], f[1]), f2bf2(f[2], f[3]), f2bf2(f[4], f[5]), f2bf2(f[6], f[7]));
}
__device__ __forceinline__ float t_sigmoid(float u) { return __builtin_amdgcn_rcpf(1.0f + __expf(-u)); }
// d silu(u) / du
__device__ __forceinline__ float t_dsilu(float u) {
    const float sg = t_sigmoid(u);
    return sg * (1.0f + u * (1.0f - sg));
}
__device__ __forceinline__ void load8f(const float* p, float (&f)[8]) {
    *(float4*)&f[0] = *(const float4*)p;
    *(float4*)&f[4] = *(const float4*)(p + 4);
}

// ---- forward: out = SiLU(h * a[b,c] + s[b,c])  (DD:181-187 after GroupNorm folded into (a, s)) ----
__global__ void __launch_bounds__(256) affine_silu_kernel(const bf16_t* __restrict__ h, const float* __restrict__ a, const float* __restrict__ s,
                                                          bf16_t* __restrict__ out, int C, size_t pix_per_sample, size_t total_units) {
    const int c8n = C / 8;
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < total_units; u += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = u / c8n;
        const int c = (int)(u % c8n) * 8;
        const size_t b = pix / pix_per_sample;
        float hv[8], av[8], sv[8];
        t_unpack8(*(const uint4*)(h + u * 8), hv);
        load8f(a + b * C + c, av);
        load8f(s + b * C + c, sv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float y = hv[j] * av[j] + sv[j];
            hv[j] = y * t_sigmoid(y);
        }
        *(uint4*)(out + u * 8) = t_pack8(hv);
    }
}

// ---- backward of out = SiLU(GN-affine(h)), pass 1: per (sample, channel) sums of du and du*h --------
// du = g * silu'(a h + s); also sum h (for the bias gradient of the conv that made h).  grid (chunks, B); partial[b][chunk][C][3]
__global__ void __launch_bounds__(256) gnbwd_reduce_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ h, const float* __restrict__ a,
                                                           const float* __restrict__ s, float* __restrict__ partial, int C, size_t pix_per_sample,
                                                           int pix_per_chunk) {
    __shared__ float red[256][25];
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, c8n = C / 8;
    const size_t p0 = (size_t)chunk * pix_per_chunk, p1 = min(pix_per_sample, p0 + pix_per_chunk);
    for (int cu0 = 0; cu0 < c8n; cu0 += 256) {
        const int lanes_c = min(c8n - cu0, 256), rows = 256 / lanes_c, my_c = tid % lanes_c, my_r = tid / lanes_c;
        float a1[8], a2[8], a3[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { a1[j] = 0.0f; a2[j] = 0.0f; a3[j] = 0.0f; }
        if (my_r < rows) {
            float av[8], sv[8];
            load8f(a + (size_t)b * C + (cu0 + my_c) * 8, av);
            load8f(s + (size_t)b * C + (cu0 + my_c) * 8, sv);
            for (size_t p = p0 + my_r; p < p1; p += rows) {
                const size_t off = (((size_t)b * pix_per_sample + p) * c8n + cu0 + my_c) * 8;
                float gv[8], hv[8];
                t_unpack8(*(const uint4*)(g + off), gv);
                t_unpack8(*(const uint4*)(h + off), hv);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float du = gv[j] * t_dsilu(hv[j] * av[j] + sv[j]);
                    a1[j] += du;
                    a2[j] += du * hv[j];
                    a3[j] += hv[j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[tid][j] = a1[j]; red[tid][8 + j] = a2[j]; red[tid][16 + j] = a3[j]; }
        __syncthreads();
        if (tid < lanes_c) {
            float* o = partial + (((size_t)b * gridDim.x + chunk) * C + (cu0 + tid) * 8) * 3;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
                for (int r = 0; r < rows; ++r) { s1 += red[r * lanes_c + tid][j]; s2 += red[r * lanes_c + tid][8 + j]; s3 += red[r * lanes_c + tid][16 + j]; }
                o[3 * j] = s1;
                o[3 * j + 1] = s2;
                o[3 * j + 2] = s3;
            }
        }
        __syncthreads();
    }
}

// pass 2 (tiny): grid (B, 8 groups).  stats[b][g] = {mean, rstd} saved by the forward.
//   dh = a*du + c2*h + c3 ; c2 = -r^2 G2/N ; c3 = -r G1/N + r^2 mu G2/N
//   dgamma += scp*S ; dbeta += scp*A1 ; dscale[b,c] = gamma*S + beta*A1 ; dshift[b,c] = A1     (S = r (A2 - mu A1))
//   dconv_bias[c] += sum_pixels dh = a*A1 + c2*sum(h) + c3*HW      (bias of the conv that produced h; optional)
__global__ void __launch_bounds__(256) gnbwd_finalize_kernel(const float* __restrict__ partial, int chunks, int C, double count, double hw,
                                                             const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ a_aff, const float* __restrict__ ss, int ss_stride, int ss_offset,
                                                             float* __restrict__ c2c3, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             float* __restrict__ dss, float* __restrict__ dconv_bias) {
    __shared__ double g1s[256], g2s[256];
    const int b = blockIdx.x, g = blockIdx.y, tid = threadIdx.x, gs = C / 8;     // gs <= 64: at most one channel per thread
    const float mu = stats[((size_t)b * 8 + g) * 2], r = stats[((size_t)b * 8 + g) * 2 + 1];
    double G1 = 0.0, G2 = 0.0, A1 = 0.0, A3 = 0.0;
    const int c = g * gs + tid;
    // all 256 threads walk the chunks (thread -> channel tid % gs, chunks tid / gs, + 256 / gs, ...): one thread per channel made
    // this a chain of `chunks` dependent loads (20 us per launch, 38 launches per training step)
    __shared__ double ra[3][256];
    {
        const int lanes = 256 / gs, kg = tid / gs, cc = g * gs + tid % gs;
        double p1 = 0.0, p2 = 0.0, p3 = 0.0;
        for (int k = kg; k < chunks; k += lanes) {
            const float* p = partial + (((size_t)b * chunks + k) * C + cc) * 3;
            p1 += (double)p[0];
            p2 += (double)p[1];
            p3 += (double)p[2];
        }
        ra[0][tid] = p1; ra[1][tid] = p2; ra[2][tid] = p3;
    }
    __syncthreads();
    if (tid < gs) {
        double A2 = 0.0;
        for (int j = 0; j < 256 / gs; ++j) {
            A1 += ra[0][j * gs + tid];
            A2 += ra[1][j * gs + tid];
            A3 += ra[2][j * gs + tid];
        }
        const float scp = ss ? ss[(size_t)b * ss_stride + ss_offset + c] + 1.0f : 1.0f;
        const double S = (double)r * (A2 - (double)mu * A1);
        gacc_add(dgamma + c, (float)(scp * S));
        gacc_add(dbeta + c, (float)(scp * A1));
        if (dss) {
            dss[(size_t)b * ss_stride + ss_offset + c] = (float)(gamma[c] * S + beta[c] * A1);
            dss[(size_t)b * ss_stride + ss_offset + C + c] = (float)A1;
        }
        G1 = (double)gamma[c] * scp * A1;
        G2 = (double)gamma[c] * scp * S;
    }
    g1s[tid] = G1;
    g2s[tid] = G2;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (tid < k) { g1s[tid] += g1s[tid + k]; g2s[tid] += g2s[tid + k]; }
        __syncthreads();
    }
    const double rr = (double)r, N = count;
    const double c2 = -rr * rr * g2s[0] / N, c3 = -rr * g1s[0] / N + rr * rr * (double)mu * g2s[0] / N;
    if (tid == 0) {
        c2c3[((size_t)b * 8 + g) * 2] = (float)c2;
        c2c3[((size_t)b * 8 + g) * 2 + 1] = (float)c3;
    }
    if (dconv_bias && tid < gs) gacc_add(dconv_bias + c, (float)((double)a_aff[(size_t)b * C + c] * A1 + c2 * A3 + c3 * hw));
}

// pass 3: dh = a * g * silu'(a h + s) + c2[b,grp] * h + c3[b,grp]
__global__ void __launch_bounds__(256) gnbwd_apply_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ h, const float* __restrict__ a,
                                                          const float* __restrict__ s, const float* __restrict__ c2c3, bf16_t* __restrict__ dh, int C,
                                                          size_t pix_per_sample, size_t total_units) {
    const int c8n = C / 8, gs = C / 8;
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < total_units; u += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = u / c8n;
        const int c = (int)(u % c8n) * 8;
        const size_t b = pix / pix_per_sample;
        float gv[8], hv[8], av[8], sv[8];
        t_unpack8(*(const uint4*)(g + u * 8), gv);
        t_unpack8(*(const uint4*)(h + u * 8), hv);
        load8f(a + b * C + c, av);
        load8f(s + b * C + c, sv);
        const int grp = c / gs;                      // 8 channels never straddle a group (gs >= 8)
        const float c2 = c2c3[(b * 8 + grp) * 2], c3 = c2c3[(b * 8 + grp) * 2 + 1];
#pragma unroll
        for (int j = 0; j < 8; ++j) gv[j] = av[j] * gv[j] * t_dsilu(hv[j] * av[j] + sv[j]) + c2 * hv[j] + c3;
        *(uint4*)(dh + u * 8) = t_pack8(gv);
    }
}

// ---- LayerNorm over channels backward (DD:116-125): y = x^ * g (+ residual); lpp lanes per pixel ----
// dx = r (t - mean(t) - x^ mean(t x^)), t = dy*g ; dg[c] += sum_pixels dy * x^   (per-workgroup partial -> atomics)
__global__ void __launch_bounds__(256) layernorm_c_bwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gw, const bf16_t* __restrict__ dy,
                                                              bf16_t* __restrict__ dx, float* __restrict__ dg, int C, float eps, size_t npix, int accumulate,
                                                              const bf16_t* __restrict__ extra) {
    extern __shared__ float dg_s[];       // [4 waves][C]: one row per wave, added in wave order (no LDS float atomics: their order is the scheduler's)
    const int lpp = C / 8, lane = threadIdx.x & 63, sub = lane % lpp, slot = lane / lpp, ppw = 64 / lpp;
    const size_t wave_global = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    float gv[8], dgl[8];
    load8f(gw + sub * 8, gv);
#pragma unroll
    for (int j = 0; j < 8; ++j) dgl[j] = 0.0f;
    const float inv_c = 1.0f / (float)C;
    for (size_t p0 = wave_global * ppw; p0 < npix; p0 += nwaves * ppw) {
        const size_t p = min(p0 + slot, npix - 1);
        const bool ok = p0 + slot < npix;
        float v[8], d[8];
        t_unpack8(*(const uint4*)(x + p * C + sub * 8), v);
        t_unpack8(*(const uint4*)(dy + p * C + sub * 8), d);
        float s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        for (int o = 1; o < lpp; o <<= 1) s += __shfl_xor(s, o, 64);
        const float mean = s * inv_c;
        float q = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] -= mean; q += v[j] * v[j]; }
        for (int o = 1; o < lpp; o <<= 1) q += __shfl_xor(q, o, 64);
        const float rstd = rsqrtf(q * inv_c + eps);
        float m1 = 0.0f, m2 = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] *= rstd;                              // x^
            if (ok) dgl[j] += d[j] * v[j];
            d[j] *= gv[j];                             // t
            m1 += d[j];
            m2 += d[j] * v[j];
        }
        for (int o = 1; o < lpp; o <<= 1) { m1 += __shfl_xor(m1, o, 64); m2 += __shfl_xor(m2, o, 64); }
        m1 *= inv_c;
        m2 *= inv_c;
        if (ok) {
            float r8[8], e8[8];
            if (accumulate) t_unpack8(*(const uint4*)(dx + p * C + sub * 8), r8);
            if (extra) t_unpack8(*(const uint4*)(extra + p * C + sub * 8), e8);      // e.g. the residual branch's gradient
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = rstd * (d[j] - m1 - v[j] * m2) + (accumulate ? r8[j] : 0.0f) + (extra ? e8[j] : 0.0f);
            *(uint4*)(dx + p * C + sub * 8) = t_pack8(d);
        }
    }
    // lanes sub, sub + lpp, ... of a wave hold partial sums of the same 8 channels: butterfly over the pixel slots, then the wave's row
#pragma unroll
    for (int j = 0; j < 8; ++j)
        for (int o = lpp; o < 64; o <<= 1) dgl[j] += __shfl_xor(dgl[j], o, 64);
    if (slot == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) dg_s[(threadIdx.x >> 6) * C + sub * 8 + j] = dgl[j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256) gacc_add(dg + i, (dg_s[i] + dg_s[C + i]) + (dg_s[2 * C + i] + dg_s[3 * C + i]));
}

// ---- final 1x1 conv backward (DD:361,417): y[o] = sum_c w[o,c] x[c] + b[o], y NCHW fp32, x NHWC bf16 -----
// dx[c] = sum_o w[o,c] dy[o] ; dw[o,c] += sum_p dy[o] x[c] ; db[o] += sum_p dy[o]
__global__ void __launch_bounds__(256) final_conv_bwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dy,
                                                             bf16_t* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db, int C, int out_dim,
                                                             size_t plane, size_t total) {
    extern __shared__ float sm[];          // [4 waves][out_dim*C + 4]: dw row and db of each wave, added in wave order (no LDS float atomics)
    const int rowf = out_dim * C + 4;
    const int lpp = C / 8, lane = threadIdx.x & 63, sub = lane % lpp, slot = lane / lpp, ppw = 64 / lpp;
    float wv[4][8], dwl[4][8], dbl[4] = {0, 0, 0, 0};
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int j = 0; j < 8; ++j) { wv[o][j] = (o < out_dim) ? w[o * C + sub * 8 + j] : 0.0f; dwl[o][j] = 0.0f; }
    const size_t wave_global = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t p0 = wave_global * ppw; p0 < total; p0 += nwaves * ppw) {
        const size_t i = min(p0 + slot, total - 1);
        const bool ok = p0 + slot < total;
        const size_t n = i / plane, pix = i % plane;
        float f[8], dxv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        t_unpack8(*(const uint4*)(x + i * C + sub * 8), f);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            if (o < out_dim) {
                const float d = ok ? dy[(n * out_dim + o) * plane + pix] : 0.0f;
                if (sub == 0) dbl[o] += d;
#pragma unroll
                for (int j = 0; j < 8; ++j) { dxv[j] += wv[o][j] * d; dwl[o][j] += d * f[j]; }
            }
        }
        if (ok) *(uint4*)(dx + i * C + sub * 8) = t_pack8(dxv);
    }
    float* const row = sm + (threadIdx.x >> 6) * rowf;
#pragma unroll
    for (int o = 0; o < 4; ++o)
        if (o < out_dim) {
            // lanes sub, sub + lpp, ... hold partial sums of the same 8 channels (db: the lanes with sub == 0): butterfly over the pixel slots
#pragma unroll
            for (int j = 0; j < 8; ++j)
                for (int k = lpp; k < 64; k <<= 1) dwl[o][j] += __shfl_xor(dwl[o][j], k, 64);
            for (int k = lpp; k < 64; k <<= 1) dbl[o] += __shfl_xor(dbl[o], k, 64);
            if (slot == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) row[o * C + sub * 8 + j] = dwl[o][j];
                if (sub == 0) row[out_dim * C + o] = dbl[o];
            }
        }
    __syncthreads();
    for (int i = threadIdx.x; i < out_dim * C; i += 256) gacc_add(dw + i, (sm[i] + sm[rowf + i]) + (sm[2 * rowf + i] + sm[3 * rowf + i]));
    if (threadIdx.x < out_dim) {
        const int i = out_dim * C + threadIdx.x;
        gacc_add(db + threadIdx.x, (sm[i] + sm[rowf + i]) + (sm[2 * rowf + i] + sm[3 * rowf + i]));
    }
}

// dst (+)= src, bf16 (gradient fan-in of skip connections / residual paths)
__global__ void __launch_bounds__(256) grad_add_kernel(bf16_t* __restrict__ dst, const bf16_t* __restrict__ src, size_t units, int accumulate) {
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < units; u += (size_t)gridDim.x * blockDim.x) {
        float a[8];
        t_unpack8(*(const uint4*)(src + u * 8), a);
        if (accumulate) {
            float b[8];
            t_unpack8(*(const uint4*)(dst + u * 8), b);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += b[j];
        }
        *(uint4*)(dst + u * 8) = t_pack8(a);
    }
}

// ---- time-embedding path backward (DD:319-324 time_mlp, DD:193-196 per-block mlp) ------------------
// dss: [B][ss_stride] gradient of every block's scale|shift row.  One ResnetBlock's Linear, 16 rows per
// workgroup:  dW[j][k] += sum_b dss[b][off+j] * ts[b][k] ; db[j] += sum_b dss[b][off+j] ;
//             dts[b][k] += sum_j W[j][k] dss[b][off+j]   (atomics: every block adds into dts)
__global__ void __launch_bounds__(256) block_mlp_bwd_kernel(const float* __restrict__ dss, const float* __restrict__ ts, const float* __restrict__ weight,
                                                            int n_out, int offset, float* __restrict__ dweight, float* __restrict__ dbias,
                                                            float* __restrict__ dts, int B, int tdim, int ss_stride) {
    // operands first, arithmetic after: the 16 x 16 tile of dss through LDS, a thread's 16 ts values and 16 weight rows in registers
    // (the nested global loads were a 60 us latency chain per launch, 19 launches per step)
    __shared__ float ds[16][17];
    const int tid = threadIdx.x, j0 = blockIdx.x * 16, j1 = min(n_out, j0 + 16), nj = j1 - j0;
    for (int kb = 0; kb < tdim; kb += 256) {         // uniform trip count (the barriers below); threads past tdim compute on zeros
        const int k = kb + tid;
        const bool kv = k < tdim;
        float wv[16], dw[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) { wv[j] = (j < nj && kv) ? weight[(size_t)(j0 + j) * tdim + k] : 0.0f; dw[j] = 0.0f; }
        for (int b0 = 0; b0 < B; b0 += 16) {
            __syncthreads();
            { const int bb = tid >> 4, jj = tid & 15; ds[bb][jj] = (b0 + bb < B && jj < nj) ? dss[(size_t)(b0 + bb) * ss_stride + offset + j0 + jj] : 0.0f; }
            __syncthreads();
            float tsv[16];
#pragma unroll
            for (int b = 0; b < 16; ++b) tsv[b] = (b0 + b < B && kv) ? ts[(size_t)(b0 + b) * tdim + k] : 0.0f;
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                float a = 0.0f;
#pragma unroll
                for (int j = 0; j < 16; ++j) { dw[j] += ds[b][j] * tsv[b]; a += wv[j] * ds[b][j]; }
                if (b0 + b < B && kv) gacc_add(dts + (size_t)(b0 + b) * tdim + k, a);
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < nj && kv) dweight[(size_t)(j0 + j) * tdim + k] += dw[j];
    }
    if (tid < nj) {
        float a = 0.0f;
        for (int b = 0; b < B; ++b) a += dss[(size_t)b * ss_stride + offset + j0 + tid];
        dbias[j0 + tid] += a;
    }
}

// per sample: recompute emb, z1, g1 = GELU(z1), temb; dtemb = dts * silu'(temb); dz1 = (W3^T dtemb) * gelu'(z1)
// scratch[b] = { emb[dim] | g1[tdim] | dtemb[tdim] | dz1[tdim] }
__global__ void __launch_bounds__(256) time_mlp_bwd_sample_kernel(const int64_t* __restrict__ t, const float* __restrict__ w1, const float* __restrict__ b1,
                                                                  const float* __restrict__ w2, const float* __restrict__ temb, const float* __restrict__ dts,
                                                                  float* __restrict__ scratch, int dim) {
    __shared__ float emb[256], z1[1024], dt[1024];
    const int b = blockIdx.x, tid = threadIdx.x, tdim = dim * 4, half_dim = dim / 2;
    float* out = scratch + (size_t)b * (dim + 3 * tdim);
    const float tv = (float)t[b];
    for (int i = tid; i < half_dim; i += 256) {
        const float k = (float)(9.210340371976184 / (double)(half_dim - 1));
        const float f = expf((float)i * -k);
        emb[i] = sinf(tv * f);
        emb[half_dim + i] = cosf(tv * f);
    }
    __syncthreads();
    for (int i = tid; i < dim; i += 256) out[i] = emb[i];
    for (int j = tid; j < tdim; j += 256) {
        float a = b1[j];
        for (int k = 0; k < dim; ++k) a += w1[(size_t)j * dim + k] * emb[k];
        z1[j] = a;
        out[dim + j] = 0.5f * a * (1.0f + erff(a * 0.70710678118654752f));
        const float te = temb[(size_t)b * tdim + j];
        const float sg = 1.0f / (1.0f + expf(-te));
        const float g = dts[(size_t)b * tdim + j] * sg * (1.0f + te * (1.0f - sg));
        dt[j] = g;
        out[dim + tdim + j] = g;
    }
    __syncthreads();
    for (int k = tid; k < tdim; k += 256) {
        float a = 0.0f;
        for (int j = 0; j < tdim; ++j) a += w2[(size_t)j * tdim + k] * dt[j];
        const float z = z1[k];
        const float dgelu = 0.5f * (1.0f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * expf(-0.5f * z * z);
        out[dim + 2 * tdim + k] = a * dgelu;
    }
}

// weight gradients: grid (tdim rows j): dW2[j][k] += sum_b dtemb[b][j] g1[b][k]; dW1[j][k] += sum_b dz1[b][j] emb[b][k]
__global__ void __launch_bounds__(256) time_mlp_bwd_weight_kernel(const float* __restrict__ scratch, float* __restrict__ dw1, float* __restrict__ db1,
                                                                  float* __restrict__ dw2, float* __restrict__ db2, int B, int dim) {
    const int j = blockIdx.x, tid = threadIdx.x, tdim = dim * 4, rec = dim + 3 * tdim;
    for (int k = tid; k < tdim; k += 256) {
        float a = 0.0f;
        for (int b = 0; b < B; ++b) a += scratch[(size_t)b * rec + dim + tdim + j] * scratch[(size_t)b * rec + dim + k];
        dw2[(size_t)j * tdim + k] += a;
    }
    for (int k = tid; k < dim; k += 256) {
        float a = 0.0f;
        for (int b = 0; b < B; ++b) a += scratch[(size_t)b * rec + dim + 2 * tdim + j] * scratch[(size_t)b * rec + k];
        dw1[(size_t)j * dim + k] += a;
    }
    if (tid == 0) {
        float a2 = 0.0f, a1 = 0.0f;
        for (int b = 0; b < B; ++b) { a2 += scratch[(size_t)b * rec + dim + tdim + j]; a1 += scratch[(size_t)b * rec + dim + 2 * tdim + j]; }
        db2[j] += a2;
        db1[j] += a1;
    }
}

static inline int tgrid(size_t total, int cap = 4096) {   // cap = 0: uncapped, see sgrid in blocks.hip
    static const int env_cap = getenv("OFD_GRID_CAP") ? atoi(getenv("OFD_GRID_CAP")) : (1 << 22);
    if (cap <= 0) cap = env_cap;
    size_t b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > (size_t)cap ? cap : b));
}

int k_affine_silu(const bf16_t* h, const float* a, const float* s, bf16_t* out, int B, int H, int W, int C, hipStream_t st) {
    const size_t units = (size_t)B * H * W * (C / 8);
    affine_silu_kernel<<<tgrid(units, 0), 256, 0, st>>>(h, a, s, out, C, (size_t)H * W, units);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

static inline void gn_bwd_chunks(int B, size_t pps, int& chunks, int& ppc) {
    // ~1024 workgroups in the reduction pass whatever the resolution, at least 64 pixels each
    chunks = cdiv(1024, B);
    const int maxc = cdiv((long)pps, 64);
    if (chunks > maxc) chunks = maxc;
    if (chunks < 1) chunks = 1;
    ppc = cdiv((long)pps, chunks);
    chunks = cdiv((long)pps, ppc);
}

size_t gn_bwd_workspace_floats(int B, int H, int W, int C) {
    int chunks, ppc;
    gn_bwd_chunks(B, (size_t)H * W, chunks, ppc);
    return (size_t)B * chunks * C * 3 + (size_t)B * 16;
}

// g: gradient w.r.t. SiLU output; h: the conv output GroupNorm normalised; (a, s): the folded affine;
// stats: [B][8][{mean, rstd}]; ss/dss: per-sample scale|shift rows of this block (NULL for block2);
// dconv_bias (optional): += bias gradient of the convolution that produced h
int k_gn_silu_backward(const bf16_t* g, const bf16_t* h, const float* a, const float* s, const float* stats, const float* gamma,
                       const float* beta, const float* ss, int ss_stride, int ss_offset, bf16_t* dh, float* dgamma, float* dbeta, float* dss,
                       float* workspace, int B, int H, int W, int C, hipStream_t st, float* dconv_bias) {
    OFD_CHECK_ARG(C % 64 == 0 && C <= 512, "gn_silu_backward: C=%d", C);
    const size_t pps = (size_t)H * W;
    int chunks, ppc;
    gn_bwd_chunks(B, pps, chunks, ppc);
    float* partial = workspace;
    float* c2c3 = workspace + (size_t)B * chunks * C * 3;
    gnbwd_reduce_kernel<<<dim3(chunks, B), 256, 0, st>>>(g, h, a, s, partial, C, pps, ppc);
    gnbwd_finalize_kernel<<<dim3(B, 8), 256, 0, st>>>(partial, chunks, C, (double)pps * (C / 8), (double)pps, stats, gamma, beta, a, ss, ss_stride,
                                                    ss_offset, c2c3, dgamma, dbeta, dss, dconv_bias);
    const size_t units = (size_t)B * pps * (C / 8);
    gnbwd_apply_kernel<<<tgrid(units, 0), 256, 0, st>>>(g, h, a, s, c2c3, dh, C, pps, units);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

int k_layernorm_c_bwd(const bf16_t* x, const float* gw, const bf16_t* dy, bf16_t* dx, float* dg, size_t npix, int C, float eps, int accumulate,
                      hipStream_t st, const bf16_t* extra) {
    OFD_CHECK_ARG(C == 64 || C == 128 || C == 256 || C == 512, "layernorm_c_bwd: C=%d", C);
    const size_t waves = (npix + (512 / C) - 1) / (512 / C);
    layernorm_c_bwd_kernel<<<tgrid(waves * 64, 1024), 256, 4 * C * sizeof(float), st>>>(x, gw, dy, dx, dg, C, eps, npix, accumulate, extra);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

int k_final_conv_bwd(const bf16_t* x, const float* w, const float* dy, bf16_t* dx, float* dw, float* db, int B, int H, int W, int C, int out_dim,
                     hipStream_t st) {
    OFD_CHECK_ARG(out_dim >= 1 && out_dim <= 4 && (C == 64 || C == 128), "final_conv_bwd: out_dim=%d C=%d", out_dim, C);
    const size_t total = (size_t)B * H * W;
    final_conv_bwd_kernel<<<tgrid((total + (512 / C) - 1) / (512 / C) * 64, 1024), 256, 4 * (out_dim * C + 4) * sizeof(float), st>>>(
        x, w, dy, dx, dw, db, C, out_dim, (size_t)H * W, total);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

int k_grad_add(bf16_t* dst, const bf16_t* src, size_t elems, int accumulate, hipStream_t st) {
    grad_add_kernel<<<tgrid(elems / 8, 0), 256, 0, st>>>(dst, src, elems / 8, accumulate);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

// dts is an accumulator across ResnetBlocks: zero it before the first block's backward
int k_block_mlp_bwd(const float* dss, const float* temb_silu, const float* weight, int n_out, int offset, float* dweight, float* dbias, float* dts,
                    int B, int tdim, int ss_stride, hipStream_t st) {
    block_mlp_bwd_kernel<<<cdiv(n_out, 16), 256, 0, st>>>(dss, temb_silu, weight, n_out, offset, dweight, dbias, dts, B, tdim, ss_stride);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}
// scratch: B * (dim + 3*4*dim) floats
int k_time_mlp_bwd(const int64_t* t, const float* temb, const float* dts, const float* w1, const float* b1, const float* w2, float* dw1, float* db1,
                   float* dw2, float* db2, float* scratch, int B, int dim, hipStream_t st) {
    OFD_CHECK_ARG(dim * 4 <= 1024 && dim <= 256, "time_mlp_bwd: dim %d too large", dim);
    time_mlp_bwd_sample_kernel<<<B, 256, 0, st>>>(t, w1, b1, w2, temb, dts, scratch, dim);
    time_mlp_bwd_weight_kernel<<<dim * 4, 256, 0, st>>>(scratch, dw1, db1, dw2, db2, B, dim);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

// deterministic mode (det.h): slot += shadow * 2^-38, shadow = 0
__global__ void __launch_bounds__(256) det_flush_kernel(long long* __restrict__ fx, float* __restrict__ f, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const long long v = fx[i];
        if (v != 0) {
            f[i] += (float)((double)v * DET_INV_SCALE);
            fx[i] = 0;
        }
    }
}
int k_det_flush(long long* fx, float* f, size_t n, hipStream_t s) {
    if (n == 0) return OFD_OK;
    size_t b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    det_flush_kernel<<<(unsigned)b, 256, 0, s>>>(fx, f, n);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

OFD_DET_DEFINE_SETTER(det_set_ctx_train_ops)

}  // namespace ofd
using namespace ofd;

extern "C" size_t ofd_gn_bwd_workspace_floats(int B, int H, int W, int C) { return gn_bwd_workspace_floats(B, H, W, C); }
extern "C" int ofd_gn_silu_backward(const void* g, const void* h, const float* a, const float* s, const float* stats, const float* gamma,
                                    const float* beta, const float* ss, int ss_stride, int ss_offset, void* dh, float* dgamma, float* dbeta,
                                    float* dss, float* dconv_bias, float* workspace, int B, int H, int W, int C, void* stream) {
    OFD_CHECK_ARG(g && h && a && s && stats && gamma && beta && dh && dgamma && dbeta && workspace, "gn_silu_backward: null argument");
    return k_gn_silu_backward((const bf16_t*)g, (const bf16_t*)h, a, s, stats, gamma, beta, ss, ss_stride, ss_offset, (bf16_t*)dh, dgamma, dbeta, dss,
                              workspace, B, H, W, C, (hipStream_t)stream, dconv_bias);
}
extern "C" int ofd_affine_silu(const void* h, const float* a, const float* s, void* out, int B, int H, int W, int C, void* stream) {
    OFD_CHECK_ARG(h && a && s && out && C % 8 == 0, "affine_silu: bad argument");
    return k_affine_silu((const bf16_t*)h, a, s, (bf16_t*)out, B, H, W, C, (hipStream_t)stream);
}
extern "C" int ofd_layernorm_c_backward(const void* x, const float* g, const void* dy, void* dx, float* dg, size_t npix, int C, float eps,
                                        int accumulate, void* stream) {
    OFD_CHECK_ARG(x && g && dy && dx && dg, "layernorm_c_backward: null argument");
    return k_layernorm_c_bwd((const bf16_t*)x, g, (const bf16_t*)dy, (bf16_t*)dx, dg, npix, C, eps, accumulate, (hipStream_t)stream, nullptr);
}
extern "C" int ofd_final_conv_backward(const void* x, const float* w, const float* dy, void* dx, float* dw, float* db, int B, int H, int W, int C,
                                       int out_dim, void* stream) {
    OFD_CHECK_ARG(x && w && dy && dx && dw && db, "final_conv_backward: null argument");
    return k_final_conv_bwd((const bf16_t*)x, w, dy, (bf16_t*)dx, dw, db, B, H, W, C, out_dim, (hipStream_t)stream);
}

// forward building blocks of the executor, exported for callers that compose their own blocks (SURVEY 8b export set)
extern "C" int ofd_layernorm_c(const void* x, const float* g, const void* residual, void* out, size_t npix, int C, float eps, void* stream) {
    OFD_CHECK_ARG(x && g && out, "layernorm_c: null argument");
    return k_layernorm_c((const bf16_t*)x, g, (const bf16_t*)residual, (bf16_t*)out, npix, C, eps, (hipStream_t)stream);
}
extern "C" int ofd_time_mlp(const int64_t* t, const float* w1, const float* b1, const float* w2, const float* b2, float* temb, float* temb_silu,
                            int B, int dim, void* stream) {
    OFD_CHECK_ARG(t && w1 && b1 && w2 && b2 && temb && temb_silu && B > 0, "time_mlp: bad argument");
    return k_time_mlp(t, w1, b1, w2, b2, temb, temb_silu, B, dim, (hipStream_t)stream);
}
extern "C" int ofd_gn_finalize(const float* partial, int B, int H, int W, int C, const float* gamma, const float* beta, const float* ss,
                               int ss_stride, int ss_offset, float* a_out, float* s_out, float* stats_out, void* stream) {
    OFD_CHECK_ARG(partial && gamma && beta && a_out && s_out, "gn_finalize: null argument");
    return k_gn_finalize(partial, B, H, W, C, gamma, beta, ss, ss_stride, ss_offset, a_out, s_out, (hipStream_t)stream, stats_out);
}
