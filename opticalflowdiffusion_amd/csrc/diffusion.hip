// Diffusion elementwise steps (denoising_diffusion.py:589-623, 666-698, 750-767, 806-812) and the
// NaN-masked squared-error reduction (warp.py:260-271 + torch.nanmean, DD:908,973).
// HBM-bound streaming kernels: float4 accesses, per-sample scalar coefficients.
#include "common.h"

namespace ofd {

__device__ __forceinline__ float clamp1(float v) { return fminf(fmaxf(v, -1.0f), 1.0f); }

// n4 = n_per_sample / 4 (host checks divisibility, else the scalar tail kernel is used)
template <int VEC>
__global__ void __launch_bounds__(256) q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                                       const float* __restrict__ a, const float* __restrict__ b,
                                                       float* __restrict__ out, size_t n_per_sample) {
    const int s = blockIdx.y;
    const float ca = a[s], cb = b[s];
    const size_t base = (size_t)s * n_per_sample, nv = n_per_sample / VEC;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
        if (VEC == 4) {
            const float4 u = ((const float4*)(x0 + base))[i], v = ((const float4*)(noise + base))[i];
            ((float4*)(out + base))[i] = make_float4(ca * u.x + cb * v.x, ca * u.y + cb * v.y, ca * u.z + cb * v.z, ca * u.w + cb * v.w);
        } else {
            out[base + i] = ca * x0[base + i] + cb * noise[base + i];
        }
    }
}

// one element group of VEC consecutive floats: a single dwordx4 access per operand when VEC == 4 (scalar dword accesses at a 16-byte lane
// stride run these kernels at ~2.3 TB/s; a lane-contiguous float4 form streams)
template <int VEC> struct EwVec { float v[VEC]; };
template <int VEC> __device__ __forceinline__ EwVec<VEC> ew_load(const float* p) {
    EwVec<VEC> r;
    if constexpr (VEC == 4) { const float4 u = *(const float4*)p; r.v[0] = u.x; r.v[1] = u.y; r.v[2] = u.z; r.v[3] = u.w; }
    else r.v[0] = p[0];
    return r;
}
template <int VEC> __device__ __forceinline__ void ew_store(float* p, const EwVec<VEC>& r) {
    if constexpr (VEC == 4) *(float4*)p = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
    else p[0] = r.v[0];
}

template <int VEC>
__global__ void __launch_bounds__(256) ddpm_update_kernel(const float* __restrict__ x_t, const float* __restrict__ mo,
                                                          const float* __restrict__ noise, const float* __restrict__ c1,
                                                          const float* __restrict__ c2, const float* __restrict__ sg,
                                                          float* __restrict__ out, float* __restrict__ x_start, size_t n_per_sample) {
    const int s = blockIdx.y;
    const float k1 = c1[s], k2 = c2[s], ks = (noise && sg) ? sg[s] : 0.0f;
    const size_t base = (size_t)s * n_per_sample, nv = n_per_sample / VEC;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = base + i * VEC;
        const EwVec<VEC> m = ew_load<VEC>(mo + e), xt = ew_load<VEC>(x_t + e);
        EwVec<VEC> nz, r, xs;
        if (ks != 0.0f) nz = ew_load<VEC>(noise + e);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float x0 = clamp1(m.v[j]);                             // DD:670-671
            float v = k1 * x0 + k2 * xt.v[j];                            // DD:615-618
            if (ks != 0.0f) v = v + ks * nz.v[j];                        // DD:688
            r.v[j] = v;
            xs.v[j] = x0;
        }
        ew_store<VEC>(out + e, r);
        if (x_start) ew_store<VEC>(x_start + e, xs);
    }
}

template <int VEC>
__global__ void __launch_bounds__(256) ddim_update_kernel(const float* __restrict__ x_t, const float* __restrict__ mo,
                                                          const float* __restrict__ noise, const float* __restrict__ sr,
                                                          const float* __restrict__ srm1, const float* __restrict__ san,
                                                          const float* __restrict__ cc, const float* __restrict__ sg, int last,
                                                          float* __restrict__ out, float* __restrict__ x_start, size_t n_per_sample) {
    const int s = blockIdx.y;
    const float k_sr = sr[s], k_srm1 = srm1[s];
    const float k_an = last ? 0.0f : san[s], k_c = last ? 0.0f : cc[s], k_s = (last || !noise || !sg) ? 0.0f : sg[s];
    const size_t base = (size_t)s * n_per_sample, nv = n_per_sample / VEC;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = base + i * VEC;
        const EwVec<VEC> m = ew_load<VEC>(mo + e);
        EwVec<VEC> xt, nz, r, xs;
        if (!last) xt = ew_load<VEC>(x_t + e);
        if (k_s != 0.0f) nz = ew_load<VEC>(noise + e);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float x0 = clamp1(m.v[j]);                             // clip_x_start (DD:655)
            float v = x0;
            if (!last) {
                const float eps = (k_sr * xt.v[j] - x0) / k_srm1;        // DD:595-599
                v = x0 * k_an + k_c * eps;                               // DD:765-766
                if (k_s != 0.0f) v = v + k_s * nz.v[j];
            }
            r.v[j] = v;
            xs.v[j] = x0;
        }
        ew_store<VEC>(out + e, r);
        if (x_start) ew_store<VEC>(x_start + e, xs);
    }
}

// per-workgroup (sum, count) -> part[2 * block]; nan_mse_total_kernel adds the workgroups up in a fixed order (no atomics: the loss is
// the same number, bit for bit, for the same inputs)
constexpr int NAN_MSE_BLOCKS = 2048;
__global__ void __launch_bounds__(256) nan_mse_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                      size_t n, double* __restrict__ part) {
    double sum = 0.0, cnt = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float p = pred[i], t = target[i];
        if (!(isnan(p) || isnan(t))) {
            const float d = p - t;
            sum += (double)(d * d);
            cnt += 1.0;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        sum += __shfl_down(sum, o, 64);
        cnt += __shfl_down(cnt, o, 64);
    }
    __shared__ double ssum[4], scnt[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { ssum[wid] = sum; scnt[wid] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
        part[2 * blockIdx.x + 1] = (scnt[0] + scnt[1]) + (scnt[2] + scnt[3]);
    }
}
__global__ void __launch_bounds__(256) nan_mse_total_kernel(double* __restrict__ result, int nblocks) {
    __shared__ double s1[256], s2[256];
    double a = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) { a += result[2 + 2 * i]; c += result[3 + 2 * i]; }
    s1[threadIdx.x] = a; s2[threadIdx.x] = c;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) { s1[threadIdx.x] += s1[threadIdx.x + k]; s2[threadIdx.x] += s2[threadIdx.x + k]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { result[0] = s1[0]; result[1] = s2[0]; }
}

// backward of mean over the non-NaN entries: dpred = 2 (pred - target) * gout / count where both are finite
__global__ void __launch_bounds__(256) nan_mse_grad_kernel(const float* __restrict__ pred, const float* __restrict__ target, size_t n,
                                                           const double* __restrict__ result, const float* __restrict__ gout,
                                                           float* __restrict__ dpred) {
    const float k = (float)(2.0 * (double)gout[0] / result[1]);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float p = pred[i], t = target[i];
        dpred[i] = (isnan(p) || isnan(t)) ? 0.0f : k * (p - t);
    }
}

static inline dim3 ew_grid(int B, size_t nv) {
    size_t b = (nv + 255) / 256;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return dim3((unsigned)b, (unsigned)B);
}

}  // namespace ofd
using namespace ofd;

#define OFD_EW_ARGS_OK(B, n) OFD_CHECK_ARG((B) > 0 && (B) <= 65535 && (n) > 0, "bad B=%d n_per_sample=%zu", (B), (size_t)(n))

extern "C" int ofd_q_sample(const float* x0, const float* noise, const float* sqrt_ac, const float* sqrt_1mac, float* out,
                            int B, size_t n, void* stream) {
    OFD_EW_ARGS_OK(B, n);
    OFD_CHECK_ARG(x0 && noise && sqrt_ac && sqrt_1mac && out, "q_sample: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (n % 4 == 0) q_sample_kernel<4><<<ew_grid(B, n / 4), 256, 0, s>>>(x0, noise, sqrt_ac, sqrt_1mac, out, n);
    else q_sample_kernel<1><<<ew_grid(B, n), 256, 0, s>>>(x0, noise, sqrt_ac, sqrt_1mac, out, n);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

extern "C" int ofd_ddpm_update(const float* x_t, const float* model_out, const float* noise, const float* coef1,
                               const float* coef2, const float* sigma, float* out, float* x_start, int B, size_t n, void* stream) {
    OFD_EW_ARGS_OK(B, n);
    OFD_CHECK_ARG(x_t && model_out && coef1 && coef2 && out, "ddpm_update: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (n % 4 == 0) ddpm_update_kernel<4><<<ew_grid(B, n / 4), 256, 0, s>>>(x_t, model_out, noise, coef1, coef2, sigma, out, x_start, n);
    else ddpm_update_kernel<1><<<ew_grid(B, n), 256, 0, s>>>(x_t, model_out, noise, coef1, coef2, sigma, out, x_start, n);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

extern "C" int ofd_ddim_update(const float* x_t, const float* model_out, const float* noise, const float* sqrt_recip_ac,
                               const float* sqrt_recipm1_ac, const float* sqrt_alpha_next, const float* c, const float* sigma,
                               int last, float* out, float* x_start, int B, size_t n, void* stream) {
    OFD_EW_ARGS_OK(B, n);
    OFD_CHECK_ARG(x_t && model_out && sqrt_recip_ac && sqrt_recipm1_ac && out, "ddim_update: null pointer");
    OFD_CHECK_ARG(last || (sqrt_alpha_next && c), "ddim_update: missing coefficients");
    hipStream_t s = (hipStream_t)stream;
    if (n % 4 == 0)
        ddim_update_kernel<4><<<ew_grid(B, n / 4), 256, 0, s>>>(x_t, model_out, noise, sqrt_recip_ac, sqrt_recipm1_ac, sqrt_alpha_next, c, sigma, last, out, x_start, n);
    else
        ddim_update_kernel<1><<<ew_grid(B, n), 256, 0, s>>>(x_t, model_out, noise, sqrt_recip_ac, sqrt_recipm1_ac, sqrt_alpha_next, c, sigma, last, out, x_start, n);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

extern "C" int ofd_nan_mse_sum(const float* pred, const float* target, size_t n, double* result, void* stream) {
    OFD_CHECK_ARG(pred && target && result && n > 0, "nan_mse_sum: bad argument");
    hipStream_t s = (hipStream_t)stream;
    size_t b = (n + 255) / 256;
    if (b > NAN_MSE_BLOCKS) b = NAN_MSE_BLOCKS;
    nan_mse_kernel<<<(unsigned)b, 256, 0, s>>>(pred, target, n, result + 2);
    nan_mse_total_kernel<<<1, 256, 0, s>>>(result, (int)b);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}
extern "C" size_t ofd_nan_mse_result_doubles(void) { return 2 + 2 * (size_t)NAN_MSE_BLOCKS; }

extern "C" int ofd_nan_mse_grad(const float* pred, const float* target, size_t n, const double* result, const float* gout, float* dpred,
                                void* stream) {
    OFD_CHECK_ARG(pred && target && result && gout && dpred && n > 0, "nan_mse_grad: bad argument");
    size_t b = (n + 255) / 256;
    if (b > 2048) b = 2048;
    nan_mse_grad_kernel<<<(unsigned)b, 256, 0, (hipStream_t)stream>>>(pred, target, n, result, gout, dpred);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}
