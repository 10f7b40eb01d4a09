// Batched training augmentation in ONE pass (the reference's Augmentor, augmentation.py:6-76, runs torchvision transforms per
// sample on the host): photometric jitter / grayscale / 3x3 Gaussian blur on the two frames, horizontal / vertical flip and a
// random resized crop (bilinear, resized back to (H, W)) on frames and flow, the flow negated along flipped axes and rescaled by the
// crop's zoom.  HBM-bound: every output element is written once; a sample that takes no transform is a copy.
// The per-sample decisions are drawn on the host side of the boundary (augmentation.py of this package) and arrive as a table.
#include "common.h"

namespace ofd {

// params[b][16]: 0 jitter on, 1 brightness, 2 contrast, 3 saturation, 4 grayscale on, 5 blur on, 6 sigma, 7 h-flip, 8 v-flip,
//                9 crop on, 10 oy, 11 ox, 12 ch, 13 cw (window origin / size as fractions of the image)
constexpr int AUG_NP = 16;

__device__ __forceinline__ float aug_gray(float r, float g, float b) { return 0.299f * r + 0.587f * g + 0.114f * b; }

// mean gray level of every (sample, frame): what the contrast factor pivots on.  The workgroups of a (sample, frame) add their shares as
// 64-bit fixed point (x 2^44, integer atomics): the sum does not depend on the order they retire in, the augmented batch is the same
// bit for bit for the same inputs and table (a double atomic made the last bits of the mean, and through the contrast pivot every
// jittered pixel, a matter of scheduling).  |mean| < 5e5, resolution 6e-14.
constexpr double AUG_MEAN_SCALE = 17592186044416.0;      // 2^44
__global__ void __launch_bounds__(256) aug_gray_mean_kernel(const float* __restrict__ img, const float* __restrict__ tgt, long long* __restrict__ means, int plane) {
    const int b = blockIdx.y >> 1, frame = blockIdx.y & 1;
    const float* p = (frame ? tgt : img) + (size_t)b * 3 * plane;
    double s = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < plane; i += gridDim.x * 256) s += (double)aug_gray(p[i], p[plane + i], p[2 * plane + i]);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __shared__ double ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicAdd((unsigned long long*)(means + blockIdx.y), (unsigned long long)__double2ll_rn(((ws[0] + ws[1]) + (ws[2] + ws[3])) / (double)plane * AUG_MEAN_SCALE));
}

__global__ void __launch_bounds__(256) augment_kernel(const float* __restrict__ img, const float* __restrict__ tgt, const float* __restrict__ flow,
                                                      const float* __restrict__ params, const long long* __restrict__ means,
                                                      float* __restrict__ o_img, float* __restrict__ o_tgt, float* __restrict__ o_flow,
                                                      int H, int W, int ref_flip) {
    const int b = blockIdx.y, plane = H * W;
    const float* P = params + (size_t)b * AUG_NP;
    const bool jit = P[0] != 0.0f, gray = P[4] != 0.0f, blur = P[5] != 0.0f, hf = P[7] != 0.0f, vf = P[8] != 0.0f, crop = P[9] != 0.0f;
    const float br = P[1], ct = P[2], st = P[3], sigma = P[6], oy = P[10], ox = P[11], ch = P[12], cw = P[13];
    float k1[3] = {0.0f, 1.0f, 0.0f};
    if (blur) {
        const float e = __expf(-0.5f / (sigma * sigma));
        const float n = 1.0f / (1.0f + 2.0f * e);
        k1[0] = e * n; k1[1] = n; k1[2] = e * n;
    }
    const float mean_g[2] = {(float)((double)means[b * 2] / AUG_MEAN_SCALE) * br, (float)((double)means[b * 2 + 1] / AUG_MEAN_SCALE) * br};    // mean gray after the brightness factor
    const float* src[2] = {img + (size_t)b * 3 * plane, tgt + (size_t)b * 3 * plane};
    const float* fsrc = flow + (size_t)b * 2 * plane;

    auto photo = [&](int frame, int y, int x, float (&rgb)[3]) {       // pointwise part, at integer coordinates of the ORIGINAL image
        const float* p = src[frame] + y * W + x;
        float r = p[0], g = p[plane], bl = p[2 * plane];
        if (jit) {
            r *= br; g *= br; bl *= br;
            const float m = mean_g[frame];
            r = (r - m) * ct + m; g = (g - m) * ct + m; bl = (bl - m) * ct + m;
            const float gy = aug_gray(r, g, bl);
            r = fminf(fmaxf((r - gy) * st + gy, 0.0f), 1.0f);
            g = fminf(fmaxf((g - gy) * st + gy, 0.0f), 1.0f);
            bl = fminf(fmaxf((bl - gy) * st + gy, 0.0f), 1.0f);
        }
        if (gray) r = g = bl = aug_gray(r, g, bl);
        rgb[0] = r; rgb[1] = g; rgb[2] = bl;
    };
    auto refl = [](int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); };
    auto blurred = [&](int frame, int y, int x, float (&rgb)[3]) {
        if (!blur) { photo(frame, y, x, rgb); return; }
        rgb[0] = rgb[1] = rgb[2] = 0.0f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                float t[3];
                photo(frame, refl(y + dy, H), refl(x + dx, W), t);
                const float w = k1[dy + 1] * k1[dx + 1];
                rgb[0] += w * t[0]; rgb[1] += w * t[1]; rgb[2] += w * t[2];
            }
    };

    for (int i = blockIdx.x * 256 + threadIdx.x; i < plane; i += gridDim.x * 256) {
        const int y = i / W, x = i - y * W;
        // sampling position in the (flipped) image: the crop window resized to (H, W), pixel centres (grid_sample align_corners=False,
        // border padding); identity without a crop
        float fy = (float)y, fx = (float)x;
        if (crop) {
            fy = fminf(fmaxf(oy * (float)H + ch * ((float)y + 0.5f) - 0.5f, 0.0f), (float)(H - 1));
            fx = fminf(fmaxf(ox * (float)W + cw * ((float)x + 0.5f) - 0.5f, 0.0f), (float)(W - 1));
        }
        const int y0 = (int)floorf(fy), x0 = (int)floorf(fx), y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
        const float wy = fy - (float)y0, wx = fx - (float)x0;
        const int ys[2] = {vf ? H - 1 - y0 : y0, vf ? H - 1 - y1 : y1}, xs[2] = {hf ? W - 1 - x0 : x0, hf ? W - 1 - x1 : x1};
        const float wts[4] = {(1.0f - wy) * (1.0f - wx), (1.0f - wy) * wx, wy * (1.0f - wx), wy * wx};
        float out_rgb[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}, out_f[2] = {0.f, 0.f};
        const int ncorner = crop ? 4 : 1;
        for (int k = 0; k < ncorner; ++k) {
            const int yy = ys[k >> 1], xx = xs[k & 1];
            const float w = crop ? wts[k] : 1.0f;
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                float t[3];
                blurred(f, yy, xx, t);
                out_rgb[f][0] += w * t[0]; out_rgb[f][1] += w * t[1]; out_rgb[f][2] += w * t[2];
            }
            out_f[0] += w * fsrc[yy * W + xx];
            out_f[1] += w * fsrc[plane + yy * W + xx];
        }
        // flips negate the flow component along the flipped axis (channel 0 = x) and a crop window of width fraction cw zooms the
        // x displacement by 1 / cw.  ref_flip (= reference_semantics): the reference's own choices instead -- the flips negate the
        // OTHER channel (augmentation.py:37-45) and the crop MULTIPLIES channel 0 by the height fraction, channel 1 by the width
        // fraction (augmentation.py:47-48: batch[:, -2:] / image_size * (h, w))
        const int cx = ref_flip ? 1 : 0, cy = ref_flip ? 0 : 1;
        if (hf) out_f[cx] = -out_f[cx];
        if (vf) out_f[cy] = -out_f[cy];
        if (ref_flip) { out_f[0] *= ch; out_f[1] *= cw; }
        else { out_f[0] /= cw; out_f[1] /= ch; }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            o_img[((size_t)b * 3 + c) * plane + i] = out_rgb[0][c];
            o_tgt[((size_t)b * 3 + c) * plane + i] = out_rgb[1][c];
        }
        o_flow[((size_t)b * 2) * plane + i] = out_f[0];
        o_flow[((size_t)b * 2 + 1) * plane + i] = out_f[1];
    }
}

// The (B, 16) decision table from (B, 14) uniform draws, one thread per sample: what Augmentor.draw (augmentation.py of this package) spells
// as ~50 tensor operations -- 50 launches of a few microseconds in front of every training step.
__global__ void __launch_bounds__(64) augment_table_kernel(const float* __restrict__ u, float* __restrict__ P, int B) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float* v = u + (size_t)b * 14;
    float* o = P + (size_t)b * AUG_NP;
    o[0] = v[0] < 0.4f ? 1.0f : 0.0f;
    o[1] = 1.0f + (v[1] - 0.5f) * 0.2f;                      // brightness, contrast, saturation in 1 +- 0.1
    o[2] = 1.0f + (v[2] - 0.5f) * 0.2f;
    o[3] = 1.0f + (v[3] - 0.5f) * 0.2f;
    o[4] = v[4] < 0.1f ? 1.0f : 0.0f;
    o[5] = v[5] < 0.2f ? 1.0f : 0.0f;
    o[6] = fmaxf(v[6] * 0.5f, 0.05f);                        // Gaussian sigma
    o[7] = v[7] < 0.3f ? 1.0f : 0.0f;
    o[8] = v[8] < 0.3f ? 1.0f : 0.0f;
    const bool crop = v[9] < 0.15f;
    const float area = 0.8f + 0.2f * v[10];                  // RandomResizedCrop scale (0.8, 1.0), ratio (0.9, 1.1) log-uniform
    const float ratio = expf((v[11] * 2.0f - 1.0f) * 0.09531018f);
    const float ch = fminf(sqrtf(area / ratio), 1.0f), cw = fminf(sqrtf(area * ratio), 1.0f);
    o[9] = crop ? 1.0f : 0.0f;
    o[10] = crop ? v[12] * (1.0f - ch) : 0.0f;
    o[11] = crop ? v[13] * (1.0f - cw) : 0.0f;
    o[12] = crop ? ch : 1.0f;
    o[13] = crop ? cw : 1.0f;
    o[14] = 0.0f;
    o[15] = 0.0f;
}

// min, max, mean and mean over the elements of the (unbiased) standard deviation ACROSS the batch of x (B, n): the statistics training_step logs
// for cond and flow (flow_diffuser.py:218-235 of the reference: torch.min / max / mean / mean(std(x, dim = 0))) in one pass over x.
// Per-workgroup partials (ws: 4 doubles per workgroup), added in a fixed order by the second kernel.
constexpr int BS_BLOCKS = 1024;
__global__ void __launch_bounds__(256) batch_stats_kernel(const float* __restrict__ x, int B, size_t n, double* __restrict__ ws) {
    float mn = 3.0e38f, mx = -3.0e38f;
    double sum = 0.0, sd = 0.0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        float s1 = 0.0f;
        for (int b = 0; b < B; ++b) {
            const float v = x[(size_t)b * n + e];
            mn = fminf(mn, v); mx = fmaxf(mx, v);
            s1 += v;
        }
        const float m = s1 / (float)B;
        float q = 0.0f;
        for (int b = 0; b < B; ++b) { const float d = x[(size_t)b * n + e] - m; q += d * d; }      // (second read: L2)
        sum += (double)s1;
        sd += (double)sqrtf(B > 1 ? q / (float)(B - 1) : __builtin_nanf(""));
    }
    __shared__ double r0[256], r1[256];
    __shared__ float r2[256], r3[256];
    r0[threadIdx.x] = sum; r1[threadIdx.x] = sd; r2[threadIdx.x] = mn; r3[threadIdx.x] = mx;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) {
            r0[threadIdx.x] += r0[threadIdx.x + k]; r1[threadIdx.x] += r1[threadIdx.x + k];
            r2[threadIdx.x] = fminf(r2[threadIdx.x], r2[threadIdx.x + k]); r3[threadIdx.x] = fmaxf(r3[threadIdx.x], r3[threadIdx.x + k]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double* o = ws + 4 * (size_t)blockIdx.x;
        o[0] = r0[0]; o[1] = r1[0]; o[2] = (double)r2[0]; o[3] = (double)r3[0];
    }
}
__global__ void __launch_bounds__(256) batch_stats_total_kernel(const double* __restrict__ ws, int nblocks, int B, size_t n, float* __restrict__ out) {
    __shared__ double r0[256], r1[256], r2[256], r3[256];
    double sum = 0.0, sd = 0.0, mn = 3.0e38, mx = -3.0e38;
    for (int i = threadIdx.x; i < nblocks; i += 256) {
        sum += ws[4 * i]; sd += ws[4 * i + 1];
        mn = fmin(mn, ws[4 * i + 2]); mx = fmax(mx, ws[4 * i + 3]);
    }
    r0[threadIdx.x] = sum; r1[threadIdx.x] = sd; r2[threadIdx.x] = mn; r3[threadIdx.x] = mx;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) {
            r0[threadIdx.x] += r0[threadIdx.x + k]; r1[threadIdx.x] += r1[threadIdx.x + k];
            r2[threadIdx.x] = fmin(r2[threadIdx.x], r2[threadIdx.x + k]); r3[threadIdx.x] = fmax(r3[threadIdx.x], r3[threadIdx.x + k]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = (float)r2[0];                                // min
        out[1] = (float)r3[0];                                // max
        out[2] = (float)(r0[0] / ((double)B * (double)n));    // mean
        out[3] = (float)(r1[0] / (double)n);                  // mean over elements of the std across the batch
    }
}

}  // namespace ofd
using namespace ofd;

extern "C" int ofd_augment_table(const float* uniforms, float* params, int B, void* stream) {
    OFD_CHECK_ARG(uniforms && params && B > 0, "augment_table: bad argument");
    augment_table_kernel<<<(B + 63) / 64, 64, 0, (hipStream_t)stream>>>(uniforms, params, B);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

extern "C" size_t ofd_batch_stats_ws_doubles(void) { return 4 * (size_t)BS_BLOCKS; }
extern "C" int ofd_batch_stats(const float* x, int B, size_t n_per_sample, double* ws, float* out4, void* stream) {
    OFD_CHECK_ARG(x && ws && out4 && B > 0 && n_per_sample > 0, "batch_stats: bad argument");
    size_t g = (n_per_sample + 255) / 256;
    if (g > BS_BLOCKS) g = BS_BLOCKS;
    batch_stats_kernel<<<(unsigned)g, 256, 0, (hipStream_t)stream>>>(x, B, n_per_sample, ws);
    batch_stats_total_kernel<<<1, 256, 0, (hipStream_t)stream>>>(ws, (int)g, B, n_per_sample, out4);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

extern "C" int ofd_augment(const float* img, const float* tgt, const float* flow, const float* params, void* means_ws, float* out_img,
                           float* out_tgt, float* out_flow, int B, int H, int W, int reference_semantics, void* stream) {
    OFD_CHECK_ARG(img && tgt && flow && params && means_ws && out_img && out_tgt && out_flow, "augment: null pointer");
    OFD_CHECK_ARG(B > 0 && H > 1 && W > 1 && (long)H * W < (1L << 30), "augment: bad shape");
    hipStream_t s = (hipStream_t)stream;
    OFD_HIP(hipMemsetAsync(means_ws, 0, (size_t)B * 2 * sizeof(double), s));
    const int plane = H * W;
    int gx = (plane + 255) / 256;
    aug_gray_mean_kernel<<<dim3(gx < 64 ? gx : 64, B * 2), 256, 0, s>>>(img, tgt, (long long*)means_ws, plane);
    augment_kernel<<<dim3(gx < 512 ? gx : 512, B), 256, 0, s>>>(img, tgt, flow, params, (const long long*)means_ws, out_img, out_tgt, out_flow, H, W,
                                                                  reference_semantics);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}
