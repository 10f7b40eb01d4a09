// Flow-warp kernels for gfx950: forward splat (scatter, restated from the reference's
// softsplat_out/_ingrad/_flowgrad, softsplat_new.py:352-700) and the bilinear grid_sample
// backward warp (warp.py:95-119).  HBM-bound gathers/scatters: no MFMA here.
//
// Forward splat design (instead of the reference's one global atomicAdd per corner):
//   * an output tile of 64x64 pixels x 4 channels is owned by one workgroup and accumulated in
//     LDS; the workgroup scans the source window = tile footprint +- radius, so flow is read
//     once per pixel (not once per channel) and the output is written once, coalesced, with no
//     zero-fill pass and no global atomics;
//   * the LDS accumulators are 64-bit FIXED POINT, not float: ds_add_f32 retires 0.37 lane-atomics
//     per clock and CU on gfx950 whatever the access pattern, ds_add_u64 5-10 (tools/probe/
//     lds_atomic_probe.hip).  Every product in*w is computed in fp32 exactly as the reference does,
//     scaled by 2^(44 - E_c) (E_c = exponent of the largest finite |in| of channel c in the window,
//     found by a first pass over the window) and added as an integer: the sum is exact and
//     order-independent (bit-reproducible, unlike float atomics), one rounding back to fp32 at the
//     end, resolution 2^-44 of the channel maximum.  Non-finite products set NaN / +inf / -inf flag
//     bits per output pixel and follow IEEE addition rules at write-out;
//   * samples whose corner lands in a tile whose window does not contain them ("far" corners,
//     |displacement| > radius) are appended to a list by the workgroup that owns the SOURCE
//     pixel and added with global atomics by a second, normally empty, kernel.
// This file is compiled with -ffp-contract=off: corner indices must be bit-exact.
#include <cstdlib>
#include "common.h"

namespace ofd {

struct SplatGeom {
    int B, C, H, W, Ho, Wo, scale, ox, oy, radius, ntx, nty;
    int tw, th;   // output tile of the general tile kernel (the scale-1 fast kernel and the pyramid kernels use S_TW x S_TH)
    int grid;     // 1: targets are grid_sample's un-normalised coordinates (adjoint of warp_backward_flow), scale 1
    int pyr_L;    // > 0: only the source pixels that are "plain" for every offset of pyramid level pyr_L take part (splat_pyramid)
};

// Pyramid level L (splat_pyramid below): a source pixel is PLAIN when for every offset (a, b) in [0, L)^2 the reference's
// remap takes its ordinary branch on both axes (SS:379-381): L - 1 <= x + flow_x < W - 1 and L - 1 <= y + flow_y < H - 1.
// (x + flow_x - a is exact in fp32 for these magnitudes, so `fltX - fox < 0` is the comparison fltX < a.)
__device__ __forceinline__ bool pyr_plain(float fltX, float fltY, int L, int H, int W) {
    return fltX >= (float)(L - 1) && fltX < (float)W - 1.0f && fltY >= (float)(L - 1) && fltY < (float)H - 1.0f;
}

#ifndef OFD_SPLAT_TH
#define OFD_SPLAT_TH 64
#endif
#ifndef OFD_SPLAT_NT
#define OFD_SPLAT_NT 1024
#endif
constexpr int S_TH = OFD_SPLAT_TH, S_TW = 64, S_CG = 4, S_NT = OFD_SPLAT_NT;
constexpr int SKIPPED = -(1 << 30);

// ---- grid_sample backward warp (WP:95-119): exact op order of the reference expression -------
__device__ __forceinline__ void grid_coords(float flow_c0, float flow_c1, int x, int y, int H, int W, float& ix, float& iy) {
    // flow.flip(1): channel 1 displaces x, channel 0 displaces y (WP:105-106)
    const float gx = (float)x + flow_c1;
    const float gy = (float)y + flow_c0;
    const float vx = 2.0f * gx / (float)max(W - 1, 1) - 1.0f;       // WP:108
    const float vy = 2.0f * gy / (float)max(H - 1, 1) - 1.0f;       // WP:109
    ix = ((vx + 1.0f) / 2.0f) * (float)(W - 1);                       // ATen align_corners un-normalise
    iy = ((vy + 1.0f) / 2.0f) * (float)(H - 1);
}

// a / b with r = RN(1 / b) prepared once: the multiply + two residual corrections of the hardware's own IEEE division sequence
// (v_rcp refinement, scaling and fix-up dropped: b is a small positive integer, r is already correctly rounded).  Bit-identical
// to a / b for finite a in the normal range (brute-forced against IEEE division on 1.1e8 numerators x 14 divisors, and by the
// index-parity tests against ATen); a = +-inf gives NaN instead of inf, which every caller treats alike (non-finite target).
// 5 instructions instead of 11.
__device__ __forceinline__ float div_by_const(float a, float b, float r) {
    float q = a * r;
    q = __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
    q = __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
    return q;
}
// grid_coords with the two divisions done that way (dw = max(W - 1, 1) as float, rw = 1 / dw; same for H)
__device__ __forceinline__ void grid_coords_rcp(float flow_c0, float flow_c1, int x, int y, int H, int W, float dw, float rw, float dh, float rh,
                                                float& ix, float& iy) {
    const float gx = (float)x + flow_c1;
    const float gy = (float)y + flow_c0;
    const float vx = div_by_const(2.0f * gx, dw, rw) - 1.0f;          // WP:108
    const float vy = div_by_const(2.0f * gy, dh, rh) - 1.0f;          // WP:109
    ix = ((vx + 1.0f) / 2.0f) * (float)(W - 1);                       // ATen align_corners un-normalise
    iy = ((vy + 1.0f) / 2.0f) * (float)(H - 1);
}

// variant 0: forward (SS:374-390), 1: ingrad (SS:515-533), 2: flowgrad (SS:628-647).
// Same float/double mix as the reference source: the bare 1.0 literals are double.
template <int VARIANT>
__device__ __forceinline__ bool splat_remap(float flow_x, float flow_y, int x, int y, const SplatGeom& g,
                                            float& fx, float& fy, float& dxx, float& dyy) {
    dxx = 0.0f;
    dyy = 0.0f;
    if (VARIANT == 0 && g.grid) {      // (flow_x, flow_y) are channels (0, 1) of the flow: grid_coords applies the reference's flip
        grid_coords(flow_x, flow_y, x, y, g.H, g.W, fx, fy);
        return fabsf(fx) < 1.0e9f && fabsf(fy) < 1.0e9f;       // the forward kernels' rule: otherwise no corner is in bounds
    }
    float fltX = (float)x + flow_x;
    float fltY = (float)y + flow_y;
    if (!isfinite(fltX) || !isfinite(fltY)) return false;
    if (g.pyr_L > 0 && !pyr_plain(fltX, fltY, g.pyr_L, g.H, g.W)) return false;
    const bool guard = (VARIANT == 0) ? (g.scale > 1) : true;
    const float fW = (float)g.W, fH = (float)g.H, fs = (float)g.scale, fox = (float)g.ox, foy = (float)g.oy;

    if (guard && (double)fltX >= (double)fW - 1.0) {
        const float k = (float)((abs(g.ox - (g.W % g.scale))) % g.scale);
        fltX = (float)((double)fltX + ((double)(fltX - fW) + 1.0) * (double)k);
        if (VARIANT == 1) fltX = (float)((double)fltX + ((double)(fltX - fW) + 1.0) * (double)fox);
        fltX = (fltX - fox) / fs;
    } else if (fltX - fox < 0.0f) {
        fltX = fltX - fox;
    } else {
        fltX = (fltX - fox) / fs;
        dxx = 1.0f / fs;
    }
    if (guard && (double)fltY >= (double)fH - 1.0) {
        const float k = (VARIANT == 2) ? foy : (float)((abs(g.oy - (g.H % g.scale))) % g.scale);
        fltY = (float)((double)fltY + ((double)(fltY - fH) + 1.0) * (double)k);
        fltY = (fltY - foy) / fs;
    } else if (fltY - foy < 0.0f) {
        fltY = fltY - foy;
    } else {
        fltY = (fltY - foy) / fs;
        dyy = 1.0f / fs;
    }
    fx = fltX;
    fy = fltY;
    return true;
}

__device__ __forceinline__ void corner_weights(float fx, float fy, int x0, int y0, float w[4]) {
    const float x1 = (float)(x0 + 1), y1 = (float)(y0 + 1), fx0 = (float)x0, fy0 = (float)y0;
    w[0] = (x1 - fx) * (y1 - fy);     // north-west
    w[1] = (fx - fx0) * (y1 - fy);    // north-east
    w[2] = (x1 - fx) * (fy - fy0);    // south-west
    w[3] = (fx - fx0) * (fy - fy0);   // south-east
}

// clamp before the int conversion so that huge finite targets cannot overflow (they are out of
// every tile either way)
__device__ __forceinline__ int floor_to_int(float v) {
    v = floorf(v);
    v = fminf(fmaxf(v, -1.0e9f), 1.0e9f);
    return (int)v;
}

// source-footprint interval [lo, hi) of output tile t along one axis
__device__ __forceinline__ void footprint(int t, int nt, int tile, int scale, int full, int& lo, int& hi) {
    lo = t * tile * scale;
    hi = (t == nt - 1) ? full : (t + 1) * tile * scale;
}

// largest finite |in| of every (sample, channel) plane, as float bits (non-negative floats order like unsigned ints)
__global__ void __launch_bounds__(256) splat_absmax_kernel(const float* __restrict__ in, unsigned int* __restrict__ absmax, size_t plane) {
    const float* p = in + (size_t)blockIdx.y * plane;
    float m = 0.0f;
    const size_t n4 = plane / 4;
#pragma unroll 4
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = ((const float4*)p)[i];
        const float a0 = fabsf(v.x), a1 = fabsf(v.y), a2 = fabsf(v.z), a3 = fabsf(v.w);
        if (a0 < 3.0e38f) m = fmaxf(m, a0);
        if (a1 < 3.0e38f) m = fmaxf(m, a1);
        if (a2 < 3.0e38f) m = fmaxf(m, a2);
        if (a3 < 3.0e38f) m = fmaxf(m, a3);
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < plane; i += (size_t)gridDim.x * 256) {
        const float a = fabsf(p[i]);
        if (a < 3.0e38f) m = fmaxf(m, a);
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {       // one atomic per workgroup: same-address atomics serialise in L2
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        if (m > 0.0f) atomicMax(absmax + blockIdx.y, __float_as_uint(m));
    }
}

constexpr int S_FIX = 44;                                  // fixed-point fraction bits relative to the channel maximum
constexpr int S_LDS_BYTES = S_CG * S_TH * S_TW * 8 + S_TH * S_TW * 4;

__global__ void __launch_bounds__(S_NT) splat_tile_kernel(const float* __restrict__ in, const float* __restrict__ flow,
                                                          float* __restrict__ out, unsigned long long* __restrict__ far_list,
                                                          unsigned int* __restrict__ far_count, unsigned int far_cap,
                                                          const unsigned int* __restrict__ absmax, SplatGeom g, int c0, int cg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_lds[];
    // output tile of g.th x g.tw pixels (64 x 64 at scale 1; smaller at coarser scales, where a 64 x 64 tile's source footprint is
    // 64 scale pixels wide and the whole level is a handful of tiles): flat [S_CG][th][tw] accumulators, [th][tw] flags
    const int tw = g.tw, th = g.th, tpx = tw * th;
    unsigned long long* acc = (unsigned long long*)s_lds;                     // [S_CG][th][tw]
    unsigned int* flags = (unsigned int*)(s_lds + (size_t)S_CG * tpx * 8);   // 3 bits per channel: nan, +inf, -inf
    __shared__ float k_s[S_CG];
    __shared__ double kinv_s[S_CG];
    const int tid = threadIdx.x;
    const int tx = blockIdx.x, ty = blockIdx.y, n = blockIdx.z;
    const int X0 = tx * tw, Y0 = ty * th;

    for (int i = tid; i < S_CG * tpx; i += S_NT) acc[i] = 0ull;
    for (int i = tid; i < tpx; i += S_NT) flags[i] = 0u;

    int fx0, fx1, fy0, fy1;
    footprint(tx, g.ntx, tw, g.scale, g.W, fx0, fx1);
    footprint(ty, g.nty, th, g.scale, g.H, fy0, fy1);
    const int wx0 = max(0, fx0 - g.radius), wx1 = min(g.W, fx1 + g.radius);
    const int wy0 = max(0, fy0 - g.radius), wy1 = min(g.H, fy1 + g.radius);
    const int ww = wx1 - wx0, wh = wy1 - wy0;
    const size_t plane = (size_t)g.H * g.W;
    const float* flow_n = flow + (size_t)n * 2 * plane;
    const float* in_n = in + ((size_t)n * g.C + c0) * plane;

    if (tid < S_CG) {   // fixed-point scale 2^(S_FIX - E_c) from the largest finite |in| of this (sample, channel) plane
        const float v = (tid < cg) ? __uint_as_float(absmax[(size_t)n * g.C + c0 + tid]) : 0.0f;
        int e = 0;
        if (v > 0.0f) frexpf(v, &e);                                  // v < 2^e
        int sh = S_FIX - e;
        sh = min(max(sh, -100), 126);
        k_s[tid] = ldexpf(1.0f, sh);
        kinv_s[tid] = ldexp(1.0, -sh);
    }
    __syncthreads();
    double kd[S_CG];
#pragma unroll
    for (int c = 0; c < S_CG; ++c) kd[c] = (double)k_s[c];

    // the scan is latency-bound if one pixel is walked at a time (flow -> remap -> image loads -> atomics):
    // S_U pixels per thread are in flight together, all loads issued before the first use
    // thread -> window column tid % 128 (the window is at most 64 + 2*radius <= 128 wide when radius <= 32,
    // wider windows loop over column blocks), rows tid / 128 + 8 i: no integer divisions in the scan
    constexpr int S_U = 4, S_COLS = 128, S_ROWS = S_NT / S_COLS;
    for (int xb = 0; xb < ww; xb += S_COLS)
    for (int r0 = tid / S_COLS; r0 < wh; r0 += S_ROWS * S_U) {
        int xs[S_U], ys[S_U];
        size_t pixs[S_U];
        float f0[S_U], f1[S_U], v[S_U][S_CG];
        bool live[S_U];
#pragma unroll
        for (int u = 0; u < S_U; ++u) {
            const int r = r0 + u * S_ROWS, col = xb + (tid % S_COLS);
            live[u] = r < wh && col < ww;
            ys[u] = wy0 + min(r, wh - 1);
            xs[u] = wx0 + min(col, ww - 1);
            pixs[u] = (size_t)ys[u] * g.W + xs[u];
            f0[u] = flow_n[pixs[u]];
            f1[u] = flow_n[plane + pixs[u]];
#pragma unroll
            for (int c = 0; c < S_CG; ++c) v[u][c] = (c < cg) ? in_n[(size_t)c * plane + pixs[u]] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < S_U; ++u) {
            const int x = xs[u], y = ys[u];
            const size_t pix = pixs[u];
            float fx, fy, d0, d1;
            if (!live[u] || !splat_remap<0>(f0[u], f1[u], x, y, g, fx, fy, d0, d1)) continue;
            const int x0 = floor_to_int(fx), y0 = floor_to_int(fy);
            const bool own = (x >= fx0) && (x < fx1) && (y >= fy0) && (y < fy1);
            const int lx0 = x0 - X0, ly0 = y0 - Y0;
            const bool touches = (lx0 >= -1) && (lx0 < tw) && (ly0 >= -1) && (ly0 < th);
            if (!touches && !own) continue;
            float w[4];
            corner_weights(fx, fy, x0, y0, w);
            if (touches) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int lx = lx0 + (k & 1), ly = ly0 + (k >> 1);
                    const int cx = x0 + (k & 1), cy = y0 + (k >> 1);
                    if (lx >= 0 && lx < tw && ly >= 0 && ly < th && cx < g.Wo && cy < g.Ho) {
#pragma unroll
                        for (int c = 0; c < S_CG; ++c)
                            if (c < cg) {
                                const float val = v[u][c] * w[k];            // the reference's fp32 product (SS:406-418)
                                if (fabsf(val) < 3.0e38f) {
                                    if (val != 0.0f) {
                                        // round(val * 2^sh) as a 64-bit integer without a float->int64 conversion (a ~20
                                        // instruction expansion on gfx950): the 1.5*2^52 trick, exact for |x| < 2^51
                                        const double d = __builtin_fma((double)val, kd[c], 6755399441055744.0);
                                        atomicAdd(&acc[(c * th + ly) * tw + lx], (unsigned long long)(__double_as_longlong(d) - 0x4338000000000000ll));
                                    }
                                } else {
                                    const unsigned bit = (val != val) ? 1u : (val > 0.0f ? 2u : 4u);
                                    atomicOr(&flags[ly * tw + lx], bit << (3 * c));
                                }
                            }
                    }
                }
            }
            // far corners: in-bounds corners whose owner tile does not scan this source pixel.  A target within
            // `radius` (in source pixels) of its source cannot have one: skip the per-corner analysis then.
            const bool maybe_far = fabsf(fx * (float)g.scale - (float)x) >= (float)(g.radius - g.scale - 1) ||
                                   fabsf(fy * (float)g.scale - (float)y) >= (float)(g.radius - g.scale - 1) || !(fx == fx) || !(fy == fy);
            if (own && c0 == 0 && maybe_far) {
                unsigned mask = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int cx = x0 + (k & 1), cy = y0 + (k >> 1);
                    if (cx < 0 || cx >= g.Wo || cy < 0 || cy >= g.Ho) continue;
                    int lo, hi;
                    footprint(cx / tw, g.ntx, tw, g.scale, g.W, lo, hi);
                    bool near = (x >= lo - g.radius) && (x < hi + g.radius);
                    footprint(cy / th, g.nty, th, g.scale, g.H, lo, hi);
                    near = near && (y >= lo - g.radius) && (y < hi + g.radius);
                    if (!near) mask |= 1u << k;
                }
                if (mask) {
                    const unsigned slot = atomicAdd(far_count, 1u);
                    if (slot < far_cap) far_list[slot] = ((unsigned long long)((size_t)n * plane + pix) << 4) | mask;
                }
            }
        }
    }
    __syncthreads();

    const size_t oplane = (size_t)g.Ho * g.Wo;
    float* out_n = out + ((size_t)n * g.C + c0) * oplane;
    for (int i = tid; i < cg * tpx; i += S_NT) {
        const int c = i / tpx, r = (i / tw) % th, col = i % tw;
        const int oy_ = Y0 + r, ox_ = X0 + col;
        if (oy_ < g.Ho && ox_ < g.Wo) {
            float v = (float)((double)(long long)acc[i] * kinv_s[c]);
            const unsigned f = (flags[r * tw + col] >> (3 * c)) & 7u;
            if (f) {       // IEEE: NaN dominates, inf - inf = NaN, otherwise the infinity
                const float inf = __builtin_huge_valf();
                v = ((f & 1u) || (f & 6u) == 6u) ? __builtin_nanf("") : ((f & 2u) ? inf : -inf);
            }
            out_n[(size_t)c * oplane + (size_t)oy_ * g.Wo + ox_] = v;
        }
    }
}

// ---- scale-1 fast path of the forward splat ---------------------------------------------------------------------------------
// scale 1, offset (0, 0), plain splat (neither grid_sample coordinates nor a pyramid mask): the reference's remap is then the
// identity (SS:377-381 with s = 1, ox = oy = 0: `flt - 0 < 0 ? flt - 0 : (flt - 0) / 1`) -- no double arithmetic, no division.
// The general kernel above is VALU-bound (93 M wave-instructions per launch at the benchmark size, profiles/r02_pmc_warp.json):
// every one of its 3.06 window visits per output pixel drags the four image channels along, and the ~1 in 3 visits that touch
// the tile run 16 branchy (corner, channel) bodies in half-empty waves.  Here
//   phase 1 reads only the FLOW of the source window (8 bytes per visit), tests `target touches the tile` and appends the
//           survivors' window coordinates to a list in LDS (one LDS atomic per wave and row: ballot + prefix count);
//   phase 2 walks the list with every lane busy: image channels of survivors only (16 bytes each), corner bounds once per corner,
//           the four channels of a corner as straight-line code.  A product is v * w in double (exact), scaled and added as a
//           64-bit integer exactly as above; it differs from the reference's fp32-rounded product by < 2^-24 of itself, far inside
//           what the order of the reference's float atomics moves.
#ifndef OFD_SPLAT_CAP
#define OFD_SPLAT_CAP 6144
#endif
constexpr int SF_CAP = OFD_SPLAT_CAP;                               // survivor list entries (u16 window coordinates); overflow is handled inline
constexpr int SF_LDS_BYTES = S_LDS_BYTES + SF_CAP * 2 + 16;

__device__ __forceinline__ void sf_accumulate(unsigned long long (*acc)[S_TH][S_TW], unsigned int (*flags)[S_TW], const float (&v)[S_CG],
                                              const double (&kd)[S_CG], float fx, float fy, int x0, int y0, int X0, int Y0, int Wo, int Ho, int cg) {
    float w[4];
    corner_weights(fx, fy, x0, y0, w);
    bool bad = false;
#pragma unroll
    for (int c = 0; c < S_CG; ++c) bad = bad || !(fabsf(v[c]) < 3.0e38f);
#pragma unroll
    for (int k = 0; k < 4; ++k) bad = bad || !(fabsf(w[k]) < 3.0e38f);
    double vd[S_CG];
#pragma unroll
    for (int c = 0; c < S_CG; ++c) vd[c] = (double)v[c] * kd[c];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int lx = x0 - X0 + (k & 1), ly = y0 - Y0 + (k >> 1);
        const int cx = x0 + (k & 1), cy = y0 + (k >> 1);
        if (!(lx >= 0 && lx < S_TW && ly >= 0 && ly < S_TH && cx < Wo && cy < Ho)) continue;
        if (!bad) {
            const double wd = (double)w[k];
#pragma unroll
            for (int c = 0; c < S_CG; ++c)
                if (c < cg) {
                    const double d = __builtin_fma(vd[c], wd, 6755399441055744.0);          // round(v w 2^sh): the 1.5 * 2^52 trick
                    atomicAdd(&acc[c][ly][lx], (unsigned long long)(__double_as_longlong(d) - 0x4338000000000000ll));
                }
        } else {                                            // a non-finite input or weight: per product, with the IEEE flags (rare)
#pragma unroll
            for (int c = 0; c < S_CG; ++c)
                if (c < cg) {
                    const float val = v[c] * w[k];
                    if (fabsf(val) < 3.0e38f) {
                        const double d = __builtin_fma((double)val, kd[c], 6755399441055744.0);
                        atomicAdd(&acc[c][ly][lx], (unsigned long long)(__double_as_longlong(d) - 0x4338000000000000ll));
                    } else {
                        const unsigned bit = (val != val) ? 1u : (val > 0.0f ? 2u : 4u);
                        atomicOr(&flags[ly][lx], bit << (3 * c));
                    }
                }
        }
    }
}

__global__ void __launch_bounds__(S_NT) splat_tile_fast_kernel(const float* __restrict__ in, const float* __restrict__ flow,
                                                               float* __restrict__ out, unsigned long long* __restrict__ far_list,
                                                               unsigned int* __restrict__ far_count, unsigned int far_cap,
                                                               const unsigned int* __restrict__ absmax, SplatGeom g, int c0, int cg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_lds[];
    unsigned long long(*acc)[S_TH][S_TW] = (unsigned long long(*)[S_TH][S_TW])s_lds;          // [S_CG][S_TH][S_TW]
    unsigned int(*flags)[S_TW] = (unsigned int(*)[S_TW])(s_lds + S_CG * S_TH * S_TW * 8);
    unsigned short* list = (unsigned short*)(s_lds + S_LDS_BYTES);
    unsigned int* list_n = (unsigned int*)(s_lds + S_LDS_BYTES + SF_CAP * 2);
    __shared__ float k_s[S_CG];
    __shared__ double kinv_s[S_CG];
    const int tid = threadIdx.x, lane = tid & 63;
    // XCD-aware tile order: workgroups whose ids are equal mod 8 share an XCD (L2) and get a contiguous run of tiles, so the
    // window overlap of neighbouring tiles (3.06 visits per pixel) is served by one L2 instead of eight
    const int ntile = g.ntx * g.nty * g.B;
    int t = blockIdx.x;
    if (ntile >= 8) {
        const int q = ntile / 8, r = ntile % 8, xcd = t % 8, idx = t / 8;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tx = t % g.ntx, ty = (t / g.ntx) % g.nty, n = t / (g.ntx * g.nty);
    const int X0 = tx * S_TW, Y0 = ty * S_TH;

    for (int i = tid; i < S_CG * S_TH * S_TW; i += S_NT) (&acc[0][0][0])[i] = 0ull;
    for (int i = tid; i < S_TH * S_TW; i += S_NT) (&flags[0][0])[i] = 0u;
    if (tid == 0) *list_n = 0u;

    const int fx0 = X0, fx1 = (tx == g.ntx - 1) ? g.W : X0 + S_TW, fy0 = Y0, fy1 = (ty == g.nty - 1) ? g.H : Y0 + S_TH;
    const int wx0 = max(0, fx0 - g.radius), wx1 = min(g.W, fx1 + g.radius);
    const int wy0 = max(0, fy0 - g.radius), wy1 = min(g.H, fy1 + g.radius);
    const int ww = wx1 - wx0, wh = wy1 - wy0;            // <= 128 each (checked by the host)
    const int plane = g.H * g.W;
    const float* flow_n = flow + (size_t)n * 2 * plane;
    const float* in_n = in + ((size_t)n * g.C + c0) * plane;

    if (tid < S_CG) {
        const float v = (tid < cg) ? __uint_as_float(absmax[(size_t)n * g.C + c0 + tid]) : 0.0f;
        int e = 0;
        if (v > 0.0f) frexpf(v, &e);
        int sh = S_FIX - e;
        sh = min(max(sh, -100), 126);
        k_s[tid] = ldexpf(1.0f, sh);
        kinv_s[tid] = ldexp(1.0, -sh);
    }
    __syncthreads();
    double kd[S_CG];
#pragma unroll
    for (int c = 0; c < S_CG; ++c) kd[c] = (double)k_s[c];

    // ---- phase 1: flow of the window -> survivors.  Thread -> window column tid % 128, rows tid / 128 + 8 i
    constexpr int S_U = 7, S_COLS = 128, S_ROWS = S_NT / S_COLS;
    const int col = tid % S_COLS;
    const int xw = wx0 + min(col, ww - 1);
    const float xlo = (float)(X0 - 1), xhi = (float)(X0 + S_TW), ylo = (float)(Y0 - 1), yhi = (float)(Y0 + S_TH), far_thr = (float)(g.radius - 2);
    for (int r0 = tid / S_COLS; r0 < wh; r0 += S_ROWS * S_U) {
        float f0[S_U], f1[S_U];
#pragma unroll
        for (int u = 0; u < S_U; ++u) {
            const int pix = (wy0 + min(r0 + u * S_ROWS, wh - 1)) * g.W + xw;
            f0[u] = flow_n[pix];
            f1[u] = flow_n[plane + pix];
        }
#pragma unroll
        for (int u = 0; u < S_U; ++u) {
            const int r = r0 + u * S_ROWS, y = wy0 + min(r, wh - 1), x = xw;
            const float fx = (float)x + f0[u], fy = (float)y + f1[u];
            // "the target touches the tile" as four float compares: for an integer bound b, floor(f) >= b <=> f >= b and floor(f) < b
            // <=> f < b; NaN and +-inf fail one of them (r03: was isfinite x 2, floor / clamp / int conversion x 2, four int compares)
            const bool touches = r < wh && col < ww && fx >= xlo && fx < xhi && fy >= ylo && fy < yhi;
            // append (r, col) to the list: one LDS atomic per wave
            const unsigned long long m = __ballot(touches);
            unsigned base = 0;
            if (m) {
                if (lane == 0) base = atomicAdd(list_n, (unsigned)__popcll(m));
                base = __shfl(base, 0, 64);
            }
            const unsigned slot = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            if (touches) {
                if (slot < (unsigned)SF_CAP) {
                    list[slot] = (unsigned short)((r << 7) | col);
                } else {                                    // list full (flows converging on this tile): done in place
                    float v[S_CG];
                    const int pix = y * g.W + x;
#pragma unroll
                    for (int c = 0; c < S_CG; ++c) v[c] = (c < cg) ? in_n[(size_t)c * plane + pix] : 0.0f;
                    sf_accumulate(acc, flags, v, kd, fx, fy, floor_to_int(fx), floor_to_int(fy), X0, Y0, g.Wo, g.Ho, cg);
                }
            }
            // far corners (|displacement| > radius): as in the general kernel, by the workgroup that owns the source pixel -- behind
            // a prefilter on the raw flow (a corner is at most 1 px beyond the target: the 2 px margin covers it and the rounding of x + flow)
            if (c0 == 0 && (fabsf(f0[u]) >= far_thr || fabsf(f1[u]) >= far_thr) && r < wh && col < ww && isfinite(fx) && isfinite(fy) &&
                (x >= fx0) && (x < fx1) && (y >= fy0) && (y < fy1)) {
                const int x0 = floor_to_int(fx), y0 = floor_to_int(fy);
                unsigned mask = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int cx = x0 + (k & 1), cy = y0 + (k >> 1);
                    if (cx < 0 || cx >= g.Wo || cy < 0 || cy >= g.Ho) continue;
                    int lo, hi;
                    footprint(cx / S_TW, g.ntx, S_TW, 1, g.W, lo, hi);
                    bool near = (x >= lo - g.radius) && (x < hi + g.radius);
                    footprint(cy / S_TH, g.nty, S_TH, 1, g.H, lo, hi);
                    near = near && (y >= lo - g.radius) && (y < hi + g.radius);
                    if (!near) mask |= 1u << k;
                }
                if (mask) {
                    const unsigned s_ = atomicAdd(far_count, 1u);
                    if (s_ < far_cap) far_list[s_] = ((unsigned long long)((size_t)n * plane + (size_t)(y * g.W + x)) << 4) | mask;
                }
            }
        }
    }
    __syncthreads();

    // ---- phase 2: the survivors, every lane busy; S_V entries per thread in flight
    const int count = (int)min(*list_n, (unsigned)SF_CAP);
    constexpr int S_V = 3;
    for (int i0 = tid; i0 < count; i0 += S_NT * S_V) {
        int px[S_V];
        float f0[S_V], f1[S_V], v[S_V][S_CG];
        int xs[S_V], ys[S_V];
        bool live[S_V];
#pragma unroll
        for (int u = 0; u < S_V; ++u) {
            const int i = i0 + u * S_NT;
            live[u] = i < count;
            const unsigned e = list[min(i, count - 1)];
            ys[u] = wy0 + (int)(e >> 7);
            xs[u] = wx0 + (int)(e & 127u);
            px[u] = ys[u] * g.W + xs[u];
            f0[u] = flow_n[px[u]];
            f1[u] = flow_n[plane + px[u]];
#pragma unroll
            for (int c = 0; c < S_CG; ++c) v[u][c] = (c < cg) ? in_n[(size_t)c * plane + px[u]] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < S_V; ++u) {
            if (!live[u]) continue;
            const float fx = (float)xs[u] + f0[u], fy = (float)ys[u] + f1[u];
            sf_accumulate(acc, flags, v[u], kd, fx, fy, floor_to_int(fx), floor_to_int(fy), X0, Y0, g.Wo, g.Ho, cg);
        }
    }
    __syncthreads();

    const size_t oplane = (size_t)g.Ho * g.Wo;
    float* out_n = out + ((size_t)n * g.C + c0) * oplane;
    for (int i = tid; i < cg * S_TH * S_TW; i += S_NT) {
        const int c = i / (S_TH * S_TW), r = (i / S_TW) % S_TH, cl = i % S_TW;
        const int oy_ = Y0 + r, ox_ = X0 + cl;
        if (oy_ < g.Ho && ox_ < g.Wo) {
            float v = (float)((double)(long long)acc[c][r][cl] * kinv_s[c]);
            const unsigned f = (flags[r][cl] >> (3 * c)) & 7u;
            if (f) {
                const float inf = __builtin_huge_valf();
                v = ((f & 1u) || (f & 6u) == 6u) ? __builtin_nanf("") : ((f & 2u) ? inf : -inf);
            }
            out_n[(size_t)c * oplane + (size_t)oy_ * g.Wo + ox_] = v;
        }
    }
}

__global__ void __launch_bounds__(256) splat_far_kernel(const float* __restrict__ in, const float* __restrict__ flow,
                                                        float* __restrict__ out, const unsigned long long* __restrict__ far_list,
                                                        const unsigned int* __restrict__ far_count, unsigned int far_cap, SplatGeom g) {
    const unsigned count = min(*far_count, far_cap);
    const size_t plane = (size_t)g.H * g.W, oplane = (size_t)g.Ho * g.Wo;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const unsigned long long e = far_list[i];
        const unsigned mask = (unsigned)(e & 15ull);
        const size_t lin = (size_t)(e >> 4);
        const int n = (int)(lin / plane);
        const size_t pix = lin % plane;
        const int y = (int)(pix / g.W), x = (int)(pix % g.W);
        float fx, fy, d0, d1;
        if (!splat_remap<0>(flow[(size_t)n * 2 * plane + pix], flow[(size_t)n * 2 * plane + plane + pix], x, y, g, fx, fy, d0, d1))
            continue;
        const int x0 = floor_to_int(fx), y0 = floor_to_int(fy);
        float w[4];
        corner_weights(fx, fy, x0, y0, w);
        for (int c = 0; c < g.C; ++c) {
            const float v = in[((size_t)n * g.C + c) * plane + pix];
            for (int k = 0; k < 4; ++k)
                if (mask & (1u << k))
                    atomicAdd(&out[((size_t)n * g.C + c) * oplane + (size_t)(y0 + (k >> 1)) * g.Wo + (x0 + (k & 1))], v * w[k]);
        }
    }
}

__global__ void splat_corners_kernel(const float* __restrict__ flow, int32_t* __restrict__ corners, SplatGeom g) {
    const size_t plane = (size_t)g.H * g.W, total = (size_t)g.B * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / plane);
        const size_t pix = i % plane;
        const int y = (int)(pix / g.W), x = (int)(pix % g.W);
        float fx, fy, d0, d1;
        const bool ok = splat_remap<0>(flow[(size_t)n * 2 * plane + pix], flow[(size_t)n * 2 * plane + plane + pix], x, y, g, fx, fy, d0, d1);
        corners[2 * i] = ok ? floor_to_int(fx) : SKIPPED;
        corners[2 * i + 1] = ok ? floor_to_int(fy) : SKIPPED;
    }
}

// gather: one thread per source pixel, all channels (flow read once)
__global__ void __launch_bounds__(256) splat_ingrad_kernel(const float* __restrict__ flow, const float* __restrict__ outgrad,
                                                           float* __restrict__ ingrad, SplatGeom g) {
    const size_t plane = (size_t)g.H * g.W, oplane = (size_t)g.Ho * g.Wo, total = (size_t)g.B * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / plane);
        const size_t pix = i % plane;
        const int y = (int)(pix / g.W), x = (int)(pix % g.W);
        float fx = 0.0f, fy = 0.0f, d0, d1;
        const bool ok = splat_remap<1>(flow[(size_t)n * 2 * plane + pix], flow[(size_t)n * 2 * plane + plane + pix], x, y, g, fx, fy, d0, d1);
        const int x0 = ok ? floor_to_int(fx) : 0, y0 = ok ? floor_to_int(fy) : 0;
        float w[4];
        corner_weights(fx, fy, x0, y0, w);
        for (int c = 0; c < g.C; ++c) {
            float acc = 0.0f;
            if (ok) {
                const float* gp = outgrad + ((size_t)n * g.C + c) * oplane;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int cx = x0 + (k & 1), cy = y0 + (k >> 1);
                    if (cx >= 0 && cx < g.Wo && cy >= 0 && cy < g.Ho) acc += gp[(size_t)cy * g.Wo + cx] * w[k];
                }
            }
            ingrad[((size_t)n * g.C + c) * plane + pix] = acc;   // skipped samples keep 0 (SS:468-474)
        }
    }
}

__global__ void __launch_bounds__(256) splat_flowgrad_kernel(const float* __restrict__ in, const float* __restrict__ flow,
                                                             const float* __restrict__ outgrad, float* __restrict__ flowgrad, SplatGeom g) {
    const size_t plane = (size_t)g.H * g.W, oplane = (size_t)g.Ho * g.Wo, total = (size_t)g.B * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / plane);
        const size_t pix = i % plane;
        const int y = (int)(pix / g.W), x = (int)(pix % g.W);
        float fx, fy, dxx, dyy;
        const bool ok = splat_remap<2>(flow[(size_t)n * 2 * plane + pix], flow[(size_t)n * 2 * plane + plane + pix], x, y, g, fx, fy, dxx, dyy);
        float gx = 0.0f, gy = 0.0f;
        if (ok) {
            const int x0 = floor_to_int(fx), y0 = floor_to_int(fy);
            const float x1 = (float)(x0 + 1), y1 = (float)(y0 + 1);
            // SS:661-675: channel 0 uses the y-weights and dfltYY, channel 1 the x-weights and dfltXX
            const float wx[4] = {-1.0f * (y1 - fy), +1.0f * (y1 - fy), -1.0f * (fy - (float)y0), +1.0f * (fy - (float)y0)};
            const float wy[4] = {(x1 - fx) * -1.0f, (fx - (float)x0) * -1.0f, (x1 - fx) * +1.0f, (fx - (float)x0) * +1.0f};
            for (int c = 0; c < g.C; ++c) {
                const float v = in[((size_t)n * g.C + c) * plane + pix];
                const float* gp = outgrad + ((size_t)n * g.C + c) * oplane;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int cx = x0 + (k & 1), cy = y0 + (k >> 1);
                    if (cx >= 0 && cx < g.Wo && cy >= 0 && cy < g.Ho) {
                        const float go = gp[(size_t)cy * g.Wo + cx];
                        gx += go * v * wx[k] * dyy;
                        gy += go * v * wy[k] * dxx;
                    }
                }
            }
        }
        flowgrad[(size_t)n * 2 * plane + pix] = gx;
        flowgrad[(size_t)n * 2 * plane + plane + pix] = gy;
    }
}

__global__ void __launch_bounds__(256) warp_prep_kernel(const float* __restrict__ first, float* __restrict__ ten_in,
                                                        int B, int C, size_t plane, int square) {
    const size_t total = (size_t)B * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / plane, pix = i % plane;
        bool any_nan = false;
        for (int c = 0; c < C; ++c) any_nan |= isnan(first[(n * C + c) * plane + pix]);
        const float w = any_nan ? 0.0f : 1.0f;
        for (int c = 0; c < C; ++c) {
            float v = first[(n * C + c) * plane + pix];
            v = isnan(v) ? 0.0f : v;
            if (square) v = v * v;
            ten_in[(n * (C + 1) + c) * plane + pix] = v * w;
        }
        ten_in[(n * (C + 1) + C) * plane + pix] = w;
    }
}

__global__ void __launch_bounds__(256) warp_holes_kernel(const float* __restrict__ splat, float* __restrict__ img,
                                                         int B, int C, size_t plane, int mode, int set_nans) {
    const size_t total = (size_t)B * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / plane, pix = i % plane;
        const float w = splat[(n * (C + 1) + C) * plane + pix];
        for (int c = 0; c < C; ++c) {
            float v = splat[(n * (C + 1) + c) * plane + pix];
            if (mode == 1) v = v / (w + 0.0000001f);
            if (set_nans && !(w > 0.0f)) v = __uint_as_float(0x7fc00000u);
            img[(n * C + c) * plane + pix] = v;
        }
    }
}

// One thread = 4 consecutive output pixels: flow and outputs move as 16-byte accesses, and the two
// corners of a row are ONE 8-byte (4-byte aligned) gather, so 6 gathers per pixel instead of 12.
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));

template <int CT>   // CT > 0: channel count known at compile time (all gathers of a thread issue back to back)
__global__ void __launch_bounds__(256) grid_warp_kernel(const float* __restrict__ second, const float* __restrict__ flow,
                                                        float* __restrict__ out, float* __restrict__ mask, int B, int C_rt, int H, int W) {
    const int C = CT > 0 ? CT : C_rt;
    const size_t plane = (size_t)H * W;
    const int wq = W / 4;                                   // host guarantees W % 4 == 0 and W >= 2 for this kernel
    const size_t total = (size_t)B * H * wq;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / ((size_t)H * wq);
        const size_t rem = i % ((size_t)H * wq);
        const int y = (int)(rem / wq), x4 = (int)(rem % wq) * 4;
        const size_t pix = (size_t)y * W + x4;
        const float4 f0 = *(const float4*)(flow + n * 2 * plane + pix);
        const float4 f1 = *(const float4*)(flow + n * 2 * plane + plane + pix);
        const float fl0[4] = {f0.x, f0.y, f0.z, f0.w}, fl1[4] = {f1.x, f1.y, f1.z, f1.w};
        float w[4][4], m[4];
        unsigned inb[4];
        int xc[4], yr0[4], yr1[4];
        bool sel[4][2];        // value of corner column x0 / x0+1 comes from .y of the pair (else .x)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float ix, iy;
            grid_coords(fl0[j], fl1[j], x4 + j, y, H, W, ix, iy);
            const float fx0 = floorf(ix), fy0 = floorf(iy);
            const bool finite = fabsf(ix) < 1.0e9f && fabsf(iy) < 1.0e9f;
            const int x0 = finite ? (int)fx0 : -10, y0 = finite ? (int)fy0 : -10;
            const float wx0 = fx0 + 1.0f - ix, wx1 = ix - fx0, wy0 = fy0 + 1.0f - iy, wy1 = iy - fy0;
            const bool bx0 = x0 >= 0 && x0 < W, bx1 = x0 + 1 >= 0 && x0 + 1 < W, by0 = y0 >= 0 && y0 < H, by1 = y0 + 1 >= 0 && y0 + 1 < H;
            w[j][0] = wx0 * wy0;
            w[j][1] = wx1 * wy0;
            w[j][2] = wx0 * wy1;
            w[j][3] = wx1 * wy1;
            inb[j] = (bx0 && by0 ? 1u : 0u) | (bx1 && by0 ? 2u : 0u) | (bx0 && by1 ? 4u : 0u) | (bx1 && by1 ? 8u : 0u);
            float ms = 0.0f;                                   // sum of in-bounds weights = grid_sample(ones)
            if (bx0 && by0) ms += wx0 * wy0;
            if (bx1 && by0) ms += wx1 * wy0;
            if (bx0 && by1) ms += wx0 * wy1;
            if (bx1 && by1) ms += wx1 * wy1;
            if (ms < 0.999f) ms = 0.0f;                        // WP:116-117
            if (ms > 0.0f) ms = 1.0f;
            m[j] = ms;
            xc[j] = min(max(x0, 0), W - 2);                    // pair (xc, xc+1) always in bounds
            sel[j][0] = (x0 != xc[j]);                         // x0 == xc+1  (x0 == W-1)
            sel[j][1] = (x0 + 1 != xc[j]);                     // x0+1 == xc+1 (normal case); x0+1 == xc when x0 == -1
            yr0[j] = min(max(y0, 0), H - 1);
            yr1[j] = min(max(y0 + 1, 0), H - 1);
        }
#pragma unroll CT > 0 ? CT : 1      // (the run-time channel count instantiation keeps its loop)
        for (int c = 0; c < C; ++c) {
            const float* sp = second + (n * C + c) * plane;
            f32x2u top[4], bot[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                top[j] = *(const f32x2u*)(sp + (size_t)yr0[j] * W + xc[j]);
                bot[j] = *(const f32x2u*)(sp + (size_t)yr1[j] * W + xc[j]);
            }
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // only in-bounds corners contribute (zeros padding); ATen's nw, ne, sw, se order
                float acc = 0.0f;
                const float t0 = sel[j][0] ? top[j].y : top[j].x, t1 = sel[j][1] ? top[j].y : top[j].x;
                const float b0 = sel[j][0] ? bot[j].y : bot[j].x, b1 = sel[j][1] ? bot[j].y : bot[j].x;
                if (inb[j] & 1u) acc += t0 * w[j][0];
                if (inb[j] & 2u) acc += t1 * w[j][1];
                if (inb[j] & 4u) acc += b0 * w[j][2];
                if (inb[j] & 8u) acc += b1 * w[j][3];
                o[j] = acc;
            }
            *(float4*)(out + (n * C + c) * plane + pix) = make_float4(o[0], o[1], o[2], o[3]);
            if (mask) *(float4*)(mask + (n * C + c) * plane + pix) = make_float4(m[0], m[1], m[2], m[3]);
        }
    }
}

// Tiled variant: the vectorised kernel above is gather-bound as soon as the flow is rough -- with |flow| up to
// 20 px and a correlation length of ~9 px every lane of a wave lands on a different source row, so one gather
// instruction touches 64 cache lines (measured 205 us vs 63 us for zero flow at B=16, 440x1024).  Here a
// persistent workgroup (1024 threads, one per CU) owns 64 x 64 output tiles and brings the source window
// (x: tile -24..+23, y: tile +-21, up to three channels: 3 x 47 KB) into LDS with the DIRECT global->LDS path
// (global_load_lds_dwordx4: no staging registers, all of a tile's loads in flight at once); the four corners
// are then paired LDS reads.  While tile t is gathered and written out, the flow of tile t+1 is already in
// registers; its window loads are issued the moment the LDS buffer is free and fly during its coordinate
// math.  Window positions outside the image are never read (the in-bounds bits gate every corner); corners
// outside the window (|flow| > 21) fall back to global loads.  Same arithmetic as the kernels above:
// bit-identical results.  80 us = 3.95 TB/s of algorithmic traffic at B=16, 440x1024 (was 205 us); an
// ablation shows the phases still mostly add up (coordinates 27, window loads 25, gathers 20, stores 17 us):
// a (tile, channel)-pipelined double-buffer variant was slower (95 us: barriers per channel).
constexpr int GT_W = 64, GT_H = 64, GT_RX = 24, GT_RY = 21, GT_THREADS = GT_H * 16, GT_WAVES = GT_THREADS / 64;
constexpr int GT_WW = GT_W + 2 * GT_RX, GT_WH = GT_H + 2 * GT_RY + 1, GT_VPR = GT_WW / 4, GT_NV = GT_WH * GT_VPR;
constexpr int GT_NQ = (GT_NV + 63) / 64, GT_CH = GT_NQ * 256;      // wave-sized chunks per channel; floats per channel buffer
constexpr int GT_LDS_BYTES = 3 * GT_CH * 4;
template <int CT>
__global__ void __launch_bounds__(GT_THREADS) grid_warp_tile_kernel(const float* __restrict__ second, const float* __restrict__ flow,
                                                                    float* __restrict__ out, float* __restrict__ mask, int B, int C_rt, int H, int W,
                                                                    int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) float win[];
    const int C = CT > 0 ? CT : C_rt;
    const size_t plane = (size_t)H * W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tpi = tiles_x * tiles_y, ntiles = tpi * B;
    auto issue = [&](int t, int c0, int cg) {      // all window loads of channels c0 .. c0+cg-1 of tile t: asynchronous, straight into LDS
        const int n = t / tpi, t_in = t - n * tpi;
        const int wx0 = (t_in % tiles_x) * GT_W - GT_RX, wy0 = (t_in / tiles_x) * GT_H - GT_RY;
        for (int q = wave; q < cg * GT_NQ; q += GT_WAVES) {
            const int c = q / GT_NQ, qc = q - c * GT_NQ;
            const int vid = min(qc * 64 + lane, GT_NV - 1), row = vid / GT_VPR, col = vid - row * GT_VPR;
            const int gy = min(max(wy0 + row, 0), H - 1), gx = min(max(wx0 + col * 4, 0), W - 4);
            const float* src = second + ((size_t)n * C + c0 + c) * plane + (size_t)gy * W + gx;
            __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(win + c * GT_CH + qc * 256), 16, 0, 0);
        }
    };
    auto flow_of = [&](int t, float4& f0, float4& f1) {
        const int n = t / tpi, t_in = t - n * tpi;
        const int y = min((t_in / tiles_x) * GT_H + (tid >> 4), H - 1), x4 = min((t_in % tiles_x) * GT_W + (tid & 15) * 4, W - 4);
        const size_t pix = (size_t)y * W + x4;
        f0 = *(const float4*)(flow + (size_t)n * 2 * plane + pix);
        f1 = *(const float4*)(flow + (size_t)n * 2 * plane + plane + pix);
    };
    // XCD-aware order: in every round of gridDim.x tiles the workgroups of one XCD (blockIdx % 8) take a contiguous run (two
    // tile rows at W = 1024), so the halos neighbouring windows share are served by that XCD's L2 instead of being fetched
    // once per XCD.
    auto tile_of = [&](int l) {
        const int g = gridDim.x, k = l / g, b = l - k * g;
        return ((g & 7) == 0 && (k + 1) * g <= ntiles) ? k * g + (b & 7) * (g >> 3) + (b >> 3) : l;
    };
    int l = blockIdx.x;
    if (l >= ntiles) return;
    int t = tile_of(l);
    float4 f0, f1;
    flow_of(t, f0, f1);
    issue(t, 0, min(C, 3));
    while (l < ntiles) {
        const int n = t / tpi, t_in = t - n * tpi;
        const int ox0 = (t_in % tiles_x) * GT_W, oy0 = (t_in / tiles_x) * GT_H, wx0 = ox0 - GT_RX, wy0 = oy0 - GT_RY;
        const int y = oy0 + (tid >> 4), x4 = ox0 + (tid & 15) * 4;
        const bool valid = y < H && x4 < W;
        const int yc = min(y, H - 1), xc4 = min(x4, W - 4);
        const size_t pix = (size_t)yc * W + xc4;
        const float fl0[4] = {f0.x, f0.y, f0.z, f0.w}, fl1[4] = {f1.x, f1.y, f1.z, f1.w};
        float w[4][4], m[4], wx[4][2], wy[4][2];
        unsigned inb[4], inw[4];
        int li[4], gi[4], x0s[4], y0s[4];
        // `inner`: every corner of the thread's four pixels is inside the image AND inside the staged window -- the case of all
        // waves away from the image border.  Then every in-bounds bit is set and the mask is 1 (the four weights sum to 1 within
        // a few ulp, far above the 0.999 threshold of WP:116), so the per-corner bookkeeping below is skipped altogether.
        bool inner = true;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float ix, iy;
            grid_coords(fl0[j], fl1[j], xc4 + j, yc, H, W, ix, iy);
            const float fx0 = floorf(ix), fy0 = floorf(iy);
            const bool finite = fabsf(ix) < 1.0e9f && fabsf(iy) < 1.0e9f;
            const int x0 = finite ? (int)fx0 : -10, y0 = finite ? (int)fy0 : -10;
            wx[j][0] = fx0 + 1.0f - ix; wx[j][1] = ix - fx0; wy[j][0] = fy0 + 1.0f - iy; wy[j][1] = iy - fy0;
            w[j][0] = wx[j][0] * wy[j][0];
            w[j][1] = wx[j][1] * wy[j][0];
            w[j][2] = wx[j][0] * wy[j][1];
            w[j][3] = wx[j][1] * wy[j][1];
            const int lx = x0 - wx0, ly = y0 - wy0;            // window coordinates of the north-west corner
            li[j] = ly * GT_WW + lx;
            gi[j] = y0 * W + x0;
            x0s[j] = x0; y0s[j] = y0;
            inner = inner && (unsigned)x0 < (unsigned)(W - 1) && (unsigned)y0 < (unsigned)(H - 1) && (unsigned)lx < (unsigned)(GT_WW - 1) &&
                    (unsigned)ly < (unsigned)(GT_WH - 1);
        }
        if (!inner) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x0 = x0s[j], y0 = y0s[j];
                const float wx0_ = wx[j][0], wx1 = wx[j][1], wy0_ = wy[j][0], wy1 = wy[j][1];
                const bool bx0 = x0 >= 0 && x0 < W, bx1 = x0 + 1 >= 0 && x0 + 1 < W, by0 = y0 >= 0 && y0 < H, by1 = y0 + 1 >= 0 && y0 + 1 < H;
                inb[j] = (bx0 && by0 ? 1u : 0u) | (bx1 && by0 ? 2u : 0u) | (bx0 && by1 ? 4u : 0u) | (bx1 && by1 ? 8u : 0u);
                float ms = 0.0f;                                   // sum of in-bounds weights = grid_sample(ones)
                if (bx0 && by0) ms += wx0_ * wy0_;
                if (bx1 && by0) ms += wx1 * wy0_;
                if (bx0 && by1) ms += wx0_ * wy1;
                if (bx1 && by1) ms += wx1 * wy1;
                if (ms < 0.999f) ms = 0.0f;                        // WP:116-117
                if (ms > 0.0f) ms = 1.0f;
                m[j] = ms;
                const int lx = x0 - wx0, ly = y0 - wy0;
                const bool wxa = lx >= 0 && lx < GT_WW, wxb = lx + 1 >= 0 && lx + 1 < GT_WW, wya = ly >= 0 && ly < GT_WH, wyb = ly + 1 >= 0 && ly + 1 < GT_WH;
                inw[j] = (wxa && wya ? 1u : 0u) | (wxb && wya ? 2u : 0u) | (wxa && wyb ? 4u : 0u) | (wxb && wyb ? 8u : 0u);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { inb[j] = 15u; inw[j] = 15u; m[j] = 1.0f; }
        }
        const bool all_in_window = (inw[0] & inw[1] & inw[2] & inw[3]) == 15u;
        const int ln = l + gridDim.x, tn = ln < ntiles ? tile_of(ln) : ntiles;
        for (int c0 = 0; c0 < C; c0 += 3) {
            const int cg = min(C - c0, 3);
            if (c0 > 0) {
                __syncthreads();                               // the previous group's reads are done
                issue(t, c0, cg);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (c0 == 0 && tn < ntiles) flow_of(tn, f0, f1);   // next tile's flow: in flight during the gather
            for (int cc = 0; cc < cg; ++cc) {
                const int c = c0 + cc;
                const float* sp = second + ((size_t)n * C + c) * plane;
                const float* wc = win + cc * GT_CH;
                float o[4];
                if (inner) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float c0v = wc[li[j]], c1v = wc[li[j] + 1], c2v = wc[li[j] + GT_WW], c3v = wc[li[j] + GT_WW + 1];
                        float acc = c0v * w[j][0];
                        acc += c1v * w[j][1];
                        acc += c2v * w[j][2];
                        acc += c3v * w[j][3];
                        o[j] = acc;
                    }
                } else if (all_in_window) {
                    // the four corners of all four pixels are staged -> paired LDS reads, in-bounds bits gate the sum
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float c0v = wc[li[j]], c1v = wc[li[j] + 1], c2v = wc[li[j] + GT_WW], c3v = wc[li[j] + GT_WW + 1];
                        float acc = 0.0f;
                        if (inb[j] & 1u) acc += c0v * w[j][0];
                        if (inb[j] & 2u) acc += c1v * w[j][1];
                        if (inb[j] & 4u) acc += c2v * w[j][2];
                        if (inb[j] & 8u) acc += c3v * w[j][3];
                        o[j] = acc;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float cv[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int dl = (k & 1) + (k >> 1) * GT_WW, dg = (k & 1) + (k >> 1) * W;
                            const bool in_win = (inw[j] >> k) & 1u, in_img = (inb[j] >> k) & 1u;
                            float v1 = wc[in_win ? li[j] + dl : 0];
                            // (opaque to the optimiser: folding the two loads into one load through a generic LDS-or-global
                            //  pointer trips "Illegal instruction: V_CMP_NE_U32 0, $src_shared_base" in hipcc 7.2)
                            asm volatile("" : "+v"(v1));
                            if (in_img && !in_win) v1 = sp[(size_t)(gi[j] + dg)];      // beyond the staged window: rare
                            cv[k] = v1;
                        }
                        // only in-bounds corners contribute (zeros padding); ATen's nw, ne, sw, se order
                        float acc = 0.0f;
                        if (inb[j] & 1u) acc += cv[0] * w[j][0];
                        if (inb[j] & 2u) acc += cv[1] * w[j][1];
                        if (inb[j] & 4u) acc += cv[2] * w[j][2];
                        if (inb[j] & 8u) acc += cv[3] * w[j][3];
                        o[j] = acc;
                    }
                }
                if (valid) {
                    *(float4*)(out + ((size_t)n * C + c) * plane + pix) = make_float4(o[0], o[1], o[2], o[3]);
                    if (mask) *(float4*)(mask + ((size_t)n * C + c) * plane + pix) = make_float4(m[0], m[1], m[2], m[3]);
                }
            }
        }
        __syncthreads();                                       // LDS is free again
        if (tn < ntiles) issue(tn, 0, min(C, 3));
        t = tn;
        l = ln;
    }
}

// Ring variant (C = 3; r03): the tile kernel above runs its phases in series -- while a tile is gathered and written out no window
// load is in flight, because the three channel windows are ONE buffer group that is refilled only when the whole tile is done
// (ablation: coordinates 27 + window loads 25 + gathers 20 + stores 17 us of a 73 us launch).  Same tiles, same 3 x 47 KB of LDS,
// same arithmetic, but the three single-channel windows are a RING over (tile, channel) items: item k is gathered from window
// k % 3 (= its channel) while the windows of items k + 1 and k + 2 are in flight, and the window an item frees is refilled one
// barrier later.  One workgroup barrier and one COUNTED vmcnt wait per item:
//   per wave the VMEM issue order is  ... [DMA(k+2): d pieces] [flow of the next tile: 2, with channel 0] [stores(k): S] ...  (in-order
//   return), so "all of DMA(k) has landed" = all but the youngest 2 S + d (+ 2) operations have; d = 3 pieces per wave and
//   channel (2 for the last wave: 47 pieces over 16 waves), S = 2 stores (1 without a mask).  For the count to be the same for every
//   wave at every item, nothing in the loop is conditional: threads outside the image park their stores in a dump buffer, the
//   loads past the workgroup's last item fetch that item again (into a window nobody reads any more).
// The flow loads and the LDS reads of the common path are inline asm: a load the compiler can see makes it drain every outstanding
// LDS-DMA (vmcnt(0)) before the first use -- the serialisation this variant removes.  Border / out-of-window threads keep the
// compiler-visible path (correct under any extra wait).  Bit-identical to the kernels above.
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
// Registers written by an asm load must not be touched before the counted wait that covers the load (a register operand on the
// wait itself makes the compiler shuffle them into place BEFORE it -- copies of data still in flight).  The waits carry no operands;
// this empty statement, placed after the wait, is what the consumers of the registers cannot be scheduled above.
__device__ __forceinline__ void gw_pin(f32x4v& a, f32x4v& b) { asm volatile("" : "+v"(a), "+v"(b) :: "memory"); }

// TH x 64 output tiles, TH * 16 threads, NBUF single-channel windows:
//   <64, 3>: one 1024-thread workgroup per CU, look-ahead of two items (the form described above);
//   <32, 2>: two 512-thread workgroups per CU (2 x 67 KB of LDS), look-ahead of one item each.  The waves of ONE workgroup run in step
//            (a barrier per item), so its phases -- window DMA issue, coordinates, LDS gather, stores -- add up whatever is in flight
//            (same-box ablation of <64, 3>: skeleton 27 + DMA 15 + gather 8 + stores 12 + divisions 6 us, removals additive); a second,
//            independent workgroup on the CU is what fills one's memory phases with the other's arithmetic.
template <int TH, int NBUF>
struct GwRing {
    static constexpr int THREADS = TH * 16, WAVES = THREADS / 64;
    static constexpr int WH = TH + 2 * GT_RY + 1, NV = WH * GT_VPR, NQ = (NV + 63) / 64, CH = NQ * 256;   // window rows, 16-byte vectors, wave-sized pieces, floats
    static constexpr int PMAX = (NQ + WAVES - 1) / WAVES, PFULL = NQ - (PMAX - 1) * WAVES;                  // pieces per wave: PMAX for waves < PFULL, else PMAX - 1
    static constexpr int LDS_BYTES = NBUF * CH * 4;
    static_assert(NBUF == 2 || NBUF == 3, "look-ahead of one or two items");
};

template <bool MASK, int TH, int NBUF>
__global__ void __launch_bounds__(TH * 16) grid_warp_ring_kernel(const float* __restrict__ second, const float* __restrict__ flow,
                                                                 float* __restrict__ out, float* __restrict__ mask, int B, int H, int W,
                                                                 int tiles_x, int tiles_y, int dbg) {
    extern __shared__ __attribute__((aligned(16))) float win[];
    using R = GwRing<TH, NBUF>;
    constexpr int C = 3, S = MASK ? 2 : 1, PMAX = R::PMAX;
    const size_t plane = (size_t)H * W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool three = wave < R::PFULL;                       // this wave issues PMAX pieces per window (else PMAX - 1)
    const int tpi = tiles_x * tiles_y, ntiles = tpi * B, g = (int)gridDim.x, bidx = (int)blockIdx.x;
    if (bidx >= ntiles) return;
    const int nt = (ntiles - 1 - bidx) / g + 1;               // tiles of this workgroup: l = bidx + i g
    const int xslot = (bidx & 7) * (g >> 3) + (bidx >> 3);    // XCD-aware order inside a full round of g tiles (see grid_warp_tile_kernel)
    const bool g8 = (g & 7) == 0;
    // everything about a tile that is the same for the whole workgroup: scalar registers, decoded ONCE per tile
    struct Tile { int n, ox0, oy0; };
    auto tile_at = [&](int i) {
        i = min(i, nt - 1);                                   // past the end: the last tile again (loads that nobody consumes)
        const int t = (g8 && (i + 1) * g <= ntiles) ? i * g + xslot : bidx + i * g;
        const int n = t / tpi, t_in = t - n * tpi, ty = t_in / tiles_x;
        return Tile{n, (t_in - ty * tiles_x) * GT_W, ty * TH};
    };
    // per-lane constants of the window pieces this wave issues: piece p = chunk wave + 16 p of a channel window
    int prow[PMAX], pcol[PMAX];
    unsigned poff[PMAX];                                      // byte offset of the piece's 16 bytes from the window origin (no clamping: interior windows)
#pragma unroll
    for (int p = 0; p < PMAX; ++p) {
        const int vid = min((wave + R::WAVES * p) * 64 + lane, R::NV - 1);
        prow[p] = vid / GT_VPR;
        pcol[p] = (vid - prow[p] * GT_VPR) * 4;
        poff[p] = (unsigned)(prow[p] * W + pcol[p]) * 4u;
    }
    auto issue = [&](const Tile& T, int c, int buf) {         // this wave's pieces of channel c's window of tile T, straight into window `buf`
        if (dbg & 1) return;                                  // (timing ablation, OFD_GW_DBG: results are wrong with any bit set)
        const int wx0 = T.ox0 - GT_RX, wy0 = T.oy0 - GT_RY;
        const float* base = second + ((size_t)T.n * C + c) * plane;
        const bool interior = wx0 >= 0 && wy0 >= 0 && wx0 + GT_WW <= W && wy0 + R::WH <= H;     // (uniform) no clamping needed
        // interior window (62 % of the tiles at 440 x 1024): uniform base of the window origin + the lane's precomputed offset, no
        // per-piece arithmetic; border windows clamp row and column (positions outside the image are never read back)
        const float* wbase = base + (interior ? wy0 * W + wx0 : 0);
        auto piece = [&](int p) {
            unsigned off = poff[p];
            if (!interior) {
                const int gy = min(max(wy0 + prow[p], 0), H - 1), gx = min(max(wx0 + pcol[p], 0), W - 4);
                off = (unsigned)(gy * W + gx) * 4u;
            }
            __builtin_amdgcn_global_load_lds((const float*)((const char*)wbase + off),
                                             (__attribute__((address_space(3))) void*)(win + buf * R::CH + (wave + R::WAVES * p) * 256), 16, 0, 0);
        };
#pragma unroll
        for (int p = 0; p < PMAX - 1; ++p) piece(p);
        if (three) piece(PMAX - 1);
    };
    // this thread's four pixels: row ty, columns tx4 .. tx4 + 3 of the tile.  Threads past the image edge take the last row / the last
    // quad of columns instead: they repeat a neighbour's work bit for bit and store the same values to the same place, so no load or
    // store in the loop is conditional (the counted waits need a fixed number of them per wave).
    const int ty = tid >> 4, tx4 = (tid & 15) * 4;
    f32x4v f0, f1;
    auto flow_issue = [&](const Tile& T) {                    // two 16-byte loads the compiler does not count
        const int y = min(T.oy0 + ty, H - 1), x4 = min(T.ox0 + tx4, W - 4);
        const unsigned off = (unsigned)(y * W + x4) * 4u;
        const float* p0 = flow + (size_t)T.n * 2 * plane;
        const float* p1 = p0 + plane;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(f0) : "v"(off), "s"(p0) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(f1) : "v"(off), "s"(p1) : "memory");
    };
    const unsigned win_addr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)win;
    const float dwf = (float)max(W - 1, 1), dhf = (float)max(H - 1, 1), rwf = 1.0f / dwf, rhf = 1.0f / dhf;

    // per-tile state of this thread's four pixels (computed once per tile, used by its three channel items)
    float w[4][4], m[4];
    unsigned inb[4], inw[4], pixb = 0;
    int li[4], gi[4];
    bool inner = true, all_in_window = true;
    auto coords = [&](const Tile& T) {                         // from the flow registers f0, f1 (which must have landed)
        const int wx0 = T.ox0 - GT_RX, wy0 = T.oy0 - GT_RY;
        const int ixlo = max(0, wx0), iylo = max(0, wy0);
        const unsigned ixspan = (unsigned)(min(W, wx0 + GT_WW) - 2 - ixlo), iyspan = (unsigned)(min(H, wy0 + R::WH) - 2 - iylo);
        const int yc = min(T.oy0 + ty, H - 1), xc4 = min(T.ox0 + tx4, W - 4);
        pixb = (unsigned)(yc * W + xc4) * 4u;
        const float fl0[4] = {f0.x, f0.y, f0.z, f0.w}, fl1[4] = {f1.x, f1.y, f1.z, f1.w};
        float wx[4][2], wy[4][2];
        int x0s[4], y0s[4];
        inner = true;
        all_in_window = true;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float ix, iy;
            if (dbg & 8) { ix = (float)(xc4 + j) + fl1[j]; iy = (float)yc + fl0[j]; }
            else grid_coords_rcp(fl0[j], fl1[j], xc4 + j, yc, H, W, dwf, rwf, dhf, rhf, ix, iy);
            const float fx0 = floorf(ix), fy0 = floorf(iy);
            const bool finite = fabsf(ix) < 1.0e9f && fabsf(iy) < 1.0e9f;
            const int x0 = finite ? (int)fx0 : -10, y0 = finite ? (int)fy0 : -10;
            wx[j][0] = fx0 + 1.0f - ix; wx[j][1] = ix - fx0; wy[j][0] = fy0 + 1.0f - iy; wy[j][1] = iy - fy0;
            w[j][0] = wx[j][0] * wy[j][0];
            w[j][1] = wx[j][1] * wy[j][0];
            w[j][2] = wx[j][0] * wy[j][1];
            w[j][3] = wx[j][1] * wy[j][1];
            const int lx = x0 - wx0, ly = y0 - wy0;            // window coordinates of the north-west corner
            li[j] = ly * GT_WW + lx;
            gi[j] = y0 * W + x0;
            x0s[j] = x0; y0s[j] = y0;
            // both corners inside the image AND inside the window, per axis as ONE range test: x0 in [max(0, wx0), min(W, wx0 + WW) - 2]
            inner = inner && (unsigned)(x0 - ixlo) <= ixspan && (unsigned)(y0 - iylo) <= iyspan;
        }
        if (!inner) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x0 = x0s[j], y0 = y0s[j];
                const float wx0_ = wx[j][0], wx1 = wx[j][1], wy0_ = wy[j][0], wy1 = wy[j][1];
                const bool bx0 = x0 >= 0 && x0 < W, bx1 = x0 + 1 >= 0 && x0 + 1 < W, by0 = y0 >= 0 && y0 < H, by1 = y0 + 1 >= 0 && y0 + 1 < H;
                inb[j] = (bx0 && by0 ? 1u : 0u) | (bx1 && by0 ? 2u : 0u) | (bx0 && by1 ? 4u : 0u) | (bx1 && by1 ? 8u : 0u);
                float ms = 0.0f;                               // sum of in-bounds weights = grid_sample(ones)
                if (bx0 && by0) ms += wx0_ * wy0_;
                if (bx1 && by0) ms += wx1 * wy0_;
                if (bx0 && by1) ms += wx0_ * wy1;
                if (bx1 && by1) ms += wx1 * wy1;
                if (ms < 0.999f) ms = 0.0f;                    // WP:116-117
                if (ms > 0.0f) ms = 1.0f;
                m[j] = ms;
                const int lx = x0 - wx0, ly = y0 - wy0;
                const bool wxa = lx >= 0 && lx < GT_WW, wxb = lx + 1 >= 0 && lx + 1 < GT_WW, wya = ly >= 0 && ly < R::WH, wyb = ly + 1 >= 0 && ly + 1 < R::WH;
                inw[j] = (wxa && wya ? 1u : 0u) | (wxb && wya ? 2u : 0u) | (wxa && wyb ? 4u : 0u) | (wxb && wyb ? 8u : 0u);
            }
            all_in_window = (inw[0] & inw[1] & inw[2] & inw[3]) == 15u;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { inb[j] = 15u; inw[j] = 15u; m[j] = 1.0f; }
        }
    };
    auto gather_store = [&](const Tile& T, int c, int buf) {
        const size_t cbase = ((size_t)T.n * C + c) * plane;    // (uniform)
        const float* sp = second + cbase;
        const float* wc = win + buf * R::CH;
        float o[4];
        if (dbg & 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = w[j][0] + w[j][1] + w[j][2] + w[j][3];
        } else if (inner) {
            f32x2v top[4], bot[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned a = win_addr + (unsigned)(buf * R::CH + li[j]) * 4u;
                asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read2_b32 %1, %2 offset0:%3 offset1:%4"
                             : "=&v"(top[j]), "=&v"(bot[j]) : "v"(a), "n"(GT_WW), "n"(GT_WW + 1) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(top[0]), "+v"(top[1]), "+v"(top[2]), "+v"(top[3]), "+v"(bot[0]), "+v"(bot[1]), "+v"(bot[2]), "+v"(bot[3]) :: "memory");
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float acc = top[j].x * w[j][0];
                acc += top[j].y * w[j][1];
                acc += bot[j].x * w[j][2];
                acc += bot[j].y * w[j][3];
                o[j] = acc;
            }
        } else if (all_in_window) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float c0v = wc[li[j]], c1v = wc[li[j] + 1], c2v = wc[li[j] + GT_WW], c3v = wc[li[j] + GT_WW + 1];
                float acc = 0.0f;
                if (inb[j] & 1u) acc += c0v * w[j][0];
                if (inb[j] & 2u) acc += c1v * w[j][1];
                if (inb[j] & 4u) acc += c2v * w[j][2];
                if (inb[j] & 8u) acc += c3v * w[j][3];
                o[j] = acc;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float cv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int dl = (k & 1) + (k >> 1) * GT_WW, dg = (k & 1) + (k >> 1) * W;
                    const bool in_win = (inw[j] >> k) & 1u, in_img = (inb[j] >> k) & 1u;
                    float v1 = wc[in_win ? li[j] + dl : 0];
                    asm volatile("" : "+v"(v1));                               // (see grid_warp_tile_kernel)
                    if (in_img && !in_win) v1 = sp[(size_t)(gi[j] + dg)];      // beyond the staged window: rare
                    cv[k] = v1;
                }
                float acc = 0.0f;
                if (inb[j] & 1u) acc += cv[0] * w[j][0];
                if (inb[j] & 2u) acc += cv[1] * w[j][1];
                if (inb[j] & 4u) acc += cv[2] * w[j][2];
                if (inb[j] & 8u) acc += cv[3] * w[j][3];
                o[j] = acc;
            }
        }
        if ((dbg & 4) && o[0] != 12345.0f) return;
        // unconditional stores: uniform channel base + this thread's 32-bit byte offset
        *(float4*)((char*)(out + cbase) + pixb) = make_float4(o[0], o[1], o[2], o[3]);
        if constexpr (MASK) *(float4*)((char*)(mask + cbase) + pixb) = make_float4(m[0], m[1], m[2], m[3]);
    };
    auto wait_plain = [&](auto nc) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(nc)::value) : "memory"); };

    Tile T = tile_at(0);
    flow_issue(T);
    issue(T, 0, 0);
    if constexpr (NBUF == 3) issue(T, 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    gw_pin(f0, f1);
    coords(T);

    // Counted waits.  Per wave the VMEM issue order of one tile (d = window pieces per wave, S = stores per item) is
    //   NBUF 3: [flow(next): 2] [DMA(i,2): d] [S] | [DMA(i+1,0): d] [S] | [DMA(i+1,1): d] [S]      item k gathers window k % 3 = its channel
    //   NBUF 2: [DMA(i,1): d] [flow(next): 2] [S] | [DMA(i,2): d] [S] | [DMA(i+1,0): d] [S]         item k gathers window k % 2
    // "the window about to be gathered has landed" = all but the youngest N operations have; the flow loads are older than the window
    // item (i, 2) waits for, so they have landed there as well.
    constexpr int DA = PMAX, DB = PMAX - 1;
    constexpr int N0a = NBUF == 3 ? 2 * S + DA : S, N0b = NBUF == 3 ? 2 * S + DB : S;
    constexpr int N1a = NBUF == 3 ? 2 * S + DA + 2 : S + 2, N1b = NBUF == 3 ? 2 * S + DB + 2 : S + 2;
    for (int i = 0; i < nt; ++i) {
        const Tile Tn = tile_at(i + 1);
        const int k0 = 3 * i;                                  // item index of (i, 0): window of item k is k % NBUF
        const int b0 = NBUF == 3 ? 0 : (k0 & 1), b1 = NBUF == 3 ? 1 : (b0 ^ 1), b2 = NBUF == 3 ? 2 : b0, b3 = NBUF == 3 ? 0 : (b0 ^ 1);
        // ---- item (i, 0)
        if (three) wait_plain(std::integral_constant<int, N0a>{}); else wait_plain(std::integral_constant<int, N0b>{});
        __builtin_amdgcn_s_barrier();                          // this item's window is complete; everybody is done with the previous item's
        if constexpr (NBUF == 3) { flow_issue(Tn); issue(T, 2, b2); }
        else { issue(T, 1, b1); flow_issue(Tn); }
        gather_store(T, 0, b0);
        // ---- item (i, 1)
        if (three) wait_plain(std::integral_constant<int, N1a>{}); else wait_plain(std::integral_constant<int, N1b>{});
        __builtin_amdgcn_s_barrier();
        if constexpr (NBUF == 3) issue(Tn, 0, b3); else issue(T, 2, b2);
        gather_store(T, 1, b1);
        // ---- item (i, 2): the next tile's flow has landed too
        if (three) wait_plain(std::integral_constant<int, N0a>{}); else wait_plain(std::integral_constant<int, N0b>{});
        __builtin_amdgcn_s_barrier();
        gw_pin(f0, f1);
        if constexpr (NBUF == 3) issue(Tn, 1, 1); else issue(Tn, 0, b3);
        gather_store(T, 2, b2);
        coords(Tn);                                            // the next tile's coordinates, while its windows are in flight
        T = Tn;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the look-ahead loads of the last items land before the workgroup gives its LDS back
}

// Band variant (C = 3; r04).  The memory-only forms of the decompositions (tools/probe/gw_stream_probe.hip: window DMA, flow loads and stores of a
// launch with no arithmetic and no waits) say what the 64 x 64 tile kernels above are bound by: 69-71 us against 53-55 us for a linear copy of
// the same planes -- every tile fetches a window 2.9 x its own size, and although L2 serves most of the overlap the window traffic is what the
// launch is made of (the tile kernels measure 70-73 us: at their pattern's bound, which is why three schedules of them changed nothing).
// Wide-short TILES are worse (8 x 512: 87 us; the window is 7 x the tile); what streams is a workgroup that owns a COLUMN BAND and slides down
// it, fetching every source row once per band: 58-64 us memory-only.  LDS holds the three channels, so the band is 128 columns wide:
//   * workgroup = (sample, band of 128 columns, segment of rows): 1024 threads, one pixel each per step of 8 output rows (wave = half a row);
//   * the window of a step is 7 groups of 8 source rows (rows y - 24 .. y + 31 of the band's columns x - 24 .. x + 151, three channels);
//     LDS is a ring of 9 such groups (3 x 72 x 176 floats = 152 KB): step s gathers from groups s .. s + 6 while group s + 8 -- needed two
//     steps later -- arrives by LDS-DMA (global_load_lds_dwordx4) into the slot group s - 1 has left; ONE barrier and ONE counted vmcnt wait per
//     step (per wave the VMEM order of a step is [flow of the next step: 2] [DMA: 1 or 2 pieces] [stores: 3 or 6]; "flow(s) and group s + 6
//     have landed" = all but the youngest d + S operations have);
//   * a source row enters LDS once per band and segment: window traffic 1.4 x (column halo) x 1.2 (the 24 + 31 warm-up rows of a segment)
//     of the image instead of 2.9 x;
//   * same arithmetic, operation order and border rules as the kernels above: bit-identical results.
// Flow loads and the LDS reads of the common path are inline asm (a load the compiler can see makes it drain the LDS-DMA in flight); rows past
// the end of a segment / columns past the image repeat the last row / column (the same values stored to the same place).
constexpr int GB_W = 128, GB_TH = 8, GB_SLOTS = 9, GB_NEED = 7, GB_RX = 24, GB_RYUP = 24, GB_THREADS = 1024;
constexpr int GB_WW = GB_W + 2 * GB_RX, GB_ROWS = GB_SLOTS * GB_TH, GB_CH = GB_ROWS * GB_WW;     // 176 columns, 72 rows, floats per channel ring
constexpr int GB_LDS_BYTES = 3 * GB_CH * 4;
constexpr int GB_GRP = GB_TH * GB_WW;                      // floats of one channel of a group: 1408 = 5.5 wave-pieces of 256 floats

template <bool MASK>
__global__ void __launch_bounds__(GB_THREADS) grid_warp_band_kernel(const float* __restrict__ second, const float* __restrict__ flow,
                                                                    float* __restrict__ out, float* __restrict__ mask, int B, int H, int W,
                                                                    int bands, int nseg, int seg_rows) {
    extern __shared__ __attribute__((aligned(16))) float win[];
    constexpr int C = 3, S = MASK ? 6 : 3;
    const size_t plane = (size_t)H * W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // block -> (sample, band, segment); the workgroups of one XCD (block % 8) take neighbouring units: the column halos two bands share and the
    // warm-up rows two segments share come from that XCD's L2
    int j = blockIdx.x;
    {
        const int g = (int)gridDim.x;
        if ((g & 7) == 0) j = (j & 7) * (g >> 3) + (j >> 3);
    }
    const int seg = j % nseg, band = (j / nseg) % bands, n = j / (nseg * bands);
    if (n >= B) return;
    const int r0 = seg * seg_rows, r1 = min(r0 + seg_rows, H);
    if (r0 >= r1) return;
    const int ox0 = band * GB_W, wx0 = ox0 - GB_RX, rbase = r0 - GB_RYUP;
    const int nsteps = (r1 - r0 + GB_TH - 1) / GB_TH;

    // ---- window pieces of this wave: piece id 0..14 = full piece k = id % 5 of channel id / 5, 15..17 = the half piece (k = 5, lanes 0..31) of
    //      channel id - 15.  Wave w issues piece w; waves 0 and 1 also pieces 16 and 17.
    const bool two = wave < 2;
    auto piece_geom = [&](int id, int& c, int& k, bool& half) { half = id >= 15; c = half ? id - 15 : id / 5; k = half ? 5 : id % 5; };
    int c_a, k_a, c_b = 0, k_b = 0;
    bool half_a, half_b = true;
    piece_geom(wave, c_a, k_a, half_a);
    if (two) piece_geom(16 + wave, c_b, k_b, half_b);
    auto lane_rc = [&](int k, int& row, int& col) { const int f = 256 * k + 4 * lane; row = f / GB_WW; col = f - row * GB_WW; };
    int prow_a, pcol_a, prow_b, pcol_b;
    lane_rc(k_a, prow_a, pcol_a);
    lane_rc(k_b, prow_b, pcol_b);
    const float* const img_n = second + (size_t)n * C * plane;
    auto dma_piece = [&](int q, int c, int k, bool half, int prow, int pcol) {        // this wave's piece of group q -> slot q % 9
        const int gy = min(max(rbase + q * GB_TH + min(prow, GB_TH - 1), 0), H - 1), gx = min(max(wx0 + pcol, 0), W - 4);
        const float* src = img_n + (size_t)c * plane + (size_t)gy * W + gx;
        float* dst = win + c * GB_CH + (q % GB_SLOTS) * GB_GRP + 256 * k;
        if (!half || lane < 32) __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    auto dma_group = [&](int q) {
        dma_piece(q, c_a, k_a, half_a, prow_a, pcol_a);
        if (two) dma_piece(q, c_b, k_b, half_b, prow_b, pcol_b);
    };

    // ---- this thread's pixel of a step: row ty of the 8, column tx of the band (clamped into the image: duplicates store the same value)
    const int ty = wave >> 1, tx = (wave & 1) * 64 + lane;
    const int xc = min(ox0 + tx, W - 1);
    const float* const flow_n = flow + (size_t)n * 2 * plane;
    float f0, f1;
    auto flow_issue = [&](int s) {                            // two loads the compiler does not count
        const int yy = min(r0 + s * GB_TH + ty, r1 - 1);
        const unsigned off = (unsigned)(yy * W + xc) * 4u;
        const float* p0 = flow_n;
        const float* p1 = flow_n + plane;
        asm volatile("global_load_dword %0, %1, %2" : "=v"(f0) : "v"(off), "s"(p0) : "memory");
        asm volatile("global_load_dword %0, %1, %2" : "=v"(f1) : "v"(off), "s"(p1) : "memory");
    };
    const unsigned win_addr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)win;
    const float dwf = (float)max(W - 1, 1), dhf = (float)max(H - 1, 1), rwf = 1.0f / dwf, rhf = 1.0f / dhf;
    const int ixlo = max(0, wx0);
    const unsigned ixspan = (unsigned)(min(W, wx0 + GB_WW) - 2 - ixlo);

    // ---- fill: groups 0 .. 7 (step 0 needs 0 .. 6), the flow of step 0
    flow_issue(0);
    for (int q = 0; q < GB_SLOTS - 1; ++q) dma_group(q);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(f0), "+v"(f1) :: "memory");

    for (int s = 0; s < nsteps; ++s) {
        __builtin_amdgcn_s_barrier();                          // group s + 6 is complete; everybody is done with step s - 1 (the slot of group s - 1 is free)
        const int ystep = r0 + s * GB_TH, wy0 = rbase + s * GB_TH;
        const int yy = min(ystep + ty, r1 - 1);
        // coordinates of this pixel (WP:105-109 -> ATen's un-normalisation), from the flow registers
        float ix, iy;
        grid_coords_rcp(f0, f1, xc, yy, H, W, dwf, rwf, dhf, rhf, ix, iy);
        // the next step's flow, then the window group two steps ahead (in this order: the wait at the top of the next step leaves the DMA in flight)
        flow_issue(min(s + 1, nsteps - 1));
        dma_group(s + GB_SLOTS - 1);
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const bool finite = fabsf(ix) < 1.0e9f && fabsf(iy) < 1.0e9f;
        const int x0 = finite ? (int)fx0 : -10, y0 = finite ? (int)fy0 : -10;
        const float wxa = fx0 + 1.0f - ix, wxb = ix - fx0, wya = fy0 + 1.0f - iy, wyb = iy - fy0;
        const float w0 = wxa * wya, w1 = wxb * wya, w2 = wxa * wyb, w3 = wxb * wyb;
        const int lx = x0 - wx0, lt = y0 - wy0;               // window coordinates of the north-west corner: column 0..175, row 0..55 of the step's window
        const int iylo = max(0, wy0);
        const unsigned iyspan = (unsigned)(min(H, wy0 + GB_NEED * GB_TH) - 2 - iylo);
        // both corners inside the image AND inside the staged window, per axis as ONE range test
        const bool inner = (unsigned)(x0 - ixlo) <= ixspan && (unsigned)(y0 - iylo) <= iyspan;
        const unsigned sbase = (unsigned)((s % GB_SLOTS) * GB_TH);
        auto ring_row = [&](int t) { const unsigned r = sbase + (unsigned)t; return min(r, r - (unsigned)GB_ROWS); };       // (t in 0..56: one wrap at most)
        const size_t pix = (size_t)yy * W + xc;
        float m = 1.0f;
        float o[C];
        if (inner) {
            const unsigned li = ring_row(lt) * GB_WW + (unsigned)lx;
            unsigned ls = li + GB_WW;
            ls = min(ls, ls - (unsigned)GB_CH);                // the south row of ring row 71 is ring row 0
            f32x2v top[C], bot[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const unsigned a0 = win_addr + (li + c * GB_CH) * 4u, a1 = win_addr + (ls + c * GB_CH) * 4u;
                asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read2_b32 %1, %3 offset1:1" : "=&v"(top[c]), "=&v"(bot[c]) : "v"(a0), "v"(a1) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(top[0]), "+v"(top[1]), "+v"(top[2]), "+v"(bot[0]), "+v"(bot[1]), "+v"(bot[2]) :: "memory");
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float acc = top[c].x * w0;
                acc += top[c].y * w1;
                acc += bot[c].x * w2;
                acc += bot[c].y * w3;
                o[c] = acc;
            }
        } else {
            const bool bx0 = x0 >= 0 && x0 < W, bx1 = x0 + 1 >= 0 && x0 + 1 < W, by0 = y0 >= 0 && y0 < H, by1 = y0 + 1 >= 0 && y0 + 1 < H;
            const unsigned inb = (bx0 && by0 ? 1u : 0u) | (bx1 && by0 ? 2u : 0u) | (bx0 && by1 ? 4u : 0u) | (bx1 && by1 ? 8u : 0u);
            float ms = 0.0f;                                   // sum of in-bounds weights = grid_sample(ones)
            if (bx0 && by0) ms += wxa * wya;
            if (bx1 && by0) ms += wxb * wya;
            if (bx0 && by1) ms += wxa * wyb;
            if (bx1 && by1) ms += wxb * wyb;
            if (ms < 0.999f) ms = 0.0f;                        // WP:116-117
            if (ms > 0.0f) ms = 1.0f;
            m = ms;
            const bool cxa = lx >= 0 && lx < GB_WW, cxb = lx + 1 >= 0 && lx + 1 < GB_WW;
            const bool rya = lt >= 0 && lt < GB_NEED * GB_TH, ryb = lt + 1 >= 0 && lt + 1 < GB_NEED * GB_TH;
            const unsigned inw = (cxa && rya ? 1u : 0u) | (cxb && rya ? 2u : 0u) | (cxa && ryb ? 4u : 0u) | (cxb && ryb ? 8u : 0u);
            const int gi = y0 * W + x0;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float* sp = img_n + (size_t)c * plane;
                float cv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool in_win = (inw >> k) & 1u, in_img = (inb >> k) & 1u;
                    const unsigned idx = in_win ? ring_row(lt + (k >> 1)) * GB_WW + (unsigned)(lx + (k & 1)) + c * GB_CH : 0u;
                    float v1 = win[idx];
                    asm volatile("" : "+v"(v1));                               // (see grid_warp_tile_kernel)
                    if (in_img && !in_win) v1 = sp[(size_t)(gi + (k & 1) + (k >> 1) * W)];      // beyond the staged window: rare
                    cv[k] = v1;
                }
                // only in-bounds corners contribute (zeros padding); ATen's nw, ne, sw, se order
                float acc = 0.0f;
                if (inb & 1u) acc += cv[0] * w0;
                if (inb & 2u) acc += cv[1] * w1;
                if (inb & 4u) acc += cv[2] * w2;
                if (inb & 8u) acc += cv[3] * w3;
                o[c] = acc;
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            out[((size_t)n * C + c) * plane + pix] = o[c];
            if constexpr (MASK) mask[((size_t)n * C + c) * plane + pix] = m;
        }
        // the next step's flow and window group s + 7 have landed once all but the youngest (pieces of this step + stores) operations have
        if (two) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 + S) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 + S) : "memory");
        asm volatile("" : "+v"(f0), "+v"(f1) :: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the look-ahead groups land before the workgroup gives its LDS back
}

// Gradient of warp_backward_flow's output with respect to the flow (ATen grid_sampler_2d_backward's grid gradient chained
// through the reference's normalisation WP:108-109; the thresholded mask has no gradient).  A gather: one thread per pixel.
__global__ void __launch_bounds__(256) grid_warp_flowgrad_kernel(const float* __restrict__ second, const float* __restrict__ flow,
                                                                 const float* __restrict__ gout, float* __restrict__ gflow, int B, int C, int H, int W) {
    const size_t plane = (size_t)H * W, total = (size_t)B * plane;
    const float mx = (float)(W - 1) / 2.0f, my = (float)(H - 1) / 2.0f;          // align_corners un-normalise multipliers
    const float dx = (float)max(W - 1, 1), dy = (float)max(H - 1, 1);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / plane, pix = i % plane;
        const int y = (int)(pix / W), x = (int)(pix % W);
        float ix, iy;
        grid_coords(flow[n * 2 * plane + pix], flow[n * 2 * plane + plane + pix], x, y, H, W, ix, iy);
        float gix = 0.0f, giy = 0.0f;
        if (fabsf(ix) < 1.0e9f && fabsf(iy) < 1.0e9f) {
            const float fx0 = floorf(ix), fy0 = floorf(iy);
            const int x0 = (int)fx0, y0 = (int)fy0;
            const bool bx0 = x0 >= 0 && x0 < W, bx1 = x0 + 1 >= 0 && x0 + 1 < W, by0 = y0 >= 0 && y0 < H, by1 = y0 + 1 >= 0 && y0 + 1 < H;
            const float ex = fx0 + 1.0f - ix, wx = ix - fx0, ey = fy0 + 1.0f - iy, wy = iy - fy0;
            for (int c = 0; c < C; ++c) {
                const float* sp = second + (n * C + c) * plane;
                const float go = gout[(n * C + c) * plane + pix];
                if (bx0 && by0) { const float v = sp[(size_t)y0 * W + x0];           gix -= v * ey * go; giy -= v * ex * go; }
                if (bx1 && by0) { const float v = sp[(size_t)y0 * W + x0 + 1];       gix += v * ey * go; giy -= v * wx * go; }
                if (bx0 && by1) { const float v = sp[(size_t)(y0 + 1) * W + x0];     gix -= v * wy * go; giy += v * ex * go; }
                if (bx1 && by1) { const float v = sp[(size_t)(y0 + 1) * W + x0 + 1]; gix += v * wy * go; giy += v * wx * go; }
            }
        }
        gflow[n * 2 * plane + plane + pix] = (mx * gix) / dx * 2.0f;      // channel 1 displaces x (the flip of WP:105)
        gflow[n * 2 * plane + pix] = (my * giy) / dy * 2.0f;
    }
}

// scalar fallback (W % 4 != 0 or W < 2)
__global__ void __launch_bounds__(256) grid_warp_scalar_kernel(const float* __restrict__ second, const float* __restrict__ flow,
                                                               float* __restrict__ out, float* __restrict__ mask, int B, int C, int H, int W) {
    const size_t plane = (size_t)H * W, total = (size_t)B * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / plane, pix = i % plane;
        const int y = (int)(pix / W), x = (int)(pix % W);
        float ix, iy;
        grid_coords(flow[n * 2 * plane + pix], flow[n * 2 * plane + plane + pix], x, y, H, W, ix, iy);
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const float w[4] = {(fx0 + 1.0f - ix) * (fy0 + 1.0f - iy), (ix - fx0) * (fy0 + 1.0f - iy), (fx0 + 1.0f - ix) * (iy - fy0), (ix - fx0) * (iy - fy0)};
        const bool finite = fabsf(ix) < 1.0e9f && fabsf(iy) < 1.0e9f;
        const int x0 = finite ? (int)fx0 : -10, y0 = finite ? (int)fy0 : -10;
        bool inb[4];
        float msum = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cx = x0 + (k & 1), cy = y0 + (k >> 1);
            inb[k] = cx >= 0 && cx < W && cy >= 0 && cy < H;
            if (inb[k]) msum += w[k];
        }
        float m = msum;
        if (m < 0.999f) m = 0.0f;
        if (m > 0.0f) m = 1.0f;
        for (int c = 0; c < C; ++c) {
            const float* sp = second + (n * C + c) * plane;
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (inb[k]) acc += sp[(size_t)(y0 + (k >> 1)) * W + (x0 + (k & 1))] * w[k];
            out[(n * C + c) * plane + pix] = acc;
            if (mask) mask[(n * C + c) * plane + pix] = m;
        }
    }
}

__global__ void grid_warp_corners_kernel(const float* __restrict__ flow, int32_t* __restrict__ corners, int B, int H, int W) {
    const size_t plane = (size_t)H * W, total = (size_t)B * plane;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / plane, pix = i % plane;
        float ix, iy;
        grid_coords(flow[n * 2 * plane + pix], flow[n * 2 * plane + plane + pix], (int)(pix % W), (int)(pix / W), H, W, ix, iy);
        corners[2 * i] = floor_to_int(ix);
        corners[2 * i + 1] = floor_to_int(iy);
    }
}


// ================================================================================================
// splat_pyramid: ALL L*L offsets of a scale-L splat in one result.  T (B, C, L*Ho, L*Wo) with
//     T[n, c, L*cy + b, L*cx + a] = softsplat_out(in, flow, scale = L, offset = (a, b))[n, c, cy, cx]      (SS:352-423)
// flow_learner.py:159-206 evaluates its photometric loss on every offset of 10 levels: 1052 splats of the same image with the
// same flow.  For a plain pixel (pyr_plain) the scale-L bilinear weight of output cell (cx, a) is 1 - |L cx + a - fx| / L: the
// offsets sample one tent of half-width L on the full-resolution grid, and that tent is the scale-1 bilinear pair convolved with
// the discrete tent t(k) = 1 - |k| / L.  So
//     T = tent_L (*) splat_scale1(plain pixels)   +   the remaining (border) pixels, scattered with the reference's branches,
// one scale-1 splat + one separable 2L-1 tap filter per level instead of L*L splats.  The backward is the same identity
// transposed: G = tent_L (*) dT (zero-extended), then the scale-1 gradient kernels on the plain pixels; border pixels gather
// with the reference's own backward remaps (SS:515-533, 628-647) per offset.  Plain pixels differ from the reference only in
// summation order (and it rounds (fx - a) / L before forming the weights): ~1e-6 relative.
constexpr int PT_W = 64, PT_H = 16, PT_MAXL = 16, PT_HALO = PT_MAXL - 1;
constexpr int PT_IW = PT_W + 2 * PT_HALO, PT_IH = PT_H + 2 * PT_HALO;

// out (planes, Ho_, Wo_) (+)= tent_L (*) in (planes, Hi, Wi) along x (DX), y (DY) or both; zero outside the input; out index (y, x)
// reads in (y - ky, x - kx)
template <bool DX, bool DY, bool ACC>
__global__ void __launch_bounds__(256) tent_kernel(const float* __restrict__ in, float* __restrict__ out, int Hi, int Wi, int Ho_, int Wo_, int L) {
    __shared__ float tile[PT_IH][PT_IW + 1];
    __shared__ float hrow[PT_IH][PT_W + 1];
    const size_t plane_i = (size_t)Hi * Wi, plane_o = (size_t)Ho_ * Wo_;
    const float* ip = in + (size_t)blockIdx.z * plane_i;
    float* op = out + (size_t)blockIdx.z * plane_o;
    const int X0 = blockIdx.x * PT_W, Y0 = blockIdx.y * PT_H, hx = DX ? L - 1 : 0, hy = DY ? L - 1 : 0;
    const int iw = PT_W + 2 * hx, ih = PT_H + 2 * hy;
    for (int i = threadIdx.x; i < ih * iw; i += 256) {
        const int r = i / iw, c = i - r * iw;
        const int y = Y0 - hy + r, x = X0 - hx + c;
        tile[r][c] = (y >= 0 && y < Hi && x >= 0 && x < Wi) ? ip[(size_t)y * Wi + x] : 0.0f;
    }
    __syncthreads();
    const float inv = 1.0f / (float)L;
    for (int i = threadIdx.x; i < ih * PT_W; i += 256) {         // horizontal pass
        const int r = i / PT_W, c = i - r * PT_W;
        float acc = tile[r][c + hx];
        if (DX)
            for (int k = 1; k < L; ++k) acc += (1.0f - (float)k * inv) * (tile[r][c + hx - k] + tile[r][c + hx + k]);
        hrow[r][c] = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PT_H * PT_W; i += 256) {       // vertical pass
        const int r = i / PT_W, c = i - r * PT_W;
        const int y = Y0 + r, x = X0 + c;
        if (y >= Ho_ || x >= Wo_) continue;
        float acc = hrow[r + hy][c];
        if (DY)
            for (int k = 1; k < L; ++k) acc += (1.0f - (float)k * inv) * (hrow[r + hy - k][c] + hrow[r + hy + k][c]);
        if (ACC) op[(size_t)y * Wo_ + x] += acc;
        else op[(size_t)y * Wo_ + x] = acc;
    }
}

// Border classes of a level (class code in the top two bits of a list entry): a pixel whose target is plain along one axis
// still takes the tent path ALONG THAT AXIS -- its contribution factorises into (per-offset reference weights on the border axis)
// x (scale-1 bilinear pair on the plain axis, tent-filtered afterwards) -- so it costs O(L) scattered values instead of O(L^2).
constexpr unsigned PYR_IDX = 0x3fffffffu;       // class 0: both axes border; 1: x border, y plain; 2: x plain, y border
__device__ __forceinline__ bool pyr_plain_x(float fltX, int L, int W) { return fltX >= (float)(L - 1) && fltX < (float)W - 1.0f; }
__device__ __forceinline__ bool pyr_plain_y(float fltY, int L, int H) { return fltY >= (float)(L - 1) && fltY < (float)H - 1.0f; }

// compact list of the border pixels of level L (finite target, not plain): one atomic per wave
__global__ void __launch_bounds__(256) pyramid_border_list_kernel(const float* __restrict__ flow, unsigned int* __restrict__ list,
                                                                  unsigned int* __restrict__ count, int B, int H, int W, int L) {
    const size_t plane = (size_t)H * W, total = (size_t)B * plane;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x; i0 < total; i0 += stride) {      // whole waves stay in the loop (ballot)
        const size_t i = i0 + threadIdx.x;
        bool border = false;
        unsigned cls = 0;
        if (i < total) {
            const size_t n = i / plane, pix = i % plane;
            const int y = (int)(pix / W), x = (int)(pix % W);
            const float fltX = (float)x + flow[n * 2 * plane + pix], fltY = (float)y + flow[n * 2 * plane + plane + pix];
            border = isfinite(fltX) && isfinite(fltY) && !pyr_plain(fltX, fltY, L, H, W);
            cls = pyr_plain_y(fltY, L, H) ? 1u : (pyr_plain_x(fltX, L, W) ? 2u : 0u);
        }
        const unsigned long long m = __ballot(border);
        const int lane = threadIdx.x & 63;
        unsigned base = 0;
        if (lane == 0 && m) base = atomicAdd(count, (unsigned)__popcll(m));
        base = __shfl(base, 0, 64);
        if (border) list[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned)i | (cls << 30);
    }
}

// border pixels of level g.scale, one work item per (pixel, offset): the reference's forward remap, scattered into T
__global__ void __launch_bounds__(256) pyramid_border_fwd_kernel(const float* __restrict__ in, const float* __restrict__ flow, float* __restrict__ T,
                                                                 const unsigned int* __restrict__ list, const unsigned int* __restrict__ count,
                                                                 SplatGeom g, int all_classes) {
    const size_t plane = (size_t)g.H * g.W;
    const int L = g.scale, L2 = L * L, Wt = L * g.Wo;
    const size_t tplane = (size_t)(L * g.Ho) * Wt;
    const size_t items = (size_t)(*count) * L2;
    for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (size_t)gridDim.x * blockDim.x) {
        const unsigned e = list[it / L2];
        if (!all_classes && (e >> 30) != 0u) continue;          // strip pixels go through pyramid_strip_fwd_kernel
        const size_t i = e & PYR_IDX;
        const int o = (int)(it % L2), a = o % L, b = o / L;
        const int n = (int)(i / plane);
        const size_t pix = i % plane;
        const int y = (int)(pix / g.W), x = (int)(pix % g.W);
        SplatGeom go = g;
        go.ox = a; go.oy = b;
        float fx, fy, d0, d1;
        if (!splat_remap<0>(flow[(size_t)n * 2 * plane + pix], flow[(size_t)n * 2 * plane + plane + pix], x, y, go, fx, fy, d0, d1)) continue;
        const int x0 = floor_to_int(fx), y0 = floor_to_int(fy);
        float w[4];
        corner_weights(fx, fy, x0, y0, w);
        for (int k = 0; k < 4; ++k) {
            const int cx = x0 + (k & 1), cy = y0 + (k >> 1);
            if (cx < 0 || cx >= g.Wo || cy < 0 || cy >= g.Ho) continue;
            const size_t t = (size_t)(L * cy + b) * Wt + (L * cx + a);
            for (int c = 0; c < g.C; ++c)
                atomicAdd(&T[((size_t)n * g.C + c) * tplane + t], in[((size_t)n * g.C + c) * plane + pix] * w[k]);
        }
    }
}

// strip pixels (border along ONE axis), forward.  AX = 0: x border, y plain -> scatter into U (planes, H, L*Wo) whose columns are
// full-resolution offset positions X = L cx + a and whose rows are scale-1 rows (the y tent filter runs afterwards);
// AX = 1: x plain, y border -> into Bm (planes, L*Ho, W), filtered along x afterwards.  One work item per (pixel, offset).
template <int AX>
__global__ void __launch_bounds__(256) pyramid_strip_fwd_kernel(const float* __restrict__ in, const float* __restrict__ flow, float* __restrict__ dst,
                                                                const unsigned int* __restrict__ list, const unsigned int* __restrict__ count,
                                                                SplatGeom g) {
    const size_t plane = (size_t)g.H * g.W;
    const int L = g.scale, Wt = L * g.Wo, Ht = L * g.Ho;
    const size_t dplane = AX == 0 ? (size_t)g.H * Wt : (size_t)Ht * g.W;
    const size_t items = (size_t)(*count) * L;
    for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (size_t)gridDim.x * blockDim.x) {
        const unsigned e = list[it / L];
        if ((e >> 30) != (AX == 0 ? 1u : 2u)) continue;
        const size_t i = e & PYR_IDX;
        const int o = (int)(it % L);
        const int n = (int)(i / plane);
        const size_t pix = i % plane;
        const int y = (int)(pix / g.W), x = (int)(pix % g.W);
        const float f0 = flow[(size_t)n * 2 * plane + pix], f1 = flow[(size_t)n * 2 * plane + plane + pix];
        SplatGeom go = g;
        go.ox = AX == 0 ? o : 0; go.oy = AX == 0 ? 0 : o;
        float fx, fy, d0, d1;
        if (!splat_remap<0>(f0, f1, x, y, go, fx, fy, d0, d1)) continue;
        // border axis: the reference's cells and weights at this offset; plain axis: the scale-1 bilinear pair of the raw target
        const float fb = AX == 0 ? fx : fy;
        const int c0 = floor_to_int(fb);
        const float wb[2] = {(float)(c0 + 1) - fb, fb - (float)c0};
        const float fp = AX == 0 ? (float)y + f1 : (float)x + f0;
        const int p0 = (int)floorf(fp);
        const float wp[2] = {(float)(p0 + 1) - fp, fp - (float)p0};
        const int nb = AX == 0 ? g.Wo : g.Ho;
        for (int k = 0; k < 2; ++k) {
            const int cb = c0 + k;
            if (cb < 0 || cb >= nb) continue;
            const int full = L * cb + o;                            // full-resolution position along the border axis
            for (int j = 0; j < 2; ++j) {
                const size_t t = AX == 0 ? (size_t)(p0 + j) * Wt + full : (size_t)full * g.W + (p0 + j);
                const float w = AX == 0 ? wb[k] * wp[j] : wp[j] * wb[k];
                for (int c = 0; c < g.C; ++c)
                    atomicAdd(&dst[((size_t)n * g.C + c) * dplane + t], in[((size_t)n * g.C + c) * plane + pix] * w);
            }
        }
    }
}

// strip pixels, backward (the transposed per-axis identity).  With dP = dT filtered along the PLAIN axis (AX = 0: dU = tent_y dT,
// (planes, H, L*Wo); AX = 1: dBm = tent_x dT, (planes, L*Ho, W)), sums over the plain axis' offsets collapse:
//   sum_o sum_cells W_cell(o) dT[cell, o]   = sum_j w_j dP[p0 + j]          (bilinear pair w of the raw target)
//   sum_o sum_cells s_cell    dT[cell, o]   = L sum_j s_j dP[p0 + j]        (s = -1, +1: the derivative of that pair)
// and the reference's ingrad (SS:489-565) / flowgrad (SS:600-700) keep their own remaps (variants 1 and 2) on the border axis.  The
// flow gradient of channel 0 multiplies by the Y branch factor and that of channel 1 by the X one (SS:664-672): the plain axis'
// factor is 1 / L.  One work item per (pixel, offset of the border axis); results are added to the zeros the scale-1 kernels wrote.
template <int AX>
__global__ void __launch_bounds__(256) pyramid_strip_bwd_kernel(const float* __restrict__ in, const float* __restrict__ flow, const float* __restrict__ dP,
                                                                float* __restrict__ ingrad, float* __restrict__ flowgrad,
                                                                const unsigned int* __restrict__ list, const unsigned int* __restrict__ count, SplatGeom g) {
    const size_t plane = (size_t)g.H * g.W;
    const int L = g.scale, Wt = L * g.Wo, Ht = L * g.Ho;
    const size_t dplane = AX == 0 ? (size_t)g.H * Wt : (size_t)Ht * g.W;
    const size_t items = (size_t)(*count) * L;
    const int nb = AX == 0 ? g.Wo : g.Ho;
    for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (size_t)gridDim.x * blockDim.x) {
        const unsigned e = list[it / L];
        if ((e >> 30) != (AX == 0 ? 1u : 2u)) continue;
        const size_t i = e & PYR_IDX;
        const int o = (int)(it % L);
        const int n = (int)(i / plane);
        const size_t pix = i % plane;
        const int y = (int)(pix / g.W), x = (int)(pix % g.W);
        const float f0 = flow[(size_t)n * 2 * plane + pix], f1 = flow[(size_t)n * 2 * plane + plane + pix];
        SplatGeom go = g;
        go.ox = AX == 0 ? o : 0; go.oy = AX == 0 ? 0 : o;
        float fx, fy, dxx, dyy;
        // border axis, variant 1 (ingrad) and variant 2 (flowgrad)
        const bool ok1 = splat_remap<1>(f0, f1, x, y, go, fx, fy, dxx, dyy);
        const float b1 = AX == 0 ? fx : fy;
        const bool ok2 = splat_remap<2>(f0, f1, x, y, go, fx, fy, dxx, dyy);
        const float b2 = AX == 0 ? fx : fy, dfl = AX == 0 ? dxx : dyy;
        const int c1 = floor_to_int(b1), c2 = floor_to_int(b2);
        const float W1[2] = {(float)(c1 + 1) - b1, b1 - (float)c1}, W2[2] = {(float)(c2 + 1) - b2, b2 - (float)c2};
        // plain axis: the raw target's bilinear pair
        const float fp = AX == 0 ? (float)y + f1 : (float)x + f0;
        const int p0 = (int)floorf(fp);
        const float wp[2] = {(float)(p0 + 1) - fp, fp - (float)p0};
        auto at = [&](const float* base, int cell, int j) {       // dP at (border cell of this offset, plain position p0 + j)
            const int full = L * cell + o;
            return base[AX == 0 ? (size_t)(p0 + j) * Wt + full : (size_t)full * g.W + (p0 + j)];
        };
        float g_border = 0.0f, g_plain = 0.0f;     // d / d(flow of the border axis), d / d(flow of the plain axis)
        for (int c = 0; c < g.C; ++c) {
            const float* dp = dP + ((size_t)n * g.C + c) * dplane;
            const float v = in ? in[((size_t)n * g.C + c) * plane + pix] : 0.0f;
            if (ingrad && ok1) {
                float acc = 0.0f;
                for (int k = 0; k < 2; ++k)
                    if (c1 + k >= 0 && c1 + k < nb) acc += W1[k] * (wp[0] * at(dp, c1 + k, 0) + wp[1] * at(dp, c1 + k, 1));
                if (acc != 0.0f) atomicAdd(&ingrad[((size_t)n * g.C + c) * plane + pix], acc);
            }
            if (flowgrad && ok2) {
                for (int k = 0; k < 2; ++k)
                    if (c2 + k >= 0 && c2 + k < nb) {
                        const float d0 = at(dp, c2 + k, 0), d1 = at(dp, c2 + k, 1);
                        const float sgn = k == 0 ? -1.0f : 1.0f;
                        // border-axis flow: sign of the border cell x plain weights, times the PLAIN axis' branch factor 1 / L
                        g_border += v * sgn * (wp[0] * d0 + wp[1] * d1) * (1.0f / (float)L);
                        // plain-axis flow: border weights x plain derivative (L x the pair difference), times the BORDER axis' factor
                        g_plain += v * W2[k] * (d1 - d0) * (float)L * dfl;
                    }
            }
        }
        if (flowgrad) {
            float* gxp = flowgrad + (size_t)n * 2 * plane + pix;
            if (AX == 0) {
                if (g_border != 0.0f) atomicAdd(gxp, g_border);
                if (g_plain != 0.0f) atomicAdd(gxp + plane, g_plain);
            } else {
                if (g_plain != 0.0f) atomicAdd(gxp, g_plain);
                if (g_border != 0.0f) atomicAdd(gxp + plane, g_border);
            }
        }
    }
}

// border pixels, backward, one work item per (pixel, offset row b): the reference's ingrad (SS:489-565) and flowgrad (SS:600-700)
// over the offsets a of that row, added to the zeros the scale-1 kernels wrote for these pixels
__global__ void __launch_bounds__(256) pyramid_border_bwd_kernel(const float* __restrict__ in, const float* __restrict__ flow, const float* __restrict__ dT,
                                                                 float* __restrict__ ingrad, float* __restrict__ flowgrad,
                                                                 const unsigned int* __restrict__ list, const unsigned int* __restrict__ count, SplatGeom g,
                                                                 int all_classes) {
    const size_t plane = (size_t)g.H * g.W;
    const int L = g.scale, Wt = L * g.Wo;
    const size_t tplane = (size_t)(L * g.Ho) * Wt;
    const size_t items = (size_t)(*count) * L;
    for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (size_t)gridDim.x * blockDim.x) {
        if (!all_classes && (list[it / L] >> 30) != 0u) continue;          // strip pixels: pyramid_strip_bwd_kernel
        const size_t i = list[it / L] & PYR_IDX;
        const int b = (int)(it % L);
        const int n = (int)(i / plane);
        const size_t pix = i % plane;
        const int y = (int)(pix / g.W), x = (int)(pix % g.W);
        const float f0 = flow[(size_t)n * 2 * plane + pix], f1 = flow[(size_t)n * 2 * plane + plane + pix];
        SplatGeom go = g;
        go.oy = b;
        float gx = 0.0f, gy = 0.0f;
        for (int c = 0; c < g.C; ++c) {
            const float v = in ? in[((size_t)n * g.C + c) * plane + pix] : 0.0f;
            const float* gp = dT + ((size_t)n * g.C + c) * tplane;
            float acc = 0.0f;
            for (int a = 0; a < L; ++a) {
                go.ox = a;
                float fx, fy, dxx, dyy;
                if (ingrad && splat_remap<1>(f0, f1, x, y, go, fx, fy, dxx, dyy)) {
                    const int x0 = floor_to_int(fx), y0 = floor_to_int(fy);
                    float w[4];
                    corner_weights(fx, fy, x0, y0, w);
                    for (int k = 0; k < 4; ++k) {
                        const int cx = x0 + (k & 1), cy = y0 + (k >> 1);
                        if (cx >= 0 && cx < g.Wo && cy >= 0 && cy < g.Ho) acc += gp[(size_t)(L * cy + b) * Wt + (L * cx + a)] * w[k];
                    }
                }
                if (flowgrad && splat_remap<2>(f0, f1, x, y, go, fx, fy, dxx, dyy)) {
                    const int x0 = floor_to_int(fx), y0 = floor_to_int(fy);
                    const float x1 = (float)(x0 + 1), y1 = (float)(y0 + 1);
                    const float wx[4] = {-1.0f * (y1 - fy), +1.0f * (y1 - fy), -1.0f * (fy - (float)y0), +1.0f * (fy - (float)y0)};
                    const float wy[4] = {(x1 - fx) * -1.0f, (fx - (float)x0) * -1.0f, (x1 - fx) * +1.0f, (fx - (float)x0) * +1.0f};
                    for (int k = 0; k < 4; ++k) {
                        const int cx = x0 + (k & 1), cy = y0 + (k >> 1);
                        if (cx >= 0 && cx < g.Wo && cy >= 0 && cy < g.Ho) {
                            const float go_ = gp[(size_t)(L * cy + b) * Wt + (L * cx + a)];
                            gx += go_ * v * wx[k] * dyy;
                            gy += go_ * v * wy[k] * dxx;
                        }
                    }
                }
            }
            if (ingrad && acc != 0.0f) atomicAdd(&ingrad[((size_t)n * g.C + c) * plane + pix], acc);
        }
        if (flowgrad) {
            if (gx != 0.0f) atomicAdd(&flowgrad[(size_t)n * 2 * plane + pix], gx);
            if (gy != 0.0f) atomicAdd(&flowgrad[(size_t)n * 2 * plane + plane + pix], gy);
        }
    }
}

// ---- photometric loss of one pyramid level on the interleaved splats (flow_learner.py:176-190, WP:273-287) -------------------
// Tin, Ttg: (B, C+1, Ht, Wt) = splat_pyramid of cat(img e^m, e^m) with the predicted flow, and of cat(tgt e, e) with zero flow.
// Per position: filled = Tin_c / (w + 1e-7) where w > 0 else NaN (fill_holes_nan), tgt = Ttg_c / (w_t + 1e-7) (soft mode's
// normalisation), Charbonnier penalty sqrt(d^2 + 1e-6) over the pairs without NaN, accumulated per offset (a, b) = (X mod L,
// Y mod L): sums / counts -> nan_charbonnier of every offset.  One kernel instead of ~20 elementwise passes per level.
__global__ void __launch_bounds__(256) pyr_charb_reduce_kernel(const float* __restrict__ Tin, const float* __restrict__ Ttg, double* __restrict__ sums,
                                                               double* __restrict__ counts, int B, int C, int Ht, int Wt, int L) {
    __shared__ float bs[PT_MAXL], bc[PT_MAXL];
    const size_t tplane = (size_t)Ht * Wt;
    const int xchunks = (Wt + 255) / 256;
    const size_t nrows = (size_t)B * Ht * xchunks;
    for (size_t rw = blockIdx.x; rw < nrows; rw += gridDim.x) {
        const int xc = (int)(rw % xchunks);
        const size_t ny = rw / xchunks;
        const int Y = (int)(ny % Ht), n = (int)(ny / Ht);
        const int X = xc * 256 + threadIdx.x;
        if (threadIdx.x < L) { bs[threadIdx.x] = 0.0f; bc[threadIdx.x] = 0.0f; }
        __syncthreads();
        if (X < Wt) {
            const size_t pos = (size_t)Y * Wt + X;
            const float* pi = Tin + (size_t)n * (C + 1) * tplane + pos;
            const float* pt = Ttg + (size_t)n * (C + 1) * tplane + pos;
            const float wi = pi[(size_t)C * tplane], wt = pt[(size_t)C * tplane];
            float s_ = 0.0f, c_ = 0.0f;
            for (int c = 0; c < C; ++c) {
                const float filled = wi > 0.0f ? pi[(size_t)c * tplane] / (wi + 0.0000001f) : __builtin_nanf("");
                const float tg = pt[(size_t)c * tplane] / (wt + 0.0000001f);
                if (filled == filled && tg == tg) {
                    const float d = tg - filled;
                    s_ += sqrtf(d * d + 1.0e-6f);
                    c_ += 1.0f;
                }
            }
            if (c_ > 0.0f) { atomicAdd(&bs[X % L], s_); atomicAdd(&bc[X % L], c_); }
        }
        __syncthreads();
        if (threadIdx.x < L && bc[threadIdx.x] > 0.0f) {
            const int o = (Y % L) * L + threadIdx.x;            // [b][a]
            atomicAdd(&sums[o], (double)bs[threadIdx.x]);
            atomicAdd(&counts[o], (double)bc[threadIdx.x]);
        }
        __syncthreads();
    }
}

// d(level loss) / dTin, level loss = mean over offsets of sums / counts; gscale[0] = the incoming gradient of the level loss
__global__ void __launch_bounds__(256) pyr_charb_grad_kernel(const float* __restrict__ Tin, const float* __restrict__ Ttg, const double* __restrict__ counts,
                                                             const float* __restrict__ gscale, float* __restrict__ dTin, int B, int C, int Ht, int Wt,
                                                             int L) {
    const size_t tplane = (size_t)Ht * Wt, total = (size_t)B * tplane;
    const float gs = gscale[0] / (float)(L * L);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / tplane);
        const size_t pos = i % tplane;
        const int Y = (int)(pos / Wt), X = (int)(pos % Wt);
        const float* pi = Tin + (size_t)n * (C + 1) * tplane + pos;
        const float* pt = Ttg + (size_t)n * (C + 1) * tplane + pos;
        float* po = dTin + (size_t)n * (C + 1) * tplane + pos;
        const float wi = pi[(size_t)C * tplane], wt = pt[(size_t)C * tplane];
        const double cnt = counts[(Y % L) * L + (X % L)];
        const float k = cnt > 0.0 ? gs / (float)cnt : 0.0f;
        const float inv = 1.0f / (wi + 0.0000001f);
        float dw = 0.0f;
        for (int c = 0; c < C; ++c) {
            float g = 0.0f;
            if (wi > 0.0f) {
                const float ti = pi[(size_t)c * tplane];
                const float filled = ti * inv, tg = pt[(size_t)c * tplane] / (wt + 0.0000001f);
                if (filled == filled && tg == tg) {
                    const float d = filled - tg;
                    const float df = k * d / sqrtf(d * d + 1.0e-6f);       // d loss / d filled
                    g = df * inv;
                    dw -= df * ti * inv * inv;
                }
            }
            po[(size_t)c * tplane] = g;
        }
        po[(size_t)C * tplane] = dw;
    }
}

static int make_geom(SplatGeom& g, int B, int C, int H, int W, int scale, int ox, int oy, int radius) {
    OFD_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0, "splat: bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
    OFD_CHECK_ARG(scale >= 1 && H / scale > 0 && W / scale > 0, "splat: bad scale %d for %dx%d", scale, H, W);
    OFD_CHECK_ARG(ox >= 0 && oy >= 0 && ox < scale && oy < scale, "splat: offset (%d,%d) must be in [0,scale)", ox, oy);
    OFD_CHECK_ARG((size_t)B * H * W < (1ull << 31), "splat: B*H*W must be < 2^31");
    g = SplatGeom{B, C, H, W, H / scale, W / scale, scale, ox, oy, radius < 0 ? 0 : radius, 0, 0, S_TW, S_TH, 0, 0};
    // coarser scales: smaller output tiles, so that a tile's source footprint stays ~64 .. 128 pixels wide and the level has enough tiles
    // to fill the chip (at scale 16 a 448 x 1024 image is ONE 64 x 64 tile per sample)
    static const int small_tiles = getenv("OFD_SPLAT_SMALL_TILES") ? atoi(getenv("OFD_SPLAT_SMALL_TILES")) : 1;
    if (scale > 1 && small_tiles) { int t = S_TW / scale; if (t < 8) t = 8; g.tw = t < S_TW ? t : S_TW; g.th = t < S_TH ? t : S_TH; }
    g.ntx = cdiv(g.Wo, g.tw);
    g.nty = cdiv(g.Ho, g.th);
    return OFD_OK;
}

static inline int stream_grid(size_t total, int block) {
    size_t b = (total + block - 1) / block;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace ofd

using namespace ofd;

constexpr int S_MAXC = 256;     // channels the per-plane maxima area of the workspace is sized for
static int splat_launch(const float* in, const float* flow, float* out, const SplatGeom& g, void* workspace, hipStream_t s);

extern "C" size_t ofd_splat_workspace_bytes(int B, int H, int W) {
    // far-corner counter | per-(sample, channel) |in| maxima | far-corner list
    return 16 + (size_t)B * S_MAXC * 4 + sizeof(unsigned long long) * (size_t)B * H * W;
}

extern "C" int ofd_splat_fwd(const float* in, const float* flow, float* out, int B, int C, int H, int W, int scale,
                             int offset_x, int offset_y, int radius, void* workspace, size_t workspace_bytes, void* stream) {
    SplatGeom g{};
    int rc = make_geom(g, B, C, H, W, scale, offset_x, offset_y, radius);
    if (rc) return rc;
    OFD_CHECK_ARG(in && flow && out && workspace, "splat_fwd: null pointer");
    OFD_CHECK_ARG(g.nty <= 65535 && B <= 65535, "splat_fwd: grid too large");
    if (workspace_bytes < ofd_splat_workspace_bytes(B, H, W)) {
        set_error("splat_fwd: workspace %zu < %zu", workspace_bytes, ofd_splat_workspace_bytes(B, H, W));
        return OFD_ERR_WORKSPACE;
    }
    OFD_CHECK_ARG(C <= S_MAXC, "splat_fwd: C=%d > %d", C, S_MAXC);
    return splat_launch(in, flow, out, g, workspace, (hipStream_t)stream);
}

static int splat_launch(const float* in, const float* flow, float* out, const SplatGeom& g, void* workspace, hipStream_t s) {
    const int B = g.B, C = g.C, H = g.H, W = g.W;
    unsigned int* count = (unsigned int*)workspace;
    unsigned int* absmax = (unsigned int*)((char*)workspace + 16);
    unsigned long long* list = (unsigned long long*)((char*)workspace + 16 + (size_t)B * S_MAXC * 4);
    const unsigned cap = (unsigned)((size_t)B * H * W);
    static bool attr = false;
    if (!attr) {
        OFD_HIP(hipFuncSetAttribute((const void*)splat_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES));
        OFD_HIP(hipFuncSetAttribute((const void*)splat_tile_fast_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SF_LDS_BYTES));
        attr = true;
    }
    static int no_fast = -1;
    if (no_fast < 0) { const char* e = getenv("OFD_SPLAT_NO_FAST"); no_fast = (e && atoi(e)) ? 1 : 0; }
    // fast path: identity remap and a window that fits the 7-bit list coordinates (radius <= 32)
    const bool fast = g.scale == 1 && g.ox == 0 && g.oy == 0 && !g.grid && g.pyr_L == 0 && g.radius >= 3 && S_TW + 2 * g.radius <= 128 && !no_fast;
    {
        OFD_HIP(hipMemsetAsync(count, 0, 16 + (size_t)B * C * 4, s));
        const size_t plane = (size_t)H * W;
        int gx = (int)((plane / 4 + 255) / 256);
        gx = gx < 1 ? 1 : (gx > 24 ? 24 : gx);
        splat_absmax_kernel<<<dim3(gx, B * C), 256, 0, s>>>(in, absmax, plane);
    }
    for (int c0 = 0; c0 < C; c0 += S_CG) {
        const int cg = (C - c0 < S_CG) ? (C - c0) : S_CG;
        if (fast) splat_tile_fast_kernel<<<g.ntx * g.nty * B, S_NT, SF_LDS_BYTES, s>>>(in, flow, out, list, count, cap, absmax, g, c0, cg);
        else splat_tile_kernel<<<dim3(g.ntx, g.nty, B), S_NT, (size_t)g.tw * g.th * (S_CG * 8 + 4), s>>>(in, flow, out, list, count, cap, absmax, g, c0, cg);
    }
    splat_far_kernel<<<256, 256, 0, s>>>(in, flow, out, list, count, cap, g);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

extern "C" size_t ofd_splat_pyramid_workspace_bytes(int B, int C, int H, int W) {
    // the splat's own workspace | three (B, C, H, W) fp32 images (forward: scale-1 splat of the plain pixels, the two strip
    // buffers; backward: the filtered gradient)
    return (ofd_splat_workspace_bytes(B, H, W) + 255) / 256 * 256 + 3 * (((size_t)B * C * H * W * 4 + 255) / 256 * 256);
}

extern "C" int ofd_splat_pyramid_fwd(const float* in, const float* flow, float* T, int B, int C, int H, int W, int L, int radius,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    OFD_CHECK_ARG(in && flow && T && workspace, "splat_pyramid_fwd: null pointer");
    OFD_CHECK_ARG(L >= 1 && L <= PT_MAXL, "splat_pyramid_fwd: level %d (1..%d)", L, PT_MAXL);
    if (workspace_bytes < ofd_splat_pyramid_workspace_bytes(B, C, H, W)) {
        set_error("splat_pyramid_fwd: workspace %zu < %zu", workspace_bytes, ofd_splat_pyramid_workspace_bytes(B, C, H, W));
        return OFD_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    SplatGeom g1, gL;
    int rc = make_geom(g1, B, C, H, W, 1, 0, 0, radius);
    if (rc) return rc;
    rc = make_geom(gL, B, C, H, W, L, 0, 0, 0);
    if (rc) return rc;
    OFD_CHECK_ARG(g1.nty <= 65535 && B <= 65535 && C <= S_MAXC && (size_t)B * C <= 65535, "splat_pyramid_fwd: grid too large");
    if (L == 1) return splat_launch(in, flow, T, g1, workspace, s);
    float* S = (float*)((char*)workspace + (ofd_splat_workspace_bytes(B, H, W) + 255) / 256 * 256);
    g1.pyr_L = L;
    rc = splat_launch(in, flow, S, g1, workspace, s);
    if (rc) return rc;
    const int Ht = L * gL.Ho, Wt = L * gL.Wo;
    OFD_CHECK_ARG((size_t)B * H * W < (1u << 30), "splat_pyramid_fwd: B*H*W must be < 2^30");
    const size_t img = ((size_t)B * C * H * W * 4 + 255) / 256 * 256;
    float* U = (float*)((char*)S + img);            // (planes, H, Wt): x-filtered plain pixels + x-border strip pixels
    float* Bm = (float*)((char*)S + 2 * img);       // (planes, Ht, W): y-border strip pixels, to be filtered along x
    // border pixels: compact list in the (now idle) far-corner list area of the splat workspace, counter in its header
    unsigned int* bcount = (unsigned int*)workspace + 2;
    unsigned int* blist = (unsigned int*)((char*)workspace + 16 + (size_t)B * S_MAXC * 4);
    OFD_HIP(hipMemsetAsync(bcount, 0, 4, s));
    pyramid_border_list_kernel<<<stream_grid((size_t)B * H * W, 256), 256, 0, s>>>(flow, blist, bcount, B, H, W, L);
    static const bool direct = getenv("OFD_PYR_DIRECT_BORDER") && atoi(getenv("OFD_PYR_DIRECT_BORDER"));     // A/B: L*L scatter for every class
    if (direct) {
        tent_kernel<true, true, false><<<dim3(cdiv(Wt, PT_W), cdiv(Ht, PT_H), B * C), 256, 0, s>>>(S, T, H, W, Ht, Wt, L);
        pyramid_border_fwd_kernel<<<4096, 256, 0, s>>>(in, flow, T, blist, bcount, gL, 1);
    } else {
        // T = tent_y( tent_x(S) + [x-border strips] ) + tent_x( [y-border strips] ) + [corner pixels]
        OFD_HIP(hipMemsetAsync(Bm, 0, (size_t)B * C * Ht * W * 4, s));
        pyramid_strip_fwd_kernel<1><<<2048, 256, 0, s>>>(in, flow, Bm, blist, bcount, gL);
        tent_kernel<true, false, false><<<dim3(cdiv(Wt, PT_W), cdiv(Ht, PT_H), B * C), 256, 0, s>>>(Bm, T, Ht, W, Ht, Wt, L);
        tent_kernel<true, false, false><<<dim3(cdiv(Wt, PT_W), cdiv(H, PT_H), B * C), 256, 0, s>>>(S, U, H, W, H, Wt, L);
        pyramid_strip_fwd_kernel<0><<<2048, 256, 0, s>>>(in, flow, U, blist, bcount, gL);
        tent_kernel<false, true, true><<<dim3(cdiv(Wt, PT_W), cdiv(Ht, PT_H), B * C), 256, 0, s>>>(U, T, H, Wt, Ht, Wt, L);
        pyramid_border_fwd_kernel<<<4096, 256, 0, s>>>(in, flow, T, blist, bcount, gL, 0);
    }
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

extern "C" int ofd_splat_pyramid_bwd(const float* in, const float* flow, const float* dT, float* ingrad, float* flowgrad, int B, int C, int H,
                                     int W, int L, void* workspace, size_t workspace_bytes, void* stream) {
    OFD_CHECK_ARG(flow && dT && (ingrad || flowgrad) && workspace, "splat_pyramid_bwd: null pointer");
    OFD_CHECK_ARG(!flowgrad || in, "splat_pyramid_bwd: the flow gradient needs the input");
    OFD_CHECK_ARG(L >= 1 && L <= PT_MAXL, "splat_pyramid_bwd: level %d (1..%d)", L, PT_MAXL);
    if (workspace_bytes < ofd_splat_pyramid_workspace_bytes(B, C, H, W)) {
        set_error("splat_pyramid_bwd: workspace %zu < %zu", workspace_bytes, ofd_splat_pyramid_workspace_bytes(B, C, H, W));
        return OFD_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    SplatGeom g1, gL;
    int rc = make_geom(g1, B, C, H, W, 1, 0, 0, 0);
    if (rc) return rc;
    rc = make_geom(gL, B, C, H, W, L, 0, 0, 0);
    if (rc) return rc;
    OFD_CHECK_ARG((size_t)B * C <= 65535, "splat_pyramid_bwd: grid too large");
    const float* G = dT;
    if (L > 1) {
        float* Gb = (float*)((char*)workspace + (ofd_splat_workspace_bytes(B, H, W) + 255) / 256 * 256);
        tent_kernel<true, true, false><<<dim3(cdiv(W, PT_W), cdiv(H, PT_H), B * C), 256, 0, s>>>(dT, Gb, L * gL.Ho, L * gL.Wo, H, W, L);
        G = Gb;
        g1.pyr_L = L;
    }
    const int grid = stream_grid((size_t)B * H * W, 256);
    if (ingrad) splat_ingrad_kernel<<<grid, 256, 0, s>>>(flow, G, ingrad, g1);
    if (flowgrad) splat_flowgrad_kernel<<<grid, 256, 0, s>>>(in, flow, G, flowgrad, g1);
    if (L > 1) {
        unsigned int* bcount = (unsigned int*)workspace + 2;
        unsigned int* blist = (unsigned int*)((char*)workspace + 16 + (size_t)B * S_MAXC * 4);
        OFD_HIP(hipMemsetAsync(bcount, 0, 4, s));
        pyramid_border_list_kernel<<<grid, 256, 0, s>>>(flow, blist, bcount, B, H, W, L);
        static const bool direct = getenv("OFD_PYR_DIRECT_BORDER") && atoi(getenv("OFD_PYR_DIRECT_BORDER"));
        if (direct) {
            pyramid_border_bwd_kernel<<<4096, 256, 0, s>>>(in, flow, dT, ingrad, flowgrad, blist, bcount, gL, 1);
        } else {
            const int Ht = L * gL.Ho, Wt = L * gL.Wo;
            const size_t img = ((size_t)B * C * H * W * 4 + 255) / 256 * 256;
            float* dU = (float*)((char*)G + img);          // tent_y dT: (planes, H, Wt)
            float* dBm = (float*)((char*)G + 2 * img);     // tent_x dT: (planes, Ht, W)
            tent_kernel<false, true, false><<<dim3(cdiv(Wt, PT_W), cdiv(H, PT_H), B * C), 256, 0, s>>>(dT, dU, Ht, Wt, H, Wt, L);
            tent_kernel<true, false, false><<<dim3(cdiv(W, PT_W), cdiv(Ht, PT_H), B * C), 256, 0, s>>>(dT, dBm, Ht, Wt, Ht, W, L);
            pyramid_strip_bwd_kernel<0><<<2048, 256, 0, s>>>(in, flow, dU, ingrad, flowgrad, blist, bcount, gL);
            pyramid_strip_bwd_kernel<1><<<2048, 256, 0, s>>>(in, flow, dBm, ingrad, flowgrad, blist, bcount, gL);
            pyramid_border_bwd_kernel<<<4096, 256, 0, s>>>(in, flow, dT, ingrad, flowgrad, blist, bcount, gL, 0);
        }
    }
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

extern "C" int ofd_pyramid_charbonnier_fwd(const float* Tin, const float* Ttg, double* sums, double* counts, int B, int C, int Ht, int Wt, int L,
                                           void* stream) {
    OFD_CHECK_ARG(Tin && Ttg && sums && counts && B > 0 && C > 0 && Ht > 0 && Wt > 0 && L >= 1 && L <= PT_MAXL, "pyramid_charbonnier_fwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    OFD_HIP(hipMemsetAsync(sums, 0, (size_t)L * L * sizeof(double), s));
    OFD_HIP(hipMemsetAsync(counts, 0, (size_t)L * L * sizeof(double), s));
    const size_t nrows = (size_t)B * Ht * cdiv(Wt, 256);
    pyr_charb_reduce_kernel<<<(unsigned)(nrows < 8192 ? nrows : 8192), 256, 0, s>>>(Tin, Ttg, sums, counts, B, C, Ht, Wt, L);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

extern "C" int ofd_pyramid_charbonnier_bwd(const float* Tin, const float* Ttg, const double* counts, const float* gscale, float* dTin, int B, int C,
                                           int Ht, int Wt, int L, void* stream) {
    OFD_CHECK_ARG(Tin && Ttg && counts && gscale && dTin && B > 0 && C > 0 && L >= 1 && L <= PT_MAXL, "pyramid_charbonnier_bwd: bad argument");
    pyr_charb_grad_kernel<<<stream_grid((size_t)B * Ht * Wt, 256), 256, 0, (hipStream_t)stream>>>(Tin, Ttg, counts, gscale, dTin, B, C, Ht, Wt, L);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

extern "C" int ofd_splat_corners(const float* flow, int32_t* corners, int B, int H, int W, int scale, int offset_x,
                                 int offset_y, void* stream) {
    SplatGeom g{};
    int rc = make_geom(g, B, 1, H, W, scale, offset_x, offset_y, 0);
    if (rc) return rc;
    OFD_CHECK_ARG(flow && corners, "splat_corners: null pointer");
    splat_corners_kernel<<<stream_grid((size_t)B * H * W, 256), 256, 0, (hipStream_t)stream>>>(flow, corners, g);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

extern "C" int ofd_splat_bwd_in(const float* flow, const float* outgrad, float* ingrad, int B, int C, int H, int W,
                                int scale, int offset_x, int offset_y, void* stream) {
    SplatGeom g{};
    int rc = make_geom(g, B, C, H, W, scale, offset_x, offset_y, 0);
    if (rc) return rc;
    OFD_CHECK_ARG(flow && outgrad && ingrad, "splat_bwd_in: null pointer");
    splat_ingrad_kernel<<<stream_grid((size_t)B * H * W, 256), 256, 0, (hipStream_t)stream>>>(flow, outgrad, ingrad, g);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

extern "C" int ofd_splat_bwd_flow(const float* in, const float* flow, const float* outgrad, float* flowgrad, int B,
                                  int C, int H, int W, int scale, int offset_x, int offset_y, void* stream) {
    SplatGeom g{};
    int rc = make_geom(g, B, C, H, W, scale, offset_x, offset_y, 0);
    if (rc) return rc;
    OFD_CHECK_ARG(in && flow && outgrad && flowgrad, "splat_bwd_flow: null pointer");
    splat_flowgrad_kernel<<<stream_grid((size_t)B * H * W, 256), 256, 0, (hipStream_t)stream>>>(in, flow, outgrad, flowgrad, g);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

extern "C" int ofd_warp_prep(const float* first, float* ten_in, int B, int C, int H, int W, int square, void* stream) {
    OFD_CHECK_ARG(first && ten_in && B > 0 && C > 0 && H > 0 && W > 0, "warp_prep: bad argument");
    warp_prep_kernel<<<stream_grid((size_t)B * H * W, 256), 256, 0, (hipStream_t)stream>>>(first, ten_in, B, C, (size_t)H * W, square);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

extern "C" int ofd_warp_holes(const float* splat, float* img, int B, int C, int Ho, int Wo, int mode, int set_nans, void* stream) {
    OFD_CHECK_ARG(splat && img && B > 0 && C > 0 && Ho > 0 && Wo > 0, "warp_holes: bad argument");
    OFD_CHECK_ARG(mode == 0 || mode == 1, "warp_holes: mode must be 0 (linear_unn) or 1 (linear)");
    warp_holes_kernel<<<stream_grid((size_t)B * Ho * Wo, 256), 256, 0, (hipStream_t)stream>>>(splat, img, B, C, (size_t)Ho * Wo, mode, set_nans);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

extern "C" int ofd_grid_warp_fwd(const float* second, const float* flow, float* out, float* mask, int B, int C, int H,
                                 int W, void* stream) {
    OFD_CHECK_ARG(second && flow && out && B > 0 && C > 0 && H > 0 && W > 0, "grid_warp_fwd: bad argument");
    static const bool no_tile = getenv("OFD_GW_TILE") && atoi(getenv("OFD_GW_TILE")) == 0;
    if (W % 4 == 0 && W >= 4 && !no_tile && (long)H * W < (1L << 30)) {
        const int tx = cdiv(W, GT_W), ty = cdiv(H, GT_H), gridn = B * tx * ty < 256 ? B * tx * ty : 256;   // one persistent workgroup per CU
        hipStream_t s_ = (hipStream_t)stream;
        static bool attr = false;
        if (!attr) {
            OFD_HIP(hipFuncSetAttribute((const void*)grid_warp_tile_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, GT_LDS_BYTES));
            OFD_HIP(hipFuncSetAttribute((const void*)grid_warp_tile_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, GT_LDS_BYTES));
            OFD_HIP(hipFuncSetAttribute((const void*)grid_warp_tile_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, GT_LDS_BYTES));
            OFD_HIP(hipFuncSetAttribute((const void*)grid_warp_tile_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, GT_LDS_BYTES));
            attr = true;
        }
        // OFD_GW_RING (opt-in, all three measure 72-75 us: the kernel is instruction-issue-bound, profiles/r03_pmc_grid_warp_ring.json): 2 = two 512-thread
        // workgroups per CU on 32 x 64 tiles, 1 = one 1024-thread workgroup on 64 x 64 tiles,
        // 0 (default) = the tile kernel above
        static const int ring = getenv("OFD_GW_RING") ? atoi(getenv("OFD_GW_RING")) : 0;
        static const int gdbg = getenv("OFD_GW_DBG") ? atoi(getenv("OFD_GW_DBG")) : 0;      // timing ablations only
        // band form (grid_warp_band_kernel): C = 3 at sizes where a launch fills the chip; OFD_GW_BAND=0 restores the tile kernel
        const int band = getenv("OFD_GW_BAND") ? atoi(getenv("OFD_GW_BAND")) : 1;        // (read per call: the tests compare the two forms in one process)
        if (C == 3 && band && ring == 0 && W >= GB_W && H >= 4 * GB_TH && (long)H * W < (1L << 29)) {
            static bool battr = false;
            if (!battr) {
                OFD_HIP(hipFuncSetAttribute((const void*)grid_warp_band_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, GB_LDS_BYTES));
                OFD_HIP(hipFuncSetAttribute((const void*)grid_warp_band_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, GB_LDS_BYTES));
                battr = true;
            }
            // segments: enough workgroups for one per CU, rows per segment a multiple of the step; a segment re-fetches 55 warm-up rows
            const int bands = cdiv(W, GB_W);
            static const int seg_env = getenv("OFD_GW_BAND_NSEG") ? atoi(getenv("OFD_GW_BAND_NSEG")) : 0;
            int nseg = seg_env > 0 ? seg_env : cdiv(256, B * bands);
            const int max_seg = H / (8 * GB_TH) > 0 ? H / (8 * GB_TH) : 1;                  // at least 64 rows per segment
            if (nseg > max_seg) nseg = max_seg;
            if (nseg < 1) nseg = 1;
            const int seg_rows = cdiv(cdiv(H, nseg), GB_TH) * GB_TH;
            nseg = cdiv(H, seg_rows);
            const int gridb = B * bands * nseg;
            if (mask) grid_warp_band_kernel<true><<<gridb, GB_THREADS, GB_LDS_BYTES, s_>>>(second, flow, out, mask, B, H, W, bands, nseg, seg_rows);
            else grid_warp_band_kernel<false><<<gridb, GB_THREADS, GB_LDS_BYTES, s_>>>(second, flow, out, mask, B, H, W, bands, nseg, seg_rows);
        }
        else if (C == 3 && ring == 2) {
            using R = GwRing<32, 2>;
            static bool rattr = false;
            if (!rattr) {
                OFD_HIP(hipFuncSetAttribute((const void*)grid_warp_ring_kernel<true, 32, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, R::LDS_BYTES));
                OFD_HIP(hipFuncSetAttribute((const void*)grid_warp_ring_kernel<false, 32, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, R::LDS_BYTES));
                rattr = true;
            }
            const int ty2 = cdiv(H, 32), nt2 = B * tx * ty2, grid2 = nt2 < 512 ? nt2 : 512;
            if (mask) grid_warp_ring_kernel<true, 32, 2><<<grid2, R::THREADS, R::LDS_BYTES, s_>>>(second, flow, out, mask, B, H, W, tx, ty2, gdbg);
            else grid_warp_ring_kernel<false, 32, 2><<<grid2, R::THREADS, R::LDS_BYTES, s_>>>(second, flow, out, mask, B, H, W, tx, ty2, gdbg);
        }
        else if (C == 3 && ring == 1) {
            using R = GwRing<64, 3>;
            static bool rattr = false;
            if (!rattr) {
                OFD_HIP(hipFuncSetAttribute((const void*)grid_warp_ring_kernel<true, 64, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, R::LDS_BYTES));
                OFD_HIP(hipFuncSetAttribute((const void*)grid_warp_ring_kernel<false, 64, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, R::LDS_BYTES));
                rattr = true;
            }
            if (mask) grid_warp_ring_kernel<true, 64, 3><<<gridn, R::THREADS, R::LDS_BYTES, s_>>>(second, flow, out, mask, B, H, W, tx, ty, gdbg);
            else grid_warp_ring_kernel<false, 64, 3><<<gridn, R::THREADS, R::LDS_BYTES, s_>>>(second, flow, out, mask, B, H, W, tx, ty, gdbg);
        }
        else if (C == 3) grid_warp_tile_kernel<3><<<gridn, GT_THREADS, GT_LDS_BYTES, s_>>>(second, flow, out, mask, B, C, H, W, tx, ty);
        else if (C == 1) grid_warp_tile_kernel<1><<<gridn, GT_THREADS, GT_LDS_BYTES, s_>>>(second, flow, out, mask, B, C, H, W, tx, ty);
        else if (C == 2) grid_warp_tile_kernel<2><<<gridn, GT_THREADS, GT_LDS_BYTES, s_>>>(second, flow, out, mask, B, C, H, W, tx, ty);
        else grid_warp_tile_kernel<0><<<gridn, GT_THREADS, GT_LDS_BYTES, s_>>>(second, flow, out, mask, B, C, H, W, tx, ty);
    } else if (W % 4 == 0 && W >= 4) {
        const int gridn = stream_grid((size_t)B * H * W / 4, 256);
        hipStream_t s_ = (hipStream_t)stream;
        if (C == 3) grid_warp_kernel<3><<<gridn, 256, 0, s_>>>(second, flow, out, mask, B, C, H, W);
        else if (C == 1) grid_warp_kernel<1><<<gridn, 256, 0, s_>>>(second, flow, out, mask, B, C, H, W);
        else if (C == 2) grid_warp_kernel<2><<<gridn, 256, 0, s_>>>(second, flow, out, mask, B, C, H, W);
        else grid_warp_kernel<0><<<gridn, 256, 0, s_>>>(second, flow, out, mask, B, C, H, W);
    }
    else
        grid_warp_scalar_kernel<<<stream_grid((size_t)B * H * W, 256), 256, 0, (hipStream_t)stream>>>(second, flow, out, mask, B, C, H, W);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

extern "C" int ofd_grid_warp_bwd(const float* second, const float* flow, const float* grad_out, float* grad_second, float* grad_flow,
                                 int B, int C, int H, int W, int radius, void* workspace, size_t workspace_bytes, void* stream) {
    OFD_CHECK_ARG(flow && grad_out && (grad_second || grad_flow) && B > 0 && C > 0 && H > 0 && W > 0, "grid_warp_bwd: bad argument");
    OFD_CHECK_ARG(!grad_flow || second, "grid_warp_bwd: the flow gradient needs the image");
    hipStream_t s = (hipStream_t)stream;
    if (grad_second) {
        // adjoint of the bilinear gather = bilinear scatter of grad_out to the same four corners: the splat kernel with
        // grid_sample's coordinates (bit-identical corner indices to ofd_grid_warp_fwd)
        SplatGeom g{};
        int rc = make_geom(g, B, C, H, W, 1, 0, 0, radius);
        if (rc) return rc;
        g.grid = 1;
        OFD_CHECK_ARG(workspace && g.nty <= 65535 && B <= 65535 && C <= S_MAXC, "grid_warp_bwd: workspace / grid");
        if (workspace_bytes < ofd_splat_workspace_bytes(B, H, W)) {
            set_error("grid_warp_bwd: workspace %zu < %zu", workspace_bytes, ofd_splat_workspace_bytes(B, H, W));
            return OFD_ERR_WORKSPACE;
        }
        rc = splat_launch(grad_out, flow, grad_second, g, workspace, s);
        if (rc) return rc;
    }
    if (grad_flow) {
        grid_warp_flowgrad_kernel<<<stream_grid((size_t)B * H * W, 256), 256, 0, s>>>(second, flow, grad_out, grad_flow, B, C, H, W);
        OFD_LAUNCH_CHECK();
    }
    return OFD_OK;
}

extern "C" int ofd_grid_warp_corners(const float* flow, int32_t* corners, int B, int H, int W, void* stream) {
    OFD_CHECK_ARG(flow && corners && B > 0 && H > 0 && W > 0, "grid_warp_corners: bad argument");
    grid_warp_corners_kernel<<<stream_grid((size_t)B * H * W, 256), 256, 0, (hipStream_t)stream>>>(flow, corners, B, H, W);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}
