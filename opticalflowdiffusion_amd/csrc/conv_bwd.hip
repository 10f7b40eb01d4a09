// Backward of the implicit-GEMM convolutions (what autograd runs for F.conv2d in the reference's
// training step, flow_diffuser.py:218-235 -> denoising_diffusion.py:823-891):
//   * data gradient  = the FORWARD kernel (conv_igemm.hip) on dY with tap-flipped, in/out
//     transposed weights (wt_transpose_kernel); concat / up-sample / unshuffle adjoints are the
//     grad_scatter kernels (slice, 2x2 sum-pool, pixel shuffle);
//   * weight gradient = conv_wgrad_kernel: dW[tap][ci][co] = sum_pixels X[p+tap][ci] dY[p][co], an MFMA
//     GEMM whose contraction index is the PIXEL.  Both operands live pixel-major in LDS (NHWC
//     rows) and are read with ds_read_b64_tr_b16, the transposing LDS read, so no transposed copy
//     of any activation is ever written; fp32 partial tiles are added with global atomics once
//     per workgroup (workgroups walk many pixel tiles);
//   * wgrad_finish_kernel: layout back to OIHW and the backward of weight standardisation
//     (denoising_diffusion.py:109-112).
#include <cstdlib>
#include <type_traits>
#include "blocks.h"
#include "det.h"
#include "conv_params.h"
#include "mfma_util.h"

namespace ofd {

// prepared forward weights [tap][Cin/8][Cout][8]  ->  dgrad weights [T-1-tap][Cout/8][Cin][8]
__global__ void __launch_bounds__(256) wt_transpose_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ wt, int taps, int Cin, int Cout) {
    const size_t total = (size_t)taps * Cin * Cout;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        // i indexes the OUTPUT: [tap'][co8][ci][j]
        const int j = (int)(i & 7);
        const size_t r = i >> 3;
        const int ci = (int)(r % Cin);
        const size_t r2 = r / Cin;
        const int co8 = (int)(r2 % (Cout / 8)), tapp = (int)(r2 / (Cout / 8));
        const int co = co8 * 8 + j, tap = taps - 1 - tapp;
        wt[i] = w[(((size_t)tap * (Cin / 8) + ci / 8) * Cout + co) * 8 + (ci & 7)];
    }
}

// all convs in one launch: desc.w = prepared weights, desc.out = transposed; a conv owns blocks [block0, next block0), each block
// walks its conv's elements with a stride of that many blocks
__global__ void __launch_bounds__(256) wt_transpose_batched_kernel(const ofd_weight_prep_desc* __restrict__ descs, int n, int total_blocks) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const ofd_weight_prep_desc d = descs[lo];
    const int nb = (lo + 1 < n ? descs[lo + 1].block0 : total_blocks) - d.block0, lb = (int)blockIdx.x - d.block0;
    const bf16_t* w = (const bf16_t*)d.w;
    bf16_t* wt = (bf16_t*)d.out;
    const int taps = d.ksize * d.ksize, Cin = d.Cin_pad, Cout = d.Cout;
    const size_t total = (size_t)taps * Cin * Cout;
    for (size_t i = (size_t)lb * 256 + threadIdx.x; i < total; i += (size_t)nb * 256) {
        const int j = (int)(i & 7);
        const size_t r = i >> 3;
        const int ci = (int)(r % Cin);
        const size_t r2 = r / Cin;
        const int co8 = (int)(r2 % (Cout / 8)), tapp = (int)(r2 / (Cout / 8));
        const int co = co8 * 8 + j, tap = taps - 1 - tapp;
        wt[i] = w[(((size_t)tap * (Cin / 8) + ci / 8) * Cout + co) * 8 + (ci & 7)];
    }
}

struct WgradParams {
    int B, H, W, Cout, Cin_total, n_src, tiles_x, tiles_y;
    ConvSrcDev src[4];
    const bf16_t* dy;
    float* dw;           // [taps][Cin_total][Cout] fp32, accumulated with atomics (zeroed by the caller)
    float* dbias;        // optional [Cout]: += column sums of dY (workgroups of the first ci block add them)
    // 3x3 only: X = SiLU(x * in_scale[b][c] + in_shift[b][c]) applied while the halo tile is staged (the forward conv's
    // GroupNorm-affine + SiLU prologue): the weight gradient of a block's second conv reads h1 instead of a materialised act1
    const float* in_scale;
    const float* in_shift;
    int no_dma;          // A/B switch (OFD_WGRAD_NO_DMA): 3x3 dY tiles through registers
    int dbg;             // conv_wgrad3_db_kernel ablation bits (OFD_WGRAD_DBG; 0 in production): 1 no DMA after the first tile, 2 no fragment reads after the first row, 4 no MFMAs
    // 3x3 only: where output pixel (b, oy, ox) of the (H, W) grid lives in dY: pixel b * dy_bs + (oy * dy_s + dy_y0) * dy_w + ox * dy_s + dy_x0
    // (plain: dy_bs = H W, dy_w = W, dy_s = 1; one phase of an up-sample conv, see conv_wgrad3_kernel: the stride-2 samples of the 2H x 2W tensor)
    long dy_bs;
    int dy_w, dy_s, dy_y0, dy_x0;
};

// Upsample(x2, nearest) + 3x3 (DD:89-93) as the forward runs it: four phases (a, b) of output pixels (2y + a, 2x + b), each a 2x2-tap conv on the
// LOW-resolution tensor with sums of the 3x3 weights.  Low-resolution row offset that kernel row K reads in phase a: a = 0 -> (-1, 0, 0), a = 1 -> (0, 0, +1).
__host__ __device__ constexpr int up2_off(int a, int K) { return a == 0 ? (K == 0 ? -1 : 0) : (K == 2 ? 1 : 0); }
// taps (ky, kx) of the 3x3 low-resolution weight gradient that phase PH = 2 a + b needs (PH < 0: a plain conv, all nine)
__host__ __device__ constexpr bool wg3_tap_on(int PH, int ky, int kx) {
    return PH < 0 || (((PH >> 1) == 0 ? ky <= 1 : ky >= 1) && ((PH & 1) == 0 ? kx <= 1 : kx >= 1));
}

// column sums of a staged dY tile [256 pixels][64 co] (128-byte rows): thread -> (co, quarter of the pixels)
__device__ __forceinline__ float ytile_colsum(const unsigned char* ys, int tid) {
    const int co = tid & 63, part = tid >> 6;
    float s = 0.0f;
#pragma unroll 8
    for (int p = part * 64; p < part * 64 + 64; ++p) s += bf2f(*(const bf16_t*)(ys + p * 128 + co * 2));
    return s;
}

// grid: (pixel-tile groups, KS kernel rows, (Cin/64)*(Cout/64)); workgroup = 4 waves, wave -> 32 ci x 32 co
template <int KS>
__global__ void __launch_bounds__(256) conv_wgrad_kernel(const WgradParams P) {
    constexpr int IWK = 32 + KS - 1, XPIX = 8 * IWK, YPIX = 256;
    constexpr int XPT = (XPIX * 8 + 255) / 256, YPT = YPIX * 8 / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xs = smem;                  // [XPIX][64 ch] bf16, pixel-major
    unsigned char* ys = smem + XPIX * 128;     // [YPIX][64 co]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int ky = blockIdx.y, ncob = P.Cout / 64, kc = blockIdx.z / ncob, cob = blockIdx.z % ncob;
    const int cit = wave & 1, cot = wave >> 1;
    const int tpi = P.tiles_x * P.tiles_y, ntiles = tpi * P.B;
    int si = 0, first = 0;
    while (si + 1 < P.n_src && kc >= first + P.src[si].chunks) { first += P.src[si].chunks; ++si; }
    const int kcl = kc - first;
    const ConvSrcDev S = P.src[si];
    const int c8 = tid & 7;

    f32x16 acc[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
    const bool do_bias = P.dbias && kc == 0 && ky == 0;
    float bsum = 0.0f;

    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int b = t / tpi, t_in = t - b * tpi;
        const int oy0 = (t_in / P.tiles_x) * 8, ox0 = (t_in % P.tiles_x) * 32;
        u32x4 xr[XPT], yr[YPT];
        unsigned xok = 0, yok = 0;
        const bf16_t* xbase = S.ptr + (size_t)b * S.SH * S.SW * S.src_channels + S.ch_offset + kcl * 64 + c8 * 8;
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int p = min((tid >> 3) + i * 32, XPIX - 1);
            const int ty = p / IWK, tx = p - ty * IWK;
            const int iy = oy0 + ty + ky - KS / 2, ix = ox0 + tx - KS / 2;
            const bool ok = iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
            xok |= (ok ? 1u : 0u) << i;
            const int cy = min(max(iy, 0), P.H - 1), cx = min(max(ix, 0), P.W - 1);
            int sy = cy, sx = cx;
            if (S.mode == 1) { sy = cy >> 1; sx = cx >> 1; }
            else if (S.mode == 2) { sy = 2 * cy + S.p1; sx = 2 * cx + S.p2; }
            xr[i] = *(const u32x4*)(xbase + ((size_t)sy * S.SW + sx) * S.src_channels);
        }
#pragma unroll
        for (int i = 0; i < YPT; ++i) {
            const int p = (tid >> 3) + i * 32;
            const int oy = oy0 + (p >> 5), ox = ox0 + (p & 31);
            const bool ok = oy < P.H && ox < P.W;
            yok |= (ok ? 1u : 0u) << i;
            yr[i] = *(const u32x4*)(P.dy + (((size_t)b * P.H + min(oy, P.H - 1)) * P.W + min(ox, P.W - 1)) * P.Cout + cob * 64 + c8 * 8);
        }
        __syncthreads();     // previous tile's operand reads are complete
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int p = min((tid >> 3) + i * 32, XPIX - 1);
            u32x4 v = xr[i];
            const bool ok = (xok >> i) & 1u;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ok ? v[j] : 0u;
            *(u32x4*)(xs + p * 128 + c8 * 16) = v;
        }
#pragma unroll
        for (int i = 0; i < YPT; ++i) {
            const int p = (tid >> 3) + i * 32;
            u32x4 v = yr[i];
            const bool ok = (yok >> i) & 1u;      // pixels of the tile overhang contribute nothing
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ok ? v[j] : 0u;
            *(u32x4*)(ys + p * 128 + c8 * 16) = v;
        }
        __syncthreads();
        if (do_bias) bsum += ytile_colsum(ys, tid);
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int xb = 0; xb < 2; ++xb) {
                const bf16x8 yf = tr_frag(ys + ((r * 32 + xb * 16) * 64 + cot * 32) * 2, 128, lane);
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const bf16x8 xf = tr_frag(xs + ((r * IWK + xb * 16 + kx) * 64 + cit * 32) * 2, 128, lane);
                    acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, yf, acc[kx], 0, 0, 0);   // rows = ci, cols = co
                }
            }
    }
    if (do_bias) gacc_add(P.dbias + cob * 64 + (tid & 63), bsum);
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
        const int tap = ky * KS + kx;
        float* d = P.dw + ((size_t)tap * P.Cin_total + kc * 64 + cit * 32) * P.Cout + cob * 64 + cot * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = (r & 3) + 8 * (r >> 2) + 4 * half;
            gacc_add(d + (size_t)ci * P.Cout, acc[kx][r]);
        }
    }
}


// 1x1, Cout = 384 (LinearAttention / Attention to_qkv, DD:222,252): dW[ci][co] = sum_p X[p][ci] dY[p][co] is a reduction over
// millions of pixels into a 64..512 x 384 matrix.  The generic kernel above gives a workgroup one 64 x 64 block of it, so dY
// (5.5 GB at full resolution) is re-read per ci block and X per co block, with 16 MFMAs per wave for 64 KB of staged operands.
// Here a workgroup owns one ci block and ALL 384 output channels (wave -> 64 ci x 96 co = 6 accumulator tiles): every pixel's
// dY row is read once per ci block, X once; 5 fragment reads per 6 MFMAs.  Pixels are walked as a flat [npix][C] array in
// tiles of 64.
constexpr int WQ_PX = 64, WQ_CO = 384, WQ_YP = WQ_CO * 2 + 64;      // dY row pitch: 16 dwords (mod 64 banks) apart
__global__ void __launch_bounds__(256, 2) conv_wgrad1_qkv_kernel(const bf16_t* __restrict__ x, int x_stride, const bf16_t* __restrict__ dy,
                                                                 float* __restrict__ dw, size_t npix, int Cin_total) {
    __shared__ __attribute__((aligned(16))) unsigned char xs[WQ_PX * 128];
    __shared__ __attribute__((aligned(16))) unsigned char ys[WQ_PX * WQ_YP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int kc = blockIdx.y;
    f32x16 acc[2][3];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][j][r] = 0.0f;
    const size_t ntiles = (npix + WQ_PX - 1) / WQ_PX;
    // (r04: the cross-tile register prefetch that helped conv_wgrad1_wide_kernel measured SLOWER here -- 128 -> 384 at half resolution 0.376 -> 0.514 ms,
    // and 0.459 with the loads in the loop but the out-of-range selects moved to the LDS write: the kernel is left as it was)
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const size_t p0 = t * WQ_PX;
        u32x4 xr[2], yr[12];
#pragma unroll
        for (int i = 0; i < 2; ++i) {                     // X tile: 64 pixels x 8 units
            const int u = tid + i * 256, p = u >> 3, c8 = u & 7;
            const size_t gp = min(p0 + p, npix - 1);
            xr[i] = *(const u32x4*)(x + gp * x_stride + kc * 64 + c8 * 8);
            if (p0 + p >= npix) xr[i] = u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) {                    // dY tile: 64 pixels x 48 units
            const int u = tid + i * 256, p = u / 48, c8 = u - p * 48;
            const size_t gp = min(p0 + p, npix - 1);
            yr[i] = *(const u32x4*)(dy + gp * WQ_CO + c8 * 8);
            if (p0 + p >= npix) yr[i] = u32x4{0u, 0u, 0u, 0u};
        }
        __syncthreads();                                  // previous tile's fragment reads are complete
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int u = tid + i * 256;
            *(u32x4*)(xs + (u >> 3) * 128 + (u & 7) * 16) = xr[i];
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int u = tid + i * 256, p = u / 48, c8 = u - p * 48;
            *(u32x4*)(ys + p * WQ_YP + c8 * 16) = yr[i];
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < WQ_PX / 16; ++ks) {
            bf16x8 xf[2], yf[3];
#pragma unroll
            for (int a = 0; a < 2; ++a) xf[a] = tr_frag(xs + (ks * 16) * 128 + a * 64, 128, lane);
#pragma unroll
            for (int j = 0; j < 3; ++j) yf[j] = tr_frag(ys + (ks * 16) * WQ_YP + (wave * 96 + j * 32) * 2, WQ_YP, lane);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[a][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[a], yf[j], acc[a][j], 0, 0, 0);   // rows = ci, cols = co
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float* d = dw + ((size_t)kc * 64 + a * 32) * WQ_CO + wave * 96 + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = (r & 3) + 8 * (r >> 2) + 4 * half;
                gacc_add(d + (size_t)ci * WQ_CO, acc[a][j][r]);
            }
        }
}

// The same idea for the other 1x1 layers (res_conv of the concatenating blocks, to_out.0; Cout = 64 / 128 / 192 / 256, bias, several
// sources): a workgroup owns one 64-channel ci block and ALL Cout output channels, so dY is read once per ci block instead of once per
// (ci, co) block pair and X once instead of once per co block -- these kernels are pure traffic (32 flop per staged byte in the generic
// form: `res_conv` 256 -> 128 at half resolution moved 3.7 GB for 1.4 GB of tensors).
//   WCO = 4: wave -> 64 ci x Cout / 4 co (Cout = 128, 256);  WCO = 2: wave -> 32 ci x Cout / 2 co (Cout = 64, 192)
template <int CO, int WCO>
__global__ void __launch_bounds__(256, 2) conv_wgrad1_wide_kernel(const WgradParams P, size_t npix) {
    constexpr int YP = CO * 2 + 64, SPAN = CO / WCO, NJ = SPAN / 32, NA = WCO == 4 ? 2 : 1, YU = CO / 8, YPT = WQ_PX * YU / 256;
    static_assert(SPAN % 32 == 0 && (WQ_PX * YU) % 256 == 0, "wave tiling");
    __shared__ __attribute__((aligned(16))) unsigned char xs[WQ_PX * 128];
    __shared__ __attribute__((aligned(16))) unsigned char ys[WQ_PX * YP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int kc = blockIdx.y, zc = blockIdx.z * CO;             // this workgroup's ci block; first of its CO output channels (Cout = 512: two column blocks of 256)
    const int cw = WCO == 4 ? wave : wave >> 1, a0 = WCO == 4 ? 0 : wave & 1;
    int si = 0, first = 0;
    while (si + 1 < P.n_src && kc >= first + P.src[si].chunks) { first += P.src[si].chunks; ++si; }
    const bf16_t* x = P.src[si].ptr + P.src[si].ch_offset + (kc - first) * 64;
    const int x_stride = P.src[si].src_channels;
    const bool do_bias = P.dbias && kc == 0;
    float bsum = 0.0f;
    f32x16 acc[NA][NJ];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][j][r] = 0.0f;
    const size_t ntiles = (npix + WQ_PX - 1) / WQ_PX;
    // the tiles are pure traffic (24 KB staged for 8 MFMAs per wave at Cout = 128): the loads of tile t + G are issued before the MFMAs of tile t
    // and land while they and the next barrier pair run (r04: without the prefetch the loads were in flight only part of the time -- 192 -> 128 at
    // half resolution ran at 1.8 TB/s)
    u32x4 xr[2], yr[YPT];
    unsigned okm = 0;                   // bit i: xr[i] is a pixel of the tensor, bit 2 + i: yr[i] -- applied when the registers are written to LDS (a select
                                        // right behind each load made the compiler wait for it: ten serial round trips per tile)
    auto load_tile = [&](size_t t) {
        const size_t p0 = t * WQ_PX;
        okm = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {                     // X tile: 64 pixels x 8 units
            const int u = tid + i * 256, p = u >> 3, c8 = u & 7;
            const size_t gp = min(p0 + p, npix - 1);
            xr[i] = *(const u32x4*)(x + gp * x_stride + c8 * 8);
            okm |= (p0 + p < npix ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < YPT; ++i) {                   // dY tile: 64 pixels x CO / 8 units
            const int u = tid + i * 256, p = u / YU, c8 = u - p * YU;
            const size_t gp = min(p0 + p, npix - 1);
            yr[i] = *(const u32x4*)(P.dy + gp * P.Cout + zc + c8 * 8);
            okm |= (p0 + p < npix ? 1u : 0u) << (2 + i);
        }
    };
    if (blockIdx.x < ntiles) load_tile(blockIdx.x);
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        __syncthreads();                                  // previous tile's fragment reads are complete
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int u = tid + i * 256;
            *(u32x4*)(xs + (u >> 3) * 128 + (u & 7) * 16) = ((okm >> i) & 1u) ? xr[i] : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < YPT; ++i) {
            const int u = tid + i * 256, p = u / YU, c8 = u - p * YU;
            *(u32x4*)(ys + p * YP + c8 * 16) = ((okm >> (2 + i)) & 1u) ? yr[i] : u32x4{0u, 0u, 0u, 0u};
        }
        __syncthreads();
        if (t + gridDim.x < ntiles) load_tile(t + gridDim.x);
        if (do_bias && tid < CO) {
#pragma unroll 8
            for (int p = 0; p < WQ_PX; ++p) bsum += bf2f(*(const bf16_t*)(ys + p * YP + tid * 2));
        }
#pragma unroll
        for (int ks = 0; ks < WQ_PX / 16; ++ks) {
            bf16x8 xf[NA], yf[NJ];
#pragma unroll
            for (int a = 0; a < NA; ++a) xf[a] = tr_frag(xs + (ks * 16) * 128 + (a0 + a) * 64, 128, lane);
#pragma unroll
            for (int j = 0; j < NJ; ++j) yf[j] = tr_frag(ys + (ks * 16) * YP + (cw * SPAN + j * 32) * 2, YP, lane);
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[a][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[a], yf[j], acc[a][j], 0, 0, 0);   // rows = ci, cols = co
        }
    }
    if (do_bias && tid < CO) gacc_add(P.dbias + zc + tid, bsum);
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float* d = P.dw + ((size_t)kc * 64 + (a0 + a) * 32) * P.Cout + zc + cw * SPAN + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = (r & 3) + 8 * (r >> 2) + 4 * half;
                gacc_add(d + (size_t)ci * P.Cout, acc[a][j][r]);
            }
        }
}

template <int CO, int WCO>
static void launch_wgrad1_wide(const WgradParams& P, size_t npix, int ncib, hipStream_t s) {
    // workgroups over all ci blocks: two per CU; four for Cout = 64, whose 20 KB / 62-register workgroups are short of loads in flight at two
    // (full-resolution res_conv 128 -> 64: 0.76 / 0.61 / 0.66 ms at 512 / 1024 / 2048; 256 -> 128 at half resolution: 0.355 / 0.378 / 0.343)
    static const int total_env = getenv("OFD_WGRAD1_WIDE_WGS") ? atoi(getenv("OFD_WGRAD1_WIDE_WGS")) : 0;
    const int total = total_env > 0 ? total_env : (CO == 64 ? 1024 : 512);
    int g = cdiv(total, ncib * (P.Cout / CO));
    const size_t nt = (npix + WQ_PX - 1) / WQ_PX;
    if ((size_t)g > nt) g = (int)nt;
    conv_wgrad1_wide_kernel<CO, WCO><<<dim3(g, ncib, P.Cout / CO), 256, 0, s>>>(P, npix);
}

// 3x3: all nine taps in one workgroup.  grid (pixel-tile groups, (Cin/64)*(Cout/64)); 4 waves, wave -> 32 ci x 32 co x 9 taps
// (144 accumulator registers).  The 10 x 34 halo tile and the 8 x 32 dY tile are read from HBM once per
// (ci block, co block): 248 flop per byte, against 83 for one kernel row per workgroup.  The halo rows are
// walked once; a halo row rr feeds output rows rr, rr-1, rr-2 (ky = 0, 1, 2), whose dY fragments stay in a
// three-row register window: 8 fragment reads per 18 MFMAs.
// PH >= 0 (phase 2 a + b of an up-sample conv): X is the low-resolution tensor, (H, W) its grid, dY the stride-2 samples (2y + a, 2x + b) of the
// full-resolution gradient; only the 2 x 2 taps the phase reads are multiplied (4 / 9 of the MFMAs, and four launches cover every dY pixel once:
// 2.25x fewer MACs than the same gradient on the virtual up-sampled tensor), and a tap's sum goes to every 3x3 weight the phase folded into it.
template <bool PRO, bool DMA, int PH = -1>      // PRO: SiLU(affine) prologue on the staged input (its own instantiation: the plain one keeps its register budget); DMA: dY tile by global_load_lds
__global__ void __launch_bounds__(256, 2) conv_wgrad3_kernel(const WgradParams P) {
    constexpr int IWK = 34, XROWS = 10, XPIX = XROWS * IWK, YPIX = 256;
    constexpr int XPT = (XPIX * 8 + 255) / 256, YPT = YPIX * 8 / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xs = smem;                  // [XPIX][64 ch] bf16, pixel-major
    unsigned char* ys = smem + XPIX * 128;     // [YPIX][64 co]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int ncob = P.Cout / 64, kc = blockIdx.y / ncob, cob = blockIdx.y % ncob;
    const int cit = wave & 1, cot = wave >> 1;
    const int tpi = P.tiles_x * P.tiles_y, ntiles = tpi * P.B;
    int si = 0, first = 0;
    while (si + 1 < P.n_src && kc >= first + P.src[si].chunks) { first += P.src[si].chunks; ++si; }
    const int kcl = kc - first;
    const ConvSrcDev S = P.src[si];
    const int c8 = tid & 7;

    const bool do_bias = P.dbias && kc == 0;
    float bsum = 0.0f;
    f32x16 acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][k][r] = 0.0f;

    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int b = t / tpi, t_in = t - b * tpi;
        const int oy0 = (t_in / P.tiles_x) * 8, ox0 = (t_in % P.tiles_x) * 32;
        if constexpr (DMA) {
            // dY tile straight into LDS (global_load_lds, lane-linear = the pixel-major [pixel][128 B] layout; pixels of a tile overhang
            // read a clamped address and are zeroed in LDS afterwards), issued FIRST so that it flies with the halo tile's loads,
            // transform and stores -- through registers it could only start after those (the two tiles do not fit the register file).
            // piece j of wave w = pixels (row 2w + j/4, columns 8 (j%4) .. +7): one per-lane base, uniform offsets per piece
            __syncthreads();     // previous tile's operand reads are complete
            if (oy0 + 8 <= P.H && ox0 + 32 <= P.W) {
                const bf16_t* base = P.dy + ((size_t)b * P.dy_bs + ((size_t)(oy0 + wave * 2) * P.dy_s + P.dy_y0) * P.dy_w + (ox0 + (lane >> 3)) * P.dy_s + P.dy_x0) * P.Cout +
                                     cob * 64 + (lane & 7) * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    __builtin_amdgcn_global_load_lds(base + ((size_t)(j >> 2) * P.dy_s * P.dy_w + (j & 3) * 8 * P.dy_s) * P.Cout,
                                                     (__attribute__((address_space(3))) void*)(ys + (wave * 8 + j) * 1024), 16, 0, 0);
            } else {
#pragma unroll 1
                for (int j = 0; j < 8; ++j) {
                    const int oy = min(oy0 + wave * 2 + (j >> 2), P.H - 1), ox = min(ox0 + (j & 3) * 8 + (lane >> 3), P.W - 1);
                    const bf16_t* src = P.dy + ((size_t)b * P.dy_bs + ((size_t)oy * P.dy_s + P.dy_y0) * P.dy_w + ox * P.dy_s + P.dy_x0) * P.Cout + cob * 64 + (lane & 7) * 8;
                    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(ys + (wave * 8 + j) * 1024), 16, 0, 0);
                }
            }
        }
        u32x4 xr[XPT];
        unsigned xok = 0;
        const bf16_t* xbase = S.ptr + (size_t)b * S.SH * S.SW * S.src_channels + S.ch_offset + kcl * 64 + c8 * 8;
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int p = min((tid >> 3) + i * 32, XPIX - 1);
            const int ty = p / IWK, tx = p - ty * IWK;
            const int iy = oy0 + ty - 1, ix = ox0 + tx - 1;
            const bool ok = iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
            xok |= (ok ? 1u : 0u) << i;
            const int cy = min(max(iy, 0), P.H - 1), cx = min(max(ix, 0), P.W - 1);
            int sy = cy, sx = cx;
            if (S.mode == 1) { sy = cy >> 1; sx = cx >> 1; }
            else if (S.mode == 2) { sy = 2 * cy + S.p1; sx = 2 * cx + S.p2; }
            xr[i] = *(const u32x4*)(xbase + ((size_t)sy * S.SW + sx) * S.src_channels);
        }
        if constexpr (!DMA) __syncthreads();     // previous tile's operand reads are complete
        float ps[8], pb[8];
        if constexpr (PRO) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                ps[j] = P.in_scale[(size_t)b * P.Cin_total + kc * 64 + c8 * 8 + j];
                pb[j] = P.in_shift[(size_t)b * P.Cin_total + kc * 64 + c8 * 8 + j];
            }
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int p = (tid >> 3) + i * 32;
            u32x4 v = xr[i];
            if constexpr (PRO) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float z0 = bf2f((bf16_t)(v[j] & 0xffffu)) * ps[2 * j] + pb[2 * j];
                    const float z1 = bf2f((bf16_t)(v[j] >> 16)) * ps[2 * j + 1] + pb[2 * j + 1];
                    v[j] = f2bf2(z0 * __builtin_amdgcn_rcpf(1.0f + __expf(-z0)), z1 * __builtin_amdgcn_rcpf(1.0f + __expf(-z1)));
                }
            }
            const bool ok = (xok >> i) & 1u;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ok ? v[j] : 0u;
            if (p < XPIX) *(u32x4*)(xs + p * 128 + c8 * 16) = v;
        }
        // (dY tile fetched after the halo tile left its registers: both at once do not fit two workgroups per CU)
        if constexpr (!DMA) {
        u32x4 yr[YPT];
        unsigned yok = 0;
#pragma unroll
        for (int i = 0; i < YPT; ++i) {
            const int p = (tid >> 3) + i * 32;
            const int oy = oy0 + (p >> 5), ox = ox0 + (p & 31);
            const bool ok = oy < P.H && ox < P.W;
            yok |= (ok ? 1u : 0u) << i;
            yr[i] = *(const u32x4*)(P.dy + ((size_t)b * P.dy_bs + ((size_t)min(oy, P.H - 1) * P.dy_s + P.dy_y0) * P.dy_w + min(ox, P.W - 1) * P.dy_s + P.dy_x0) * P.Cout + cob * 64 + c8 * 8);
        }
#pragma unroll
        for (int i = 0; i < YPT; ++i) {
            const int p = (tid >> 3) + i * 32;
            u32x4 v = yr[i];
            const bool ok = (yok >> i) & 1u;      // pixels of the tile overhang contribute nothing
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ok ? v[j] : 0u;
            *(u32x4*)(ys + p * 128 + c8 * 16) = v;
        }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of the dY tile have landed
            if (oy0 + 8 > P.H || ox0 + 32 > P.W) {                // tile overhang: those pixels contribute nothing
#pragma unroll 1
                for (int j = 0; j < 8; ++j) {
                    const int p = (wave * 8 + j) * 8 + (lane >> 3);
                    if (oy0 + (p >> 5) >= P.H || ox0 + (p & 31) >= P.W) *(u32x4*)(ys + p * 128 + (lane & 7) * 16) = u32x4{0u, 0u, 0u, 0u};
                }
            }
        }
        __syncthreads();
        if (do_bias) bsum += ytile_colsum(ys, tid);
        bf16x8 yw[3][2];     // dY fragments of output rows rr, rr-1, rr-2
#pragma unroll
        for (int rr = 0; rr < XROWS; ++rr) {
#pragma unroll
            for (int xb = 0; xb < 2; ++xb) {
                yw[2][xb] = yw[1][xb];
                yw[1][xb] = yw[0][xb];
                if (rr < 8) yw[0][xb] = tr_frag(ys + ((rr * 32 + xb * 16) * 64 + cot * 32) * 2, 128, lane);
            }
#pragma unroll
            for (int xb = 0; xb < 2; ++xb)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    if (!(wg3_tap_on(PH, 0, kx) || wg3_tap_on(PH, 1, kx) || wg3_tap_on(PH, 2, kx))) continue;
                    const bf16x8 xf = tr_frag(xs + ((rr * IWK + xb * 16 + kx) * 64 + cit * 32) * 2, 128, lane);
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
                        if (rr - ky >= 0 && rr - ky < 8 && wg3_tap_on(PH, ky, kx))
                            acc[ky][kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, yw[ky][xb], acc[ky][kx], 0, 0, 0);   // rows = ci, cols = co
                }
        }
    }
    if (do_bias) gacc_add(P.dbias + cob * 64 + (tid & 63), bsum);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            // (ky, kx): the 3x3 WEIGHT this sum goes to; a phase adds the sum of the low-resolution tap that weight was folded into
            const int ay = PH < 0 ? ky : up2_off(PH >> 1, ky) + 1, ax = PH < 0 ? kx : up2_off(PH & 1, kx) + 1;
            float* d = P.dw + ((size_t)(ky * 3 + kx) * P.Cin_total + kc * 64 + cit * 32) * P.Cout + cob * 64 + cot * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = (r & 3) + 8 * (r >> 2) + 4 * half;
                gacc_add(d + (size_t)ci * P.Cout, acc[ay][ax][r]);
            }
        }
}

// ---- 3x3, double-buffered (r03): the kernel above alternates "stage a tile" and "multiply it" inside a workgroup and relies on the second
// resident workgroup to fill the gaps (staging alone 14.5 ms, MFMAs alone 17.0 ms, together 25.2 ms per training step).  Here ONE workgroup of
// 8 waves per CU owns both LDS halves (2 x 75 KB): while tile t is multiplied, the halo tile and the dY tile of tile t + 1 arrive by LDS-DMA
// (global_load_lds_dwordx4, no staging registers) in the other half -- one wait + one barrier per tile.
//   * wave -> (16-pixel column block xb, 32 ci x 32 co quadrant): all ten halo rows, nine accumulator tiles (144 registers), 72 MFMAs per tile;
//   * the operand fragments are read with inline-asm ds_read_b64_tr_b16 (one base register per operand, immediate offsets): an LDS load the
//     compiler can see makes it wait for every outstanding LDS-DMA, i.e. for the NEXT tile.  Row rr + 1's fragments are requested before row
//     rr's MFMAs are issued and awaited (lgkmcnt(0)) after them;
//   * halo pixels outside the image / pixels of a tile overhang are fetched from a clamped address and zeroed in LDS by the lane that fetched
//     them, after its own vmcnt(0) and before the barrier.
template <int OFF>
__device__ __forceinline__ void lds_tr8(s16x4& dst, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
struct TrFrag {
    s16x4 lo, hi;
    __device__ __forceinline__ bf16x8 get() const {
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
};
template <int OFF>
__device__ __forceinline__ void tr_read(TrFrag& f, unsigned addr) {      // rows 128 bytes apart
    lds_tr8<OFF>(f.lo, addr);
    lds_tr8<OFF + 512>(f.hi, addr);
}
__device__ __forceinline__ void lds_landed(TrFrag& a, TrFrag& b, TrFrag& c, TrFrag& d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi), "+v"(c.lo), "+v"(c.hi), "+v"(d.lo), "+v"(d.hi) : : "memory");
}
template <int I, int N, class Fn>
__device__ __forceinline__ void wg_static_for(Fn&& fn) {
    if constexpr (I < N) {
        fn(std::integral_constant<int, I>{});
        wg_static_for<I + 1, N>(fn);
    }
}

__global__ void __launch_bounds__(512, 1) conv_wgrad3_db_kernel(const WgradParams P) {
    constexpr int IWK = 34, XPIX = 10 * IWK, XPIECES = (XPIX + 7) / 8, XB = XPIECES * 1024, BUF = XB + 256 * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, half = lane >> 5;
    const int ncob = P.Cout / 64, kc = blockIdx.y / ncob, cob = blockIdx.y % ncob;
    const int quad = wave & 3, xb = wave >> 2, cit = quad & 1, cot = quad >> 1;
    const int tpi = P.tiles_x * P.tiles_y, ntiles = tpi * P.B;
    int si = 0, first = 0;
    while (si + 1 < P.n_src && kc >= first + P.src[si].chunks) { first += P.src[si].chunks; ++si; }
    const int kcl = kc - first;
    const ConvSrcDev S = P.src[si];
    const bool do_bias = P.dbias && kc == 0;
    float bsum = 0.0f;
    f32x16 acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][k][r] = 0.0f;

    // tile-invariant part of this lane's DMA addresses: (row, column) of its halo pixels inside the tile, its swizzled channel unit
    // (LDS rows are 128 bytes = all 32 banks: unswizzled, the four rows a 16-lane group of a transposing read touches collide 4-way.  The
    // 16-byte unit u of tile pixel p therefore holds channel unit u ^ 2 (p & 3) -- the DMA is lane-linear, so the permutation is applied
    // to each lane's SOURCE address, and again to the fragment reads' addresses)
    const int H = P.H, W = P.W, SW = S.SW, sch = S.src_channels, tiles_x = P.tiles_x;
    // source pixel of conv-input pixel (cy, cx), branch-free: same size (c), nearest up-sampled (c >> 1), pixel-unshuffled sub-pixel (2 c + p)
    const int m_mul = S.mode == 2 ? 2 : 1, m_shr = S.mode == 1 ? 1 : 0, m_ay = S.mode == 2 ? S.p1 : 0, m_ax = S.mode == 2 ? S.p2 : 0;
    const int dy_s = P.dy_s, dy_w = P.dy_w, dy_y0 = P.dy_y0, dy_x0 = P.dy_x0, Cout = P.Cout;
    const int swz = ((lane & 7) ^ (2 * ((lane >> 3) & 3))) * 8;          // (piece * 8 is a multiple of 4: pixel & 3 = (lane >> 3) & 3)
    int x_ty[6], x_tx[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int pc = min((wave + 8 * i) * 8 + (lane >> 3), XPIX - 1);
        x_ty[i] = pc / IWK - 1;
        x_tx[i] = pc - (pc / IWK) * IWK - 1;
    }
    const bf16_t* xsrc = S.ptr + S.ch_offset + kcl * 64 + swz;
    const size_t x_bs = (size_t)S.SH * SW * sch;
    const bf16_t* ysrc = P.dy + cob * 64 + swz;
    const int y_r = (wave * 32 + (lane >> 3)) >> 5, y_c = (wave * 32 + (lane >> 3)) & 31;      // dY piece j of this wave: pixel (y_r, y_c + 8 j)
    // DMA piece K of a tile (K = 0..5: halo pieces wave + 8 K; 6..9: dY pieces 4 wave + K - 6) into `buf`; returns the bit of `zm` (the mask of this
    // lane's pieces that must read as zero: out-of-image halo pixels, tile overhang) -- (b, oy0, ox0) of the tile in the Tile struct
    struct Tile { const bf16_t* xb_; const bf16_t* yb_; int oy0, ox0; };
    auto tile_of = [&](int t) {
        const int b = t / tpi, t_in = t - b * tpi;
        Tile T;
        T.oy0 = (t_in / tiles_x) * 8; T.ox0 = (t_in % tiles_x) * 32;
        T.xb_ = xsrc + (size_t)b * x_bs;
        T.yb_ = ysrc + (size_t)b * P.dy_bs * Cout;
        return T;
    };
    auto issue_piece = [&](auto KK, const Tile& T, unsigned char* buf) -> unsigned {
        constexpr int K = decltype(KK)::value;
        if constexpr (K < 6) {
            const int piece = wave + 8 * K;                 // wave-uniform
            if (piece >= XPIECES) return 0u;
            const int iy = T.oy0 + x_ty[K], ix = T.ox0 + x_tx[K];
            const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
            const int sy = ((cy * m_mul) >> m_shr) + m_ay, sx = ((cx * m_mul) >> m_shr) + m_ax;
            __builtin_amdgcn_global_load_lds(T.xb_ + (size_t)(((unsigned)sy * (unsigned)SW + (unsigned)sx) * (unsigned)sch), (__attribute__((address_space(3))) void*)(buf + piece * 1024), 16, 0, 0);
            return ((cy == iy && cx == ix) ? 0u : 1u) << K;
        } else {
            constexpr int j = K - 6;
            const int oy = T.oy0 + y_r, ox = T.ox0 + y_c + 8 * j;
            __builtin_amdgcn_global_load_lds(T.yb_ + (size_t)((((unsigned)(min(oy, H - 1) * dy_s + dy_y0)) * (unsigned)dy_w + (unsigned)(min(ox, W - 1) * dy_s + dy_x0)) * (unsigned)Cout),
                                             (__attribute__((address_space(3))) void*)(buf + XB + (wave * 4 + j) * 1024), 16, 0, 0);
            return ((oy < H && ox < W) ? 0u : 1u) << (8 + j);
        }
    };

    // per-lane base addresses of the transposing reads inside a buffer (see tr_frag): rows = pixels, 128 bytes apart
    const int li = lane & 15, tq = li >> 2, tp = li & 3, tcb = (lane >> 4) & 1;
    // swizzled: this lane's row is base row + 8 half + tq (+ 4); k = (base row) & 3 selects one of four per-lane offsets
    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto lane_off = [&](int unit0, int k) {      // unit0: first 16-byte unit of the wave's 32-channel block
        return (unsigned)((8 * half + tq) * 128 + (((unit0 + tcb * 2 + (tp >> 1)) ^ (2 * ((k + tq) & 3))) * 16) + (tp & 1) * 8);
    };
    unsigned x_off[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) x_off[k] = lane_off(cit * 4, k) + (unsigned)(xb * 16 * 128);
    const unsigned y_off = lane_off(cot * 4, 0) + (unsigned)(XB + xb * 16 * 128);

    int t = blockIdx.x, cur = 0;
    unsigned zm = 0;
    if (t < ntiles) {
        const Tile T0 = tile_of(t);
        wg_static_for<0, 10>([&](auto KK) { zm |= issue_piece(KK, T0, smem); });
    }
    for (; t < ntiles; t += gridDim.x) {
        unsigned char* buf = smem + cur * BUF;
        if (!(P.dbg & 8)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of tile t have landed
        if (zm) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
                if ((zm >> i) & 1u) *(u32x4*)(buf + (wave + 8 * i) * 1024 + lane * 16) = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if ((zm >> (8 + j)) & 1u) *(u32x4*)(buf + XB + (wave * 4 + j) * 1024 + lane * 16) = u32x4{0u, 0u, 0u, 0u};
        }
        __syncthreads();                                         // tile t is complete; everyone is done reading the other half (tile t - 1)
        if (do_bias) {                                           // (before the next DMA is requested: these are loads the compiler sees)
            const int co = tid & 63, part = tid >> 6;
            const unsigned char* ys = buf + XB;
#pragma unroll 8
            for (int p = part * 32; p < part * 32 + 32; ++p) bsum += bf2f(*(const bf16_t*)(ys + p * 128 + (((co >> 3) ^ (2 * (p & 3))) * 16) + (co & 7) * 2));
        }
        // the next tile's DMA pieces are requested BETWEEN the MFMAs of rows 0..4 (two per row): their address arithmetic runs in the shadow of
        // this wave's own MFMAs instead of ahead of them (the two waves of a SIMD are in step: nobody else would feed the pipe meanwhile)
        const int tn = t + gridDim.x;
        const bool has_next = tn < ntiles && !(P.dbg & 1);
        const Tile Tn = tile_of(has_next ? tn : t);
        unsigned char* nbuf = smem + (cur ^ 1) * BUF;
        zm = 0;
        unsigned xa[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) xa[k] = smem_base + (unsigned)(cur * BUF) + x_off[k];
        const unsigned ya = smem_base + (unsigned)(cur * BUF) + y_off;
        // halo row rr feeds output rows rr, rr - 1, rr - 2 (ky = 0, 1, 2); fragment sets alternate between rows
        TrFrag xf[2][3], yn[2];
        bf16x8 yw[3];                                            // dY fragments of output rows rr, rr - 1, rr - 2
        tr_read<0>(xf[0][0], xa[0]); tr_read<128>(xf[0][1], xa[1]); tr_read<256>(xf[0][2], xa[2]);
        tr_read<0>(yn[0], ya);
        wg_static_for<0, 10>([&](auto RR) {
            constexpr int rr = decltype(RR)::value, c = rr & 1, nx = c ^ 1;
            lds_landed(xf[c][0], xf[c][1], xf[c][2], yn[c]);
            yw[2] = yw[1];
            yw[1] = yw[0];
            if constexpr (rr < 8) yw[0] = yn[c].get();
            if (rr + 1 < 10 && !(P.dbg & 2)) {
                tr_read<((rr + 1) * IWK) * 128>(xf[nx][0], xa[((rr + 1) * IWK) & 3]);
                tr_read<((rr + 1) * IWK + 1) * 128>(xf[nx][1], xa[((rr + 1) * IWK + 1) & 3]);
                tr_read<((rr + 1) * IWK + 2) * 128>(xf[nx][2], xa[((rr + 1) * IWK + 2) & 3]);
                if constexpr (rr + 1 < 8) tr_read<(rr + 1) * 32 * 128>(yn[nx], ya);
            }
            if constexpr (rr < 5) {
                if (has_next) {
                    zm |= issue_piece(std::integral_constant<int, 2 * rr>{}, Tn, nbuf);
                    zm |= issue_piece(std::integral_constant<int, 2 * rr + 1>{}, Tn, nbuf);
                }
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const bf16x8 xv = xf[c][kx].get();
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
                    if (rr - ky >= 0 && rr - ky < 8 && !(P.dbg & 4)) acc[ky][kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xv, yw[ky], acc[ky][kx], 0, 0, 0);   // rows = ci, cols = co
            }
        });
        cur ^= 1;
    }
    if (do_bias) gacc_add(P.dbias + cob * 64 + (tid & 63), bsum);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            float* d = P.dw + ((size_t)(ky * 3 + kx) * P.Cin_total + kc * 64 + cit * 32) * P.Cout + cob * 64 + cot * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = (r & 3) + 8 * (r >> 2) + 4 * half;
                gacc_add(d + (size_t)ci * P.Cout, acc[ky][kx][r]);
            }
        }
}

// 7x7 init conv (Cin padded to 16, Cout = 64): dW[tap][ci<16][co] = sum_p X16[p + tap][ci] dY[p][co].
// MFMA over pixels like conv_wgrad_kernel.  The halo tile keeps 16 channels = 32 B per pixel, so the
// "second 16-channel block" of a transposing fragment read is simply the NEXT PIXEL: one 32-row A
// operand carries the taps (ky, kx) and (ky, kx+1).  8 waves: wave -> (n-tile of 32 co, two ky rows),
// 8 accumulator tiles each; workgroups walk many 8x32 pixel tiles and add their sums once.
// CH = 8 (r04: the input packed to 8 channels, what the forward conv reads): 16 B per pixel, a 32-row A operand carries FOUR taps (kx .. kx + 3) x 8
// channels -- 2 MFMAs per kernel row and pixel block instead of 4, 4 accumulator tiles per wave; the accumulator keeps its [tap][16][64] layout
// (channels 8 .. 15 stay zero).
template <int CH>
__global__ void __launch_bounds__(512) conv7_wgrad_kernel(const bf16_t* __restrict__ x16, const bf16_t* __restrict__ dy, float* __restrict__ dw,
                                                          int B, int H, int W, int tiles_x, int tiles_y, float* __restrict__ dbias) {
    constexpr int XW = 40, XPIX = 14 * XW;                // 38 columns needed (+1 for the phantom tap kx = 7)
    constexpr int PB = CH * 2, UPP = PB / 16, TPF = 64 / PB, NK = 8 / TPF;      // bytes and 16-byte units per pixel; taps per A fragment; fragments per kernel row
    __shared__ __attribute__((aligned(16))) unsigned char xs[XPIX * PB];
    __shared__ __attribute__((aligned(16))) unsigned char ys[256 * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int nt = wave & 1, kg = wave >> 1;               // ky rows of this wave: kg and kg + 4 (kg + 4 < 7)
    const int tpi = tiles_x * tiles_y, ntiles = tpi * B;
    float bsum = 0.0f;
    f32x16 acc[2][NK];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][k][r] = 0.0f;
    // One 8-wave workgroup per CU (170 registers): nothing else hides a tile's loads, so the tile of step t + 1 is fetched into registers
    // before the MFMAs of step t and written to LDS after them.
    constexpr int XU = (XPIX * UPP + 511) / 512;           // 16-byte units of the halo tile per thread (3; CH = 8: 2)
    // (Out-of-range rows are zeroed when the registers are written to LDS, from the bits of `okm`: a select right behind each load made the compiler
    // wait for all seven before the MFMAs -- the prefetch hid nothing.)
    u32x4 xr[XU], yr[4];
    unsigned okm = 0;
    auto fetch = [&](int t) {
        okm = 0;
        const int b = t / tpi, t_in = t - b * tpi;
        const int oy0 = (t_in / tiles_x) * 8, ox0 = (t_in % tiles_x) * 32;
#pragma unroll
        for (int k = 0; k < XU; ++k) {
            const int i = min(tid + k * 512, XPIX * UPP - 1);
            const int p = i / UPP, u = i % UPP, ty = p / XW, tx = p - ty * XW;
            const int iy = oy0 + ty - 3, ix = ox0 + tx - 3;
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            xr[k] = *(const u32x4*)(x16 + (((size_t)b * H + min(max(iy, 0), H - 1)) * W + min(max(ix, 0), W - 1)) * CH + u * 8);
            okm |= (ok ? 1u : 0u) << k;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + i * 512, p = id >> 3, c8 = id & 7;
            const int oy = oy0 + (p >> 5), ox = ox0 + (p & 31);
            const bool ok = oy < H && ox < W;
            yr[i] = *(const u32x4*)(dy + (((size_t)b * H + min(oy, H - 1)) * W + min(ox, W - 1)) * 64 + c8 * 8);
            okm |= (ok ? 1u : 0u) << (XU + i);
        }
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        __syncthreads();                                   // previous step's fragment reads are complete
#pragma unroll
        for (int k = 0; k < XU; ++k) {
            const int i = tid + k * 512;
            if (i < XPIX * UPP) *(u32x4*)(xs + (i / UPP) * PB + (i % UPP) * 16) = ((okm >> k) & 1u) ? xr[k] : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + i * 512;
            *(u32x4*)(ys + (id >> 3) * 128 + (id & 7) * 16) = ((okm >> (XU + i)) & 1u) ? yr[i] : u32x4{0u, 0u, 0u, 0u};
        }
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);      // (after the barrier: its fence would wait for the loads)
        if (dbias) {
            const int co = tid & 63, part = tid >> 6;
            for (int p = part * 32; p < part * 32 + 32; ++p) bsum += bf2f(*(const bf16_t*)(ys + p * 128 + co * 2));
        }
#pragma unroll 2
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int xb = 0; xb < 2; ++xb) {
                const bf16x8 yf = tr_frag(ys + ((r * 32 + xb * 16) * 64 + nt * 32) * 2, 128, lane);
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int ky = kg + 4 * a;
                    if (ky < 7) {
#pragma unroll
                        for (int k = 0; k < NK; ++k) {
                            const bf16x8 xf = tr_frag(xs + ((r + ky) * XW + xb * 16 + TPF * k) * PB, PB, lane);
                            acc[a][k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, yf, acc[a][k], 0, 0, 0);   // rows = (tap kx .. kx + TPF - 1, ci), cols = co
                        }
                    }
                }
            }
    }
    if (dbias) gacc_add(dbias + (tid & 63), bsum);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int ky = kg + 4 * a;
        if (ky >= 7) continue;
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * half, kx = TPF * k + m / CH, ci = m % CH;
                if (kx < 7) gacc_add(dw + ((size_t)(ky * 7 + kx) * 16 + ci) * 64 + nt * 32 + l31, acc[a][k][r]);
            }
    }
}

// dY [npix][C] bf16 -> out[C] += column sums (bias gradients).  Workgroups stride over the pixels
// (lanes along the channel octets, the rest of the workgroup along pixels), one atomic per channel each.
__global__ void __launch_bounds__(256) channel_sum_kernel(const bf16_t* __restrict__ dy, float* __restrict__ out, size_t npix, int C) {
    __shared__ float red[256][9];
    const int c8n = C / 8, tid = threadIdx.x;
    for (int cu = 0; cu < c8n; cu += 256) {       // (C <= 2048)
        const int lanes_c = min(c8n - cu, 256);    // threads covering different channel octets
        const int rows = 256 / lanes_c;            // pixel phases
        const int my_c = tid % lanes_c, my_r = tid / lanes_c;
        float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (my_r < rows)
            for (size_t p = (size_t)blockIdx.x * rows + my_r; p < npix; p += (size_t)gridDim.x * rows) {
                const uint4 v = *(const uint4*)(dy + p * C + (cu + my_c) * 8);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[2 * j] += bf2f((bf16_t)(w[j] & 0xffffu));
                    a[2 * j + 1] += bf2f((bf16_t)(w[j] >> 16));
                }
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) red[tid][j] = a[j];
        __syncthreads();
        if (tid < lanes_c) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float s = 0.0f;
                for (int r = 0; r < rows; ++r) s += red[r * lanes_c + tid][j];
                gacc_add(out + (cu + tid) * 8 + j, s);
            }
        }
        __syncthreads();
    }
}

// dW accumulator [tap][Cin_pad (engine channel order)][Cout] -> gradient of the fp32 OIHW parameter,
// through weight standardisation when ws_eps >= 0 (DD:109-112):
//   w^ = (w - mean) * rstd ;  dw = rstd * (g - mean(g) - w^ * mean(g * w^))
// one workgroup per output channel; accumulate != 0 adds to dst
__global__ void __launch_bounds__(256) wgrad_finish_kernel(const float* __restrict__ acc, const float* __restrict__ w_raw, float* __restrict__ dst,
                                                           int Cout, int Cin, int Cin_pad, int ksize, float ws_eps, int unshuffle, int accumulate) {
    const int o = blockIdx.x, tid = threadIdx.x, taps = ksize * ksize, n = Cin * taps;
    __shared__ double sh[256];
    auto block_sum = [&](double v) {
        sh[tid] = v;
        __syncthreads();
        for (int k = 128; k > 0; k >>= 1) { if (tid < k) sh[tid] += sh[tid + k]; __syncthreads(); }
        const double r = sh[0];
        __syncthreads();
        return r;
    };
    auto gval = [&](int i) {          // i over the reference's (ci, tap) order
        const int ci = i / taps, tap = i % taps;
        int cp = ci;
        if (unshuffle) { const int Cq = Cin / 4; cp = (ci & 3) * Cq + (ci >> 2); }   // reference c*4+sub -> engine sub*Cq+c
        return acc[((size_t)tap * Cin_pad + cp) * Cout + o];
    };
    const float* wo = w_raw + (size_t)o * n;
    float mean = 0.0f, rstd = 1.0f, mg = 0.0f, mgw = 0.0f;
    if (ws_eps >= 0.0f) {
        double s = 0.0;
        for (int i = tid; i < n; i += 256) s += (double)wo[i];
        const double m = block_sum(s) / n;
        double v = 0.0;
        for (int i = tid; i < n; i += 256) { const double d = (double)wo[i] - m; v += d * d; }
        const double var = block_sum(v) / n;
        mean = (float)m;
        rstd = rsqrtf((float)var + ws_eps);
        // (the accumulator reads are one 4-byte element every Cout floats: eight of them in flight per thread)
        double a = 0.0, b2 = 0.0;
        for (int i0 = tid; i0 < n; i0 += 256 * 8) {
            float gv[8], wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + u * 256, n - 1);
                gv[u] = gval(i);
                wv[u] = wo[i];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u * 256 < n) {
                    const double g = (double)gv[u];
                    a += g;
                    b2 += g * (double)((wv[u] - mean) * rstd);
                }
        }
        mg = (float)(block_sum(a) / n);
        mgw = (float)(block_sum(b2) / n);
    }
    for (int i0 = tid; i0 < n; i0 += 256 * 8) {
        float gv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) gv[u] = gval(min(i0 + u * 256, n - 1));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 256;
            if (i < n) {
                float g = gv[u];
                if (ws_eps >= 0.0f) g = rstd * (g - mg - (wo[i] - mean) * rstd * mgw);
                float* d = dst + (size_t)o * n + i;
                *d = accumulate ? (*d + g) : g;
            }
        }
    }
}

// adjoint of the loader's source modes: D [B][H][W][Ctot] (dgrad output) -> gradient of one source
template <int mode>      // (compile-time: the four loads of the 2x2 sum are issued together; with a run-time trip count each waited for the previous one)
__global__ void __launch_bounds__(256) grad_scatter_kernel(const bf16_t* __restrict__ D, int Ctot, int ch_off, bf16_t* __restrict__ dst, int C,
                                                           int B, int H, int W, int p1, int p2, int accumulate) {
    // mode 0: same size; 1: dst is (H/2, W/2), sum over the 2x2 block; 2: dst is (2H, 2W), writes sub-pixel (p1, p2)
    const int c8n = C / 8;
    const int DH = mode == 1 ? H / 2 : H, DW_ = mode == 1 ? W / 2 : W;     // iteration space
    const size_t total = (size_t)B * DH * DW_ * c8n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cu = (int)(i % c8n);
        size_t r = i / c8n;
        const int x = (int)(r % DW_);
        r /= DW_;
        const int y = (int)(r % DH), b = (int)(r / DH);
        float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        constexpr int reps = mode == 1 ? 4 : 1;
        uint4 v[reps];
#pragma unroll
        for (int k = 0; k < reps; ++k) {
            const int sy = mode == 1 ? 2 * y + (k >> 1) : y, sx = mode == 1 ? 2 * x + (k & 1) : x;
            v[k] = *(const uint4*)(D + (((size_t)b * H + sy) * W + sx) * Ctot + ch_off + cu * 8);
        }
        uint4 o = make_uint4(0u, 0u, 0u, 0u);
        size_t dpix;
        if (mode == 2) dpix = ((size_t)b * (2 * H) + 2 * y + p1) * (2 * W) + 2 * x + p2;
        else dpix = ((size_t)b * DH + y) * DW_ + x;
        bf16_t* d = dst + dpix * C + cu * 8;
        if (accumulate) o = *(const uint4*)d;
#pragma unroll
        for (int k = 0; k < reps; ++k) {
            const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[2 * j] += bf2f((bf16_t)(w[j] & 0xffffu));
                a[2 * j + 1] += bf2f((bf16_t)(w[j] >> 16));
            }
        }
        if (accumulate) {
            const uint32_t w[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[2 * j] += bf2f((bf16_t)(w[j] & 0xffffu));
                a[2 * j + 1] += bf2f((bf16_t)(w[j] >> 16));
            }
        }
        *(uint4*)d = make_uint4(f2bf2(a[0], a[1]), f2bf2(a[2], a[3]), f2bf2(a[4], a[5]), f2bf2(a[6], a[7]));
    }
}

static inline int sgrid_b(size_t total, int block = 256, int cap = 4096) {
    size_t b = (total + block - 1) / block;
    return (int)(b < 1 ? 1 : (b > (size_t)cap ? cap : b));
}

int k_wt_transpose(const bf16_t* w, bf16_t* wt, int taps, int Cin, int Cout, hipStream_t s) {
    wt_transpose_kernel<<<sgrid_b((size_t)taps * Cin * Cout), 256, 0, s>>>(w, wt, taps, Cin, Cout);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

// dw (fp32 [taps][Cin_total][Cout]) must be zeroed by the caller; x sources as in the forward conv
int k_conv_wgrad(const ofd_conv_args* a, const bf16_t* dy, float* dw, hipStream_t s, float* dbias) {
    OFD_CHECK_ARG(a && dy && dw, "conv_wgrad: null argument");
    OFD_CHECK_ARG(a->ksize == 1 || a->ksize == 3, "conv_wgrad: ksize %d (7x7 has its own kernel)", a->ksize);
    OFD_CHECK_ARG(a->Cout % 64 == 0 && a->n_src >= 1 && a->n_src <= 4, "conv_wgrad: bad configuration");
    WgradParams P{};
    P.B = a->B; P.H = a->H; P.W = a->W; P.Cout = a->Cout; P.n_src = a->n_src;
    P.tiles_x = cdiv(a->W, 32); P.tiles_y = cdiv(a->H, 8);
    int cin = 0;
    for (int i = 0; i < a->n_src; ++i) {
        const ofd_conv_src& s_ = a->src[i];
        OFD_CHECK_ARG(s_.src && s_.channels % 64 == 0, "conv_wgrad: source %d channels", i);
        ConvSrcDev& d = P.src[i];
        d.ptr = (const bf16_t*)s_.src; d.chunks = s_.channels / 64; d.src_channels = s_.src_channels; d.ch_offset = s_.ch_offset;
        d.mode = s_.upsample ? 1 : (s_.unshuffle ? 2 : 0);
        d.SH = s_.upsample ? a->H / 2 : (s_.unshuffle ? a->H * 2 : a->H);
        d.SW = s_.upsample ? a->W / 2 : (s_.unshuffle ? a->W * 2 : a->W);
        d.p1 = s_.p1; d.p2 = s_.p2;
        cin += s_.channels;
    }
    P.Cin_total = cin; P.dy = dy; P.dw = dw; P.dbias = dbias;
    P.dy_bs = (long)a->H * a->W; P.dy_w = a->W; P.dy_s = 1; P.dy_y0 = 0; P.dy_x0 = 0;
    OFD_CHECK_ARG(!a->in_scale || (a->in_shift && a->ksize == 3), "conv_wgrad: the input prologue is a 3x3 feature");
    P.in_scale = a->in_scale; P.in_shift = a->in_shift;
    { static const int nd = getenv("OFD_WGRAD_NO_DMA") ? atoi(getenv("OFD_WGRAD_NO_DMA")) : 0; P.no_dma = nd; }
    P.dbg = getenv("OFD_WGRAD_DBG") ? atoi(getenv("OFD_WGRAD_DBG")) : 0;
    const int ntiles = P.tiles_x * P.tiles_y * P.B, combos = (cin / 64) * (a->Cout / 64);
    int gx = cdiv(1024, combos * a->ksize);     // ~4 workgroups per CU in total; each walks ntiles / gx pixel tiles
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    OFD_CHECK_ARG(combos <= 65535, "conv_wgrad: too many channel blocks");
    static const bool no_wq = getenv("OFD_NO_WGRAD_QKV") && atoi(getenv("OFD_NO_WGRAD_QKV"));
    if (a->ksize == 1 && a->Cout == WQ_CO && a->n_src == 1 && P.src[0].mode == 0 && !dbias && !no_wq) {
        const size_t npix = (size_t)a->B * a->H * a->W;
        const int ncib = cin / 64;
        int g = cdiv(512, ncib);                                 // two workgroups per CU
        const size_t nt = (npix + WQ_PX - 1) / WQ_PX;
        if ((size_t)g > nt) g = (int)nt;
        conv_wgrad1_qkv_kernel<<<dim3(g, ncib), 256, 0, s>>>(P.src[0].ptr + P.src[0].ch_offset, P.src[0].src_channels, dy, dw, npix, cin);
        OFD_LAUNCH_CHECK();
        return OFD_OK;
    }
    static const bool no_wide = getenv("OFD_NO_WGRAD1_WIDE") && atoi(getenv("OFD_NO_WGRAD1_WIDE"));
    if (a->ksize == 1 && !no_wide && !P.in_scale && (a->Cout == 64 || a->Cout == 128 || a->Cout == 192 || a->Cout == 256 || a->Cout == 512)) {
        bool plain = true;
        for (int i = 0; i < a->n_src; ++i) plain = plain && P.src[i].mode == 0;
        if (plain) {
            const size_t npix = (size_t)a->B * a->H * a->W;
            const int ncib = cin / 64;
            if (a->Cout == 64) launch_wgrad1_wide<64, 2>(P, npix, ncib, s);
            else if (a->Cout == 128) launch_wgrad1_wide<128, 4>(P, npix, ncib, s);
            else if (a->Cout == 192) launch_wgrad1_wide<192, 2>(P, npix, ncib, s);
            else launch_wgrad1_wide<256, 4>(P, npix, ncib, s);          // (Cout = 512: two column blocks of 256 per ci block)
            OFD_LAUNCH_CHECK();
            return OFD_OK;
        }
    }
    if (a->ksize == 3) {
        constexpr int LDS = 10 * 34 * 128 + 256 * 128;
        static bool attr = false;
        if (!attr) {
            OFD_HIP(hipFuncSetAttribute((const void*)conv_wgrad3_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
            OFD_HIP(hipFuncSetAttribute((const void*)conv_wgrad3_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
            OFD_HIP(hipFuncSetAttribute((const void*)conv_wgrad3_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
            OFD_HIP(hipFuncSetAttribute((const void*)conv_wgrad3_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
            attr = true;
        }
        static const bool no_phase = getenv("OFD_NO_WGRAD_PHASES") && atoi(getenv("OFD_NO_WGRAD_PHASES"));
        static const int phase_min_combos = getenv("OFD_WGRAD_PHASE_MIN_COMBOS") ? atoi(getenv("OFD_WGRAD_PHASE_MIN_COMBOS")) : 4;
        if (!no_phase && !P.no_dma && !P.in_scale && a->n_src == 1 && P.src[0].mode == 1 && a->H % 2 == 0 && a->W % 2 == 0 && combos >= phase_min_combos) {
            // up-sample conv: four phase passes on the low-resolution grid (see conv_wgrad3_kernel, PH).  The passes stage as many tiles as the
            // plain form (the low-resolution halo tile once per phase), so only the MFMA share of the time shrinks: 0.99 -> 0.85 ms (192 -> 128 at
            // 220 x 512), 0.955 -> 0.80 ms (256 -> 192 at 110 x 256), but 0.99 -> 1.07 ms for the two channel-block pairs of 128 -> 64 at full
            // resolution, which stays on the plain form
            static bool attr2 = false;
            if (!attr2) {
                OFD_HIP(hipFuncSetAttribute((const void*)conv_wgrad3_kernel<false, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
                OFD_HIP(hipFuncSetAttribute((const void*)conv_wgrad3_kernel<false, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
                OFD_HIP(hipFuncSetAttribute((const void*)conv_wgrad3_kernel<false, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
                OFD_HIP(hipFuncSetAttribute((const void*)conv_wgrad3_kernel<false, true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
                attr2 = true;
            }
            WgradParams Q = P;
            Q.H = a->H / 2; Q.W = a->W / 2;
            Q.tiles_x = cdiv(Q.W, 32); Q.tiles_y = cdiv(Q.H, 8);
            Q.src[0].mode = 0;                   // (SH, SW are the low-resolution dimensions already)
            Q.dy_s = 2;
            const int nt2 = Q.tiles_x * Q.tiles_y * Q.B;
            gx = cdiv(512, combos);
            if (gx < 1) gx = 1;
            if (gx > nt2) gx = nt2;
            for (int ph = 0; ph < 4; ++ph) {
                Q.dy_y0 = ph >> 1; Q.dy_x0 = ph & 1;
                if (ph > 0) Q.dbias = P.dbias;   // (every phase adds the column sums of its own dY samples)
                if (ph == 0) conv_wgrad3_kernel<false, true, 0><<<dim3(gx, combos), 256, LDS, s>>>(Q);
                else if (ph == 1) conv_wgrad3_kernel<false, true, 1><<<dim3(gx, combos), 256, LDS, s>>>(Q);
                else if (ph == 2) conv_wgrad3_kernel<false, true, 2><<<dim3(gx, combos), 256, LDS, s>>>(Q);
                else conv_wgrad3_kernel<false, true, 3><<<dim3(gx, combos), 256, LDS, s>>>(Q);
            }
            OFD_LAUNCH_CHECK();
            return OFD_OK;
        }
        static const int db = getenv("OFD_WGRAD3_DB") ? atoi(getenv("OFD_WGRAD3_DB")) : 1;
        // (its per-sample element offsets are 32-bit UNSIGNED arithmetic: a plane of 2^32 elements or more keeps the kernel above)
        bool small_planes = (size_t)a->H * a->W * a->Cout < (1ull << 32);
        for (int i = 0; i < a->n_src; ++i) small_planes = small_planes && (size_t)P.src[i].SH * P.src[i].SW * P.src[i].src_channels < (1ull << 32);
        if (db && !P.no_dma && !P.in_scale && small_planes) {
            // double-buffered form: ONE 8-wave workgroup per CU (2 x 75 KB of LDS), each walking ntiles / gx tiles
            constexpr int LDS_DB = 2 * (43 * 1024 + 256 * 128);
            static bool attr_db = false;
            if (!attr_db) { OFD_HIP(hipFuncSetAttribute((const void*)conv_wgrad3_db_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DB)); attr_db = true; }
            gx = 256 * db / combos;
            if (gx < 1) gx = 1;
            if (gx > ntiles) gx = ntiles;
            conv_wgrad3_db_kernel<<<dim3(gx, combos), 512, LDS_DB, s>>>(P);
            OFD_LAUNCH_CHECK();
            return OFD_OK;
        }
        gx = cdiv(512, combos);                  // two workgroups per CU fit (75.5 KB LDS each)
        if (gx < 1) gx = 1;
        if (gx > ntiles) gx = ntiles;
        if (P.no_dma) {
            if (P.in_scale) conv_wgrad3_kernel<true, false><<<dim3(gx, combos), 256, LDS, s>>>(P);
            else conv_wgrad3_kernel<false, false><<<dim3(gx, combos), 256, LDS, s>>>(P);
        } else {
            // (the prologue instantiation is over its register budget either way: 13 spilled registers through registers, 20 with the DMA)
            if (P.in_scale) conv_wgrad3_kernel<true, false><<<dim3(gx, combos), 256, LDS, s>>>(P);
            else conv_wgrad3_kernel<false, true><<<dim3(gx, combos), 256, LDS, s>>>(P);
        }
    } else {
        constexpr int LDS = 8 * 32 * 128 + 256 * 128;
        static bool attr = false;
        if (!attr) { OFD_HIP(hipFuncSetAttribute((const void*)conv_wgrad_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); attr = true; }
        conv_wgrad_kernel<1><<<dim3(gx, 1, combos), 256, LDS, s>>>(P);
    }
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

int k_conv7_wgrad(const bf16_t* x16, const bf16_t* dy, float* dw, int B, int H, int W, hipStream_t s, float* dbias, int channels) {
    const int tx = cdiv(W, 32), ty = cdiv(H, 8);
    OFD_CHECK_ARG(channels == 16 || channels == 8, "conv7_wgrad: input packed to %d channels", channels);
    static const int grid_env = getenv("OFD_CONV7_WGRAD_GRID") ? atoi(getenv("OFD_CONV7_WGRAD_GRID")) : 0;
    int grid = tx * ty * B;
    const int cap = grid_env > 0 ? grid_env : (channels == 8 ? 512 : 768);       // (the 8-channel form: two workgroups per CU at 128 registers, all resident: 0.45 ms at 512, 0.47 at 768, 0.58 at 1024)
    if (grid > cap) grid = cap;
    if (channels == 8) conv7_wgrad_kernel<8><<<grid, 512, 0, s>>>(x16, dy, dw, B, H, W, tx, ty, dbias);
    else conv7_wgrad_kernel<16><<<grid, 512, 0, s>>>(x16, dy, dw, B, H, W, tx, ty, dbias);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

int k_channel_sum(const bf16_t* dy, float* out, size_t npix, int C, hipStream_t s) {
    OFD_CHECK_ARG(C % 8 == 0 && C <= 2048, "channel_sum: C=%d", C);
    const int rows = 256 / (C / 8 < 256 ? C / 8 : 256);
    size_t grid = (npix + (size_t)rows * 16 - 1) / ((size_t)rows * 16);      // >= 16 pixels per thread ...
    if (grid > 512) grid = 512;                                             // ... and at most 512 atomics per channel
    if (grid < 1) grid = 1;
    channel_sum_kernel<<<(unsigned)grid, 256, 0, s>>>(dy, out, npix, C);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

int k_wt_transpose_batched(const ofd_weight_prep_desc* d_descs, int n, int total_blocks, hipStream_t s) {
    wt_transpose_batched_kernel<<<total_blocks, 256, 0, s>>>(d_descs, n, total_blocks);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

int k_wgrad_finish(const float* acc, const float* w_raw, float* dst, int Cout, int Cin, int Cin_pad, int ksize, float ws_eps, int unshuffle,
                   int accumulate, hipStream_t s) {
    wgrad_finish_kernel<<<Cout, 256, 0, s>>>(acc, w_raw, dst, Cout, Cin, Cin_pad, ksize, ws_eps, unshuffle, accumulate);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

int k_grad_scatter(const bf16_t* D, int Ctot, int ch_off, bf16_t* dst, int C, int B, int H, int W, int mode, int p1, int p2, int accumulate,
                   hipStream_t s) {
    OFD_CHECK_ARG(C % 8 == 0 && ch_off % 8 == 0, "grad_scatter: channel window");
    const size_t total = (size_t)B * (mode == 1 ? H / 2 : H) * (mode == 1 ? W / 2 : W) * (C / 8);
    // (one 16-byte unit per thread: a read + write kernel without a per-thread preamble runs best uncapped)
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (mode == 1) grad_scatter_kernel<1><<<grid, 256, 0, s>>>(D, Ctot, ch_off, dst, C, B, H, W, p1, p2, accumulate);
    else if (mode == 2) grad_scatter_kernel<2><<<grid, 256, 0, s>>>(D, Ctot, ch_off, dst, C, B, H, W, p1, p2, accumulate);
    else grad_scatter_kernel<0><<<grid, 256, 0, s>>>(D, Ctot, ch_off, dst, C, B, H, W, p1, p2, accumulate);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

OFD_DET_DEFINE_SETTER(det_set_ctx_conv_bwd)

}  // namespace ofd

using namespace ofd;

extern "C" int ofd_conv_dgrad_weight_prep(const void* w_fwd, void* w_t, int Cout, int Cin, int ksize, void* stream) {
    OFD_CHECK_ARG(w_fwd && w_t && Cout % 8 == 0 && Cin % 8 == 0, "dgrad_weight_prep: bad argument");
    return k_wt_transpose((const bf16_t*)w_fwd, (bf16_t*)w_t, ksize * ksize, Cin, Cout, (hipStream_t)stream);
}
extern "C" int ofd_conv_wgrad(const ofd_conv_args* fwd, const void* dy, float* dw_acc, void* stream) {
    return k_conv_wgrad(fwd, (const bf16_t*)dy, dw_acc, (hipStream_t)stream, nullptr);
}
extern "C" int ofd_conv7_wgrad(const void* x16, const void* dy, float* dw_acc, int B, int H, int W, void* stream) {
    OFD_CHECK_ARG(x16 && dy && dw_acc, "conv7_wgrad: null argument");
    return k_conv7_wgrad((const bf16_t*)x16, (const bf16_t*)dy, dw_acc, B, H, W, (hipStream_t)stream, nullptr, 16);
}
extern "C" int ofd_conv_wgrad_finish(const float* dw_acc, const float* w_oihw, float* dst_oihw, int Cout, int Cin, int Cin_pad, int ksize,
                                     float ws_eps, int unshuffle, int accumulate, void* stream) {
    OFD_CHECK_ARG(dw_acc && w_oihw && dst_oihw, "wgrad_finish: null argument");
    return k_wgrad_finish(dw_acc, w_oihw, dst_oihw, Cout, Cin, Cin_pad, ksize, ws_eps, unshuffle, accumulate, (hipStream_t)stream);
}
extern "C" int ofd_grad_scatter(const void* D, int Ctot, int ch_off, void* dst, int C, int B, int H, int W, int mode, int p1, int p2,
                                int accumulate, void* stream) {
    OFD_CHECK_ARG(D && dst, "grad_scatter: null argument");
    return k_grad_scatter((const bf16_t*)D, Ctot, ch_off, (bf16_t*)dst, C, B, H, W, mode, p1, p2, accumulate, (hipStream_t)stream);
}
extern "C" int ofd_channel_sum(const void* dy, float* out, size_t npix, int C, void* stream) {
    OFD_CHECK_ARG(dy && out, "channel_sum: null argument");
    return k_channel_sum((const bf16_t*)dy, out, npix, C, (hipStream_t)stream);
}
