// 3x3 convolution, Cin = Cout = 64 (the eight full-resolution and four half-resolution C -> C convs of the
// UNet, DD:190-214): "register-window" implicit GEMM.
//
// The generic kernel (conv_igemm.hip) reads 0.75 - 1.0 LDS operand fragments per MFMA (weights AND pixels come
// from LDS for every tap) and its 64-channel variants run LDS-bound (tools/probe/lds_mfma_probe.hip: MFMA issue
// drops 15 - 20 % at that ratio, and the ping-pong variant measured 40 % MFMA issue in its MFMA phase).  Here:
//   * the WEIGHTS live in registers for the whole launch: a wave owns 32 output channels x all 64 input
//     channels x 9 taps = 36 fragments = 144 VGPRs (persistent workgroups walk the tiles);
//   * a pixel fragment of halo row rr serves the output rows rr, rr-1, rr-2 (ky = 0, 1, 2) of the wave's
//     four rows: 18 LDS reads per 36 MFMAs, and no weight reads at all;
//   * the input tile (10 x 34 pixels x 64 channels) goes global -> LDS directly (global_load_lds_dwordx4, no
//     staging registers), pixel-major with an XOR swizzle of the 16-byte units so that the loads are fully
//     coalesced and the fragment reads conflict-free; the next tile's loads are issued before the epilogue;
//   * GroupNorm-affine + SiLU prologue (block2 convs) and the zero padding are applied in place in LDS.
// Measured (B=16, 440x1024, 531 GFLOP): plain conv 0.62 ms = 853 TF/s against 0.65 ms for the ping-pong kernel, but
// 0.91 vs 0.84 ms with the prologue, so the ping-pong kernel stays the default and this one is opt-in
// (OFD_CONV_RW=1).  What the experiment established (ablation with OFD_CONV_DBG-style switches): the MFMA loop alone
// runs at 1830 TF/s -- the practical peak of tools/probe/lds_mfma_probe.hip -- so LDS operand traffic is NOT what
// holds the 64-channel convs at ~0.8 PF; epilogue, stores and load waits simply add up serially because all eight
// waves of the one resident workgroup are in the same phase (144 weight + 64 accumulator registers allow only two
// waves per SIMD).  A two-group ping-pong of this kernel needs ~300 live registers and spilled into its MFMA loop.
// Workgroup = 8 waves = (4 row groups of 4 rows) x (2 halves of the 64 output channels) on a 16 x 32 pixel tile, one
// persistent workgroup per CU, two LDS tile buffers: the loads of tile t+1 are issued before the MFMAs of tile t.
#include "blocks.h"
#include "conv_params.h"
#include "mfma_util.h"

namespace ofd {

constexpr int RW_TH = 16, RW_TW = 32, RW_IH = RW_TH + 2, RW_IW = RW_TW + 2, RW_NPIX = RW_IH * RW_IW;   // 612
constexpr int RW_THREADS = 512, RW_WAVES = 8;
constexpr int RW_NLOAD = (RW_NPIX * 8 + 63) / 64;                     // wave-level 1 KB load instructions per tile (77)
constexpr int RW_XBYTES = RW_NLOAD * 1024;
constexpr int RW_LDS = 2 * RW_XBYTES + 512;                           // two tile buffers + prologue scale | shift table
constexpr int RW_STORES = 8;                                          // epilogue stores per lane and tile (always issued)

__device__ uint4 rw_sink[64];                                         // where the stores of out-of-image rows go

// swizzle key = column of the pixel inside its tile row: the same for every row, so a fragment address is
// (compile-time row offset) + (one of three per-lane column terms) and nothing has to be kept per row
__device__ __forceinline__ int rw_swz(int tx) { return (tx >> 1) & 7; }
__device__ __forceinline__ float rw_silu(float y) { return y * __builtin_amdgcn_rcpf(1.0f + __expf(-y)); }

__global__ void __launch_bounds__(RW_THREADS, 1) conv3x3_c64_rw_kernel(const ConvParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* aff = (float*)(smem + 2 * RW_XBYTES);      // [64 scale | 64 shift] of the current sample
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int rg = wave >> 1, nt = wave & 1;          // 4 row groups of 4 rows x 2 halves of the output channels
    const int tiles_y = (P.H + RW_TH - 1) / RW_TH, tpi = P.tiles_x * tiles_y, ntiles = tpi * P.B;
    const int tpi8 = P.tiles_x * P.tiles_y;           // GroupNorm partial sums keep the 8-row tile layout of the generic kernel
    const bf16_t* src = P.src[0].ptr;
    const int sch = P.src[0].src_channels, soff = P.src[0].ch_offset;

    // weights: [tap][ci/8][co][8] bf16; fragment (tap, k-step ks) of this wave's 32 output channels
    bf16x8 wf[9][4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            wf[tap][ks] = *(const bf16x8*)(P.weight + (((size_t)tap * 8 + ks * 2 + half) * 64 + nt * 32 + l31) * 8);

    auto issue = [&](int t, unsigned char* xs) {       // input tile of tile t: straight into LDS, asynchronous
        const int b = t / tpi, t_in = t - b * tpi;
        const int oy0 = (t_in / P.tiles_x) * RW_TH, ox0 = (t_in % P.tiles_x) * RW_TW;
        const bf16_t* base = src + (size_t)b * P.H * P.W * sch + soff;
        for (int q = wave; q < RW_NLOAD; q += RW_WAVES) {
            const int s = q * 64 + lane, p = min(s >> 3, RW_NPIX - 1), v = s & 7;
            const int ty = p / RW_IW, tx = p - ty * RW_IW;
            const int iy = min(max(oy0 - 1 + ty, 0), P.H - 1), ix = min(max(ox0 - 1 + tx, 0), P.W - 1);
            const int u = v ^ rw_swz(tx);
            __builtin_amdgcn_global_load_lds(base + ((size_t)iy * P.W + ix) * sch + u * 8,
                                             (__attribute__((address_space(3))) void*)(xs + q * 1024), 16, 0, 0);
        }
    };

    int t = blockIdx.x, cur = 0;
    if (t < ntiles) issue(t, smem);
    int aff_b = -1;
    bool first = true;
    for (; t < ntiles; t += gridDim.x, cur ^= 1) {
        unsigned char* xs = smem + cur * RW_XBYTES;
        const int b = t / tpi, t_in = t - b * tpi;
        const int ty16 = t_in / P.tiles_x, txi = t_in % P.tiles_x;
        const int oy0 = ty16 * RW_TH, ox0 = txi * RW_TW;
        // this tile's loads were issued before the previous tile's stores: memory operations retire in order, so
        // "at most RW_STORES outstanding" means the loads have landed while the stores may still be in flight
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        first = false;
        __syncthreads();                               // tile t is complete in LDS; everyone is done with the other buffer
        if (t + (int)gridDim.x < ntiles) issue(t + gridDim.x, smem + (cur ^ 1) * RW_XBYTES);   // flies during this tile's MFMAs
        if (P.in_scale && b != aff_b) {
            if (tid < 128) aff[tid] = (tid < 64) ? P.in_scale[(size_t)b * 64 + tid] : P.in_shift[(size_t)b * 64 + tid - 64];
            aff_b = b;
            __syncthreads();
        }
        // in place: prologue y = SiLU(x * scale + shift) (DD:186-187 of the previous Block), then the zero padding
        const bool border = oy0 == 0 || ox0 == 0 || oy0 + RW_TH >= P.H || ox0 + RW_TW >= P.W;
        if (P.in_scale || border) {
            for (int s = tid; s < RW_NPIX * 8; s += RW_THREADS) {
                const int p = s >> 3, v = s & 7, ty = p / RW_IW, tx = p - ty * RW_IW;
                const int iy = oy0 - 1 + ty, ix = ox0 - 1 + tx;
                const bool inside = iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
                u32x4* cell = (u32x4*)(xs + s * 16);
                if (!inside) {
                    u32x4 z = {0u, 0u, 0u, 0u};
                    *cell = z;
                } else if (P.in_scale) {
                    const int u = v ^ rw_swz(tx);
                    const float4 s0 = *(const float4*)&aff[u * 8], s1 = *(const float4*)&aff[u * 8 + 4];
                    const float4 h0 = *(const float4*)&aff[64 + u * 8], h1 = *(const float4*)&aff[64 + u * 8 + 4];
                    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                    const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                    u32x4 x = *cell;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float lo = rw_silu(bf2f((bf16_t)(x[j] & 0xffffu)) * sc[2 * j] + sh[2 * j]);
                        const float hi = rw_silu(bf2f((bf16_t)(x[j] >> 16)) * sc[2 * j + 1] + sh[2 * j + 1]);
                        x[j] = f2bf2(lo, hi);
                    }
                    *cell = x;
                }
            }
            __syncthreads();
        }
        f32x16 acc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[r][k] = 0.0f;
        // per-lane column terms of the fragment addresses: unit (2 ks + half) ^ key = (2 ks ^ (key & 6)) + (half ^ (key & 1))
        const unsigned char* xcol[3];
        int xkey[3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int key = rw_swz(l31 + kx);
            xkey[kx] = key & 6;
            xcol[kx] = xs + (l31 + kx) * 128 + ((half ^ (key & 1)) * 16);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int rr = 0; rr < 6; ++rr) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const bf16x8 xf = *(const bf16x8*)(xcol[kx] + (rg * 4 + rr) * RW_IW * 128 + (((ks * 2) ^ xkey[kx]) * 16));
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
                        if (rr - ky >= 0 && rr - ky < 4)
                            acc[rr - ky] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ky * 3 + kx][ks], xf, acc[rr - ky], 0, 0, 0);
                }
            }
        }

        // ---- epilogue: bias, bf16, 16-byte stores (always RW_STORES per lane), GroupNorm partial sums of the values as stored
        float stat[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) stat[i] = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oy = oy0 + rg * 4 + r, ox = ox0 + l31;
            const bool ok = oy < P.H && ox < P.W;
            const size_t pix = ((size_t)b * P.H + min(oy, P.H - 1)) * P.W + min(ox, P.W - 1);
            uint2 q[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (P.bias) bv = *(const float4*)(P.bias + nt * 32 + 8 * g + 4 * half);
                const float v0 = acc[r][4 * g] + bv.x, v1 = acc[r][4 * g + 1] + bv.y;
                const float v2 = acc[r][4 * g + 2] + bv.z, v3 = acc[r][4 * g + 3] + bv.w;
                q[g] = make_uint2(f2bf2(v0, v1), f2bf2(v2, v3));
                if (P.gn_partial && ok) {
                    const float q0 = bf2f((bf16_t)(q[g].x & 0xffffu)), q1 = bf2f((bf16_t)(q[g].x >> 16));
                    const float q2 = bf2f((bf16_t)(q[g].y & 0xffffu)), q3 = bf2f((bf16_t)(q[g].y >> 16));
                    stat[g * 2] += (q0 + q1) + (q2 + q3);
                    stat[g * 2 + 1] += (q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const auto rx = __builtin_amdgcn_permlane32_swap(q[g].x, q[g + 1].x, false, false);
                const auto ry = __builtin_amdgcn_permlane32_swap(q[g].y, q[g + 1].y, false, false);
                uint4* dst = ok ? (uint4*)(P.out + pix * 64 + nt * 32 + 8 * g + 8 * half) : &rw_sink[lane];
                *dst = make_uint4(rx[0], ry[0], rx[1], ry[1]);
            }
        }
        const int ty8 = ty16 * 2 + (rg >> 1);
        if (P.gn_partial && ty8 < P.tiles_y) {       // [b][8-row tile][wave slot][8 octets][2]: this wave's four octets, zeros for the other half
            const int slot = (ty8 * P.tiles_x + txi) * 4 + (rg & 1) * 2 + nt;       // (element address: gn_partial_index, conv_params.h)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = stat[i];
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                stat[i] = v;
            }
            if (lane < 8) {
                float v = stat[0];
#pragma unroll
                for (int i = 1; i < 8; ++i) v = (lane == i) ? stat[i] : v;
                P.gn_partial[gn_partial_index(b, tpi8 * 4, slot, 8, nt * 4 + (lane >> 1)) + (lane & 1)] = v;
                P.gn_partial[gn_partial_index(b, tpi8 * 4, slot, 8, (1 - nt) * 4 + (lane >> 1)) + (lane & 1)] = 0.0f;
            }
        }
    }
}

int launch_conv3x3_c64_rw(const ConvParams& P, hipStream_t s) {
    static bool attr = false;
    if (!attr) { OFD_HIP(hipFuncSetAttribute((const void*)conv3x3_c64_rw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RW_LDS)); attr = true; }
    const int ntiles = P.tiles_x * cdiv(P.H, RW_TH) * P.B;
    static int rw_grid = -1;
    if (rw_grid < 0) { const char* e = getenv("OFD_RW_GRID"); rw_grid = e ? atoi(e) : 256; }       // one persistent workgroup per CU
    conv3x3_c64_rw_kernel<<<ntiles < rw_grid ? ntiles : rw_grid, RW_THREADS, RW_LDS, s>>>(P);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

}  // namespace ofd
