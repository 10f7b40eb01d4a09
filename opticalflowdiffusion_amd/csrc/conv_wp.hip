// 3x3 implicit-GEMM convolution with WAVE-PRIVATE weights (gfx950, MFMA 32x32x16 bf16, fp32 accumulate, NHWC bf16).
//
// Same operation, prologue / epilogue fusions and data layouts as conv_igemm.hip (the reference's F.conv2d calls at
// denoising_diffusion.py:114,200,297,339,354 with the GroupNorm-apply + SiLU of DD:181-187 in the loader); what changes is who
// owns which operand.  In conv_igemm.hip the four waves of a workgroup share a weight slab in LDS: one workgroup barrier, one
// register -> LDS weight copy and ~7 VALU / SALU instructions per MFMA -- measured 43 % MFMA-busy (profiles/r02_sq_counters.json).
// Here
//   * a wave owns a 32-output-channel slice and ALL 8 rows of a 8 x 32 pixel block: its MFMA A operand (weights) is then private
//     to the wave and is read straight from global memory / L2 into registers as ONE 16-byte load per lane per fragment
//     (weights are stored [tap][Cin/8][Cout][8], an A fragment is two contiguous 512-byte runs), six fragments ahead of its use:
//     no weight traffic through LDS, no per-tap barrier, no staging VALU;
//   * only the input tile (+halo) goes through LDS, 32 channels at a time, DOUBLE buffered: the tile of chunk k+1 is fetched
//     while chunk k computes and written (after the prologue transform) behind its first MFMAs: ONE workgroup barrier per
//     144 MFMAs per wave, at which nobody waits for memory;
//   * a pixel fragment read from LDS feeds the three kernel rows: 10 row fragments per (kx, k-step) for 24 MFMAs
//     (0.42 ds_read_b128 per MFMA against 0.75), every read = one base register + immediate.
// Workgroup = 4 waves = NS channel slices x PH row blocks: <4,1> = 8 x 32 pixels x 128 channels, <2,2> = 16 x 32 pixels x 64 channels.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "conv_params.h"
#include "mfma_util.h"

#ifndef OFD_WP_PEEL
#define OFD_WP_PEEL 1
#endif
#ifndef OFD_WP_DOT2
#define OFD_WP_DOT2 1
#endif

// OFD_WP_STAMPS=1 (diagnostic builds only, tools/probe/wp_stamps.py): every wave of conv3x3_wp_kernel records s_memtime at its phase
// boundaries into a buffer of its own that nothing else reads; the production build compiles none of it
#ifndef OFD_WP_STAMPS
#define OFD_WP_STAMPS 0
#endif

namespace ofd {

namespace wp {

#if OFD_WP_STAMPS
constexpr int STAMP_SLOTS = 16, STAMP_WAVES = 1 << 16;
__device__ unsigned long long g_wp_stamps[STAMP_SLOTS * STAMP_WAVES];
#define WP_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); stamps[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define WP_DRAIN_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define WP_STAMP(i) do { } while (0)
#define WP_DRAIN_VM() do { } while (0)
#endif

constexpr int CK = 32, NC = CK / 8, IW = 34, TW = 32, RING = 6, FRAGS = 18;   // 18 weight fragments per 32-channel chunk

template <int NS, int PH>
struct Cfg {
    static constexpr int NTHREADS = 64 * NS * PH;            // 4 waves; <8,1>: 8 waves = 256 output channels over one staged tile
    static constexpr int BN = 32 * NS, ROWS = 8 * PH, IH = ROWS + 2, NPIX = IH * IW;
    static constexpr int US = (NPIX + 1) * 16;               // octet row of the unit-major tile [NC][NPIX + 1][16 B]; NPIX + 1 is odd
    static constexpr int XB = NC * US;
    static constexpr int LDS_BYTES = 2 * XB;
    static constexpr int XPT = (NPIX * NC + NTHREADS - 1) / NTHREADS;
    static_assert(NS * PH == 4 || (NS == 8 && PH == 1), "4 waves, or 8 channel slices");
    static_assert((NPIX + 1) % 2 == 1, "odd slot count keeps the staging writes of a pixel's octets on distinct banks");
};

__device__ __forceinline__ float silu_f(float y) { return y * __builtin_amdgcn_rcpf(1.0f + __expf(-y)); }

typedef __attribute__((ext_vector_type(4))) unsigned int u4;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bf16x8 as_frag(u4 v) { return __builtin_bit_cast(bf16x8, v); }

// butterfly reduction of 8 per-lane values over the wave: afterwards the lanes with (lane & 7) == 0 ... hold in v[0] the total
// of value index (lane >> 3) (same scheme as conv_igemm.hip's WaveReduce)
__device__ __forceinline__ void wave_reduce8(float (&v)[8]) {
    // r04: on the VALU's own cross-lane paths (v_permlane32_swap / v_permlane16_swap exchange two values between half-waves / 16-lane rows
    // in one instruction, DPP inside a row) instead of ten ds_bpermute round trips through the LDS pipe
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 4; ++i) {                     // lanes < 32 keep value i, lanes >= 32 value i + 4: each adds what the other half holds of it
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 4]), false, false);
        v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {                     // rows 0, 2 keep value i, rows 1, 3 value i + 2
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 2]), false, false);
        v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    {
        const bool up = (lane & 8) != 0;              // lanes 0-7 of a row keep value 0, lanes 8-15 value 1
        const float send = up ? v[0] : v[1], keep = up ? v[1] : v[0];
        v[0] = keep + __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(send), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
    }
    // the eight lanes of a group: i + (7 - i), then pairs inside a quad, then the two quads' lanes 0 / 2: lane 8 k holds the total
    v[0] += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v[0]), 0x141 /* row_half_mirror */, 0xf, 0xf, false));
    v[0] += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v[0]), 0xb1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false));
    v[0] += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v[0]), 0x4e /* quad_perm:[2,3,0,1] */, 0xf, 0xf, false));
}

// PRO: the GroupNorm-affine + SiLU prologue is compiled in (P.in_scale != nullptr).  The chunk body below is ONE basic block (no
// run-time branch between its 144 MFMAs), so that the scheduler can put the LDS reads of a group behind the MFMAs of the previous one.
template <int NS, int PH, bool PRO>
__global__ void __launch_bounds__(64 * NS * PH, (NS * PH == 4) ? 2 : 1) conv3x3_wp_kernel(const ConvParams P) {
    using C = Cfg<NS, PH>;
    constexpr int NTHREADS = C::NTHREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int ns = wave % NS, ph = wave / NS;
#if OFD_WP_STAMPS
    unsigned long long stamps[STAMP_SLOTS] = {};
    stamps[14] = __builtin_amdgcn_s_memrealtime();
    WP_STAMP(0);
#endif

    // XCD-aware tile order (blocks that share an XCD get a contiguous run of tiles: halo rows of neighbours hit one L2)
    const int tiles_y = (P.H + C::ROWS - 1) / C::ROWS;
    const int tpi = P.tiles_x * tiles_y, ntiles = tpi * P.B;
    int tile = blockIdx.x, cy = blockIdx.y;
    if (P.cy_fast) {                                  // block j -> XCD j % 8, channel block (j / 8) % NY, tile slot j / 8 / NY
        const int ny = P.Cout / C::BN, j = blockIdx.x, g = j >> 3;
        cy = g % ny;
        tile = (g / ny) * 8 + (j & 7);
        if (tile >= ntiles) return;
    }
    if (ntiles >= 8) {
        const int q = ntiles / 8, r = ntiles % 8, xcd = tile % 8, idx = tile / 8;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int b = tile / tpi, t_in = tile % tpi;
    const int oy0 = (t_in / P.tiles_x) * C::ROWS, ox0 = (t_in % P.tiles_x) * TW;
    const int n0 = cy * C::BN;
    const int cb = n0 + 32 * ns;                      // this wave's 32 output channels

    // ---- weights: buffer loads, per-lane offset fixed for the launch, per-fragment offset scalar
    const int cin8 = P.Cin_total / 8, n32 = P.total_chunks * 2;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)P.weight, 0, 9 * P.Cin_total * P.Cout * 2, 0x00020000);
    const int w_lane = (half * P.Cout + cb + l31) * 16;
    const int w_row = P.Cout * 16;                    // bytes per [Cin/8] row
    auto load_w = [&](int kc, int fi) -> u4 {         // fragment fi = (ks, kx, ky) of 32-channel chunk kc
        const int g = fi / 3, ky = fi % 3, ks = g / 3, kx = g % 3;
        const int row = (ky * 3 + kx) * cin8 + kc * NC + ks * 2;
        return __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, w_lane, row * w_row, 0));
    };
    u4 ring[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) ring[i] = load_w(0, i);

    // ---- input staging map (invariant over the chunks): unit u = tid + i * 256 -> octet c8 = tid % 4, tile pixel p = u / 4
    const int c8 = tid % NC;
    int pyx[C::XPT];            // clamped source row << 16 | clamped source column (of the OUTPUT-resolution image)
    unsigned okmask = 0;
#pragma unroll
    for (int i = 0; i < C::XPT; ++i) {
        const int p = min(tid / NC + i * (NTHREADS / NC), C::NPIX - 1);
        const int ty = p / IW, tx = p - ty * IW;
        const int iy = oy0 - 1 + ty, ix = ox0 - 1 + tx;
        const bool ok = iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
        okmask |= (ok ? 1u : 0u) << i;
        pyx[i] = (min(max(iy, 0), P.H - 1) << 16) | min(max(ix, 0), P.W - 1);
    }

    int src_i = 0, src_first = 0;                     // source that owns 64-channel chunk kc >> 1, and its first 64-channel chunk
    auto load_x = [&](int kc, u4 (&xs)[C::XPT]) {
        const int k64 = kc >> 1;
        while (k64 >= src_first + P.src[src_i].chunks) {
            src_first += P.src[src_i].chunks;
            ++src_i;
        }
        const ConvSrcDev& S = P.src[src_i];
        const bf16_t* base = S.ptr + (size_t)b * S.SH * S.SW * S.src_channels + S.ch_offset + (k64 - src_first) * 64 + (kc & 1) * CK + c8 * 8;
        const int up = S.mode == 1 ? 1 : 0;           // nearest x2 up-sampling of the source (DD:91) is a shift of the coordinates
#pragma unroll
        for (int i = 0; i < C::XPT; ++i) {
            const int sy = (pyx[i] >> 16) >> up, sx = (pyx[i] & 0xffff) >> up;
            xs[i] = *(const u4*)(base + ((size_t)sy * S.SW + sx) * S.src_channels);
        }
    };
    auto write_x = [&](int kc, const u4 (&xs)[C::XPT], unsigned char* xbuf) {
        float ps[8], pb[8];
        if constexpr (PRO) {
            const float* sp = P.in_scale + (size_t)b * P.Cin_total + kc * CK + c8 * 8;
            const float* bp = P.in_shift + (size_t)b * P.Cin_total + kc * CK + c8 * 8;
            *(float4*)&ps[0] = *(const float4*)sp; *(float4*)&ps[4] = *(const float4*)(sp + 4);
            *(float4*)&pb[0] = *(const float4*)bp; *(float4*)&pb[4] = *(const float4*)(bp + 4);
        }
#pragma unroll
        for (int i = 0; i < C::XPT; ++i) {
            const int p = min(tid / NC + i * (NTHREADS / NC), C::NPIX - 1);
            u4 v = xs[i];
            if constexpr (PRO) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float lo = silu_f(bf2f((bf16_t)(v[j] & 0xffffu)) * ps[2 * j] + pb[2 * j]);
                    const float hi = silu_f(bf2f((bf16_t)(v[j] >> 16)) * ps[2 * j + 1] + pb[2 * j + 1]);
                    v[j] = f2bf2(lo, hi);
                }
            }
            const bool ok = (okmask >> i) & 1u;       // zero padding is applied AFTER the prologue (DD:181-187 -> DD:114)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ok ? v[j] : 0u;
            *(u4*)(xbuf + c8 * C::US + p * 16) = v;
        }
    };

    f32x16 acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[r][k] = 0.0f;

    u4 xs[C::XPT];
    WP_STAMP(1);
    load_x(0, xs);
    WP_DRAIN_VM();
    WP_STAMP(2);
    write_x(0, xs, smem);
    WP_STAMP(3);

    const int xrow_off = half * C::US + (8 * ph * IW + l31) * 16;
    // one 32-channel chunk: 144 MFMAs per wave between two workgroup barriers.  LAST (the peeled final chunk) fetches and stages
    // nothing: there is no next chunk (a 64-channel layer has two chunks -- re-staging the last one, as the un-peeled loop did to stay
    // one basic block, was a third of its prologue arithmetic and of its input reads; OFD_WP_PEEL=0 builds that form)
    auto chunk = [&](const int kc, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const int kn = LAST ? kc : kc + 1;
        if constexpr (!LAST) load_x(kn, xs);
        __syncthreads();                              // tile kc complete; every wave is done reading the other buffer (chunk kc-1)
        WP_STAMP(LAST ? 8 : 4);
        const unsigned char* xrow = smem + (kc & 1) * C::XB + xrow_off;
        unsigned char* xnext = smem + ((kc + 1) & 1) * C::XB;
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            const int ks = g / 3, kx = g % 3;
            bf16x8 x[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) x[j] = *(const bf16x8*)(xrow + (j * IW + kx) * 16 + ks * 2 * C::US);
            bf16x8 a[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int fi = g * 3 + ky;
                a[ky] = as_frag(ring[fi % RING]);
                if (fi + RING < FRAGS) ring[fi % RING] = load_w(kc, fi + RING);
                else if constexpr (!LAST) ring[fi % RING] = load_w(kn, fi + RING - FRAGS);
            }
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ky], x[r + ky], acc[r], 0, 0, 0);
            if constexpr (!LAST) { if (g == 1) { WP_STAMP(5); write_x(kn, xs, xnext); WP_STAMP(6); } }
        }
        WP_STAMP(LAST ? 9 : 7);
    };
#if OFD_WP_PEEL
    for (int kc = 0; kc < n32 - 1; ++kc) chunk(kc, std::false_type{});
    chunk(n32 - 1, std::true_type{});
#else
    for (int kc = 0; kc < n32; ++kc) {
        const int kn = min(kc + 1, n32 - 1);          // past the end: the last chunk again (fetched and staged into the buffer nobody reads any more)
        load_x(kn, xs);
        __syncthreads();
        const unsigned char* xrow = smem + (kc & 1) * C::XB + xrow_off;
        unsigned char* xnext = smem + ((kc + 1) & 1) * C::XB;
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            const int ks = g / 3, kx = g % 3;
            bf16x8 x[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) x[j] = *(const bf16x8*)(xrow + (j * IW + kx) * 16 + ks * 2 * C::US);
            bf16x8 a[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int fi = g * 3 + ky;
                a[ky] = as_frag(ring[fi % RING]);
                ring[fi % RING] = (fi + RING < FRAGS) ? load_w(kc, fi + RING) : load_w(kn, fi + RING - FRAGS);
            }
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ky], x[r + ky], acc[r], 0, 0, 0);
            if (g == 1) write_x(kn, xs, xnext);
        }
    }
#endif

    // ---- epilogue: bias, residual forms, bf16 16-byte stores (one v_permlane32_swap per dword pairs two register quads),
    //      GroupNorm partial sums of the values as stored
    float stat[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) stat[i] = 0.0f;
    const bool second = P.split > 0 && cb >= P.split;
    bf16_t* const o_base = second ? P.out2 : P.out;
    const bf16_t* const r_base = second ? P.residual2 : P.residual;
    const int o_stride = P.split > 0 ? (second ? P.Cout - P.split : P.split) : P.Cout;
    const int o_c0 = cb - (second ? P.split : 0);
    float4 bias4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias4[g] = P.bias ? *(const float4*)(P.bias + cb + 8 * g + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (P.pool2) {
        // 2x2 sum-pooled output (ConvParams::pool2): rows 2i, 2i + 1 are two accumulator tiles of this wave, pixels l31, l31 ^ 1 neighbouring
        // lanes; the even lane stores pooled pixel ((oy0 + 8 ph) / 2 + i, (ox0 + l31) / 2) of the (H/2, W/2) tensor (H, W even: a pooled
        // pixel is inside the image with all four of its pixels or with none)
        const int PH2 = P.H >> 1, PW2 = P.W >> 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int py = ((oy0 + 8 * ph) >> 1) + i, px = (ox0 + l31) >> 1;
            const bool ok = py < PH2 && px < PW2 && !(l31 & 1) && !(P.dbg & 16);
            const size_t pix = ((size_t)b * PH2 + min(py, PH2 - 1)) * PW2 + min(px, PW2 - 1);
            uint2 q[4], rr[4];
            if (r_base) {
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    const uint4 t4 = *(const uint4*)(r_base + pix * o_stride + o_c0 + 8 * g + 8 * half);
                    const auto sx = __builtin_amdgcn_permlane32_swap(t4.x, t4.z, false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(t4.y, t4.w, false, false);
                    rr[g] = make_uint2(sx[0], sy[0]);
                    rr[g + 1] = make_uint2(sx[1], sy[1]);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[k] = acc[2 * i][4 * g + k] + acc[2 * i + 1][4 * g + k];
                    v[k] += __shfl_xor(v[k], 1, 64);
                }
                v[0] += 4.0f * bias4[g].x; v[1] += 4.0f * bias4[g].y; v[2] += 4.0f * bias4[g].z; v[3] += 4.0f * bias4[g].w;
                if (r_base) {
                    const uint2 t = rr[g];
                    v[0] += bf2f((bf16_t)(t.x & 0xffffu));
                    v[1] += bf2f((bf16_t)(t.x >> 16));
                    v[2] += bf2f((bf16_t)(t.y & 0xffffu));
                    v[3] += bf2f((bf16_t)(t.y >> 16));
                }
                q[g] = make_uint2(f2bf2(v[0], v[1]), f2bf2(v[2], v[3]));
            }
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const auto rx = __builtin_amdgcn_permlane32_swap(q[g].x, q[g + 1].x, false, false);
                const auto ry = __builtin_amdgcn_permlane32_swap(q[g].y, q[g + 1].y, false, false);
                if (ok) *(uint4*)(o_base + pix * o_stride + o_c0 + 8 * g + 8 * half) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
            }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int oy = oy0 + 8 * ph + r, ox = ox0 + l31;
        const bool ok = oy < P.H && ox < P.W && !(P.dbg & 16);
        const size_t pix = ((size_t)b * P.H + min(oy, P.H - 1)) * P.W + min(ox, P.W - 1);
        uint2 q[4], ra[4], rr[4];
        if (P.residual_b) {      // second plain residual: into the accumulators first (its registers are free again before the loads below)
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const uint4 t4 = *(const uint4*)(P.residual_b + pix * o_stride + o_c0 + 8 * g + 8 * half);
                const auto sx = __builtin_amdgcn_permlane32_swap(t4.x, t4.z, false, false);
                const auto sy = __builtin_amdgcn_permlane32_swap(t4.y, t4.w, false, false);
                acc[r][4 * g] += bf2f((bf16_t)(sx[0] & 0xffffu));
                acc[r][4 * g + 1] += bf2f((bf16_t)(sx[0] >> 16));
                acc[r][4 * g + 2] += bf2f((bf16_t)(sy[0] & 0xffffu));
                acc[r][4 * g + 3] += bf2f((bf16_t)(sy[0] >> 16));
                acc[r][4 * g + 4] += bf2f((bf16_t)(sx[1] & 0xffffu));
                acc[r][4 * g + 5] += bf2f((bf16_t)(sx[1] >> 16));
                acc[r][4 * g + 6] += bf2f((bf16_t)(sy[1] & 0xffffu));
                acc[r][4 * g + 7] += bf2f((bf16_t)(sy[1] >> 16));
            }
        }
        if (P.res_act) {
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const uint4 t4 = *(const uint4*)(P.res_act + pix * P.Cout + cb + 8 * g + 8 * half);
                const auto sx = __builtin_amdgcn_permlane32_swap(t4.x, t4.z, false, false);
                const auto sy = __builtin_amdgcn_permlane32_swap(t4.y, t4.w, false, false);
                ra[g] = make_uint2(sx[0], sy[0]);
                ra[g + 1] = make_uint2(sx[1], sy[1]);
            }
        }
        if (r_base) {
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const uint4 t4 = *(const uint4*)(r_base + pix * o_stride + o_c0 + 8 * g + 8 * half);
                const auto sx = __builtin_amdgcn_permlane32_swap(t4.x, t4.z, false, false);
                const auto sy = __builtin_amdgcn_permlane32_swap(t4.y, t4.w, false, false);
                rr[g] = make_uint2(sx[0], sy[0]);
                rr[g + 1] = make_uint2(sx[1], sy[1]);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = cb + 8 * g + 4 * half;
            float v[4] = {acc[r][4 * g] + bias4[g].x, acc[r][4 * g + 1] + bias4[g].y, acc[r][4 * g + 2] + bias4[g].z, acc[r][4 * g + 3] + bias4[g].w};
            if (P.res_act) {
                const uint2 t = ra[g];
                const float4 sc = *(const float4*)(P.res_scale + (size_t)b * P.Cout + c);
                const float4 sh = *(const float4*)(P.res_shift + (size_t)b * P.Cout + c);
                v[0] += silu_f(bf2f((bf16_t)(t.x & 0xffffu)) * sc.x + sh.x);
                v[1] += silu_f(bf2f((bf16_t)(t.x >> 16)) * sc.y + sh.y);
                v[2] += silu_f(bf2f((bf16_t)(t.y & 0xffffu)) * sc.z + sh.z);
                v[3] += silu_f(bf2f((bf16_t)(t.y >> 16)) * sc.w + sh.w);
            }
            if (r_base) {
                const uint2 t = rr[g];
                v[0] += bf2f((bf16_t)(t.x & 0xffffu));
                v[1] += bf2f((bf16_t)(t.x >> 16));
                v[2] += bf2f((bf16_t)(t.y & 0xffffu));
                v[3] += bf2f((bf16_t)(t.y >> 16));
            }
            q[g] = make_uint2(f2bf2(v[0], v[1]), f2bf2(v[2], v[3]));
            if (P.gn_partial && ok) {
#if OFD_WP_DOT2
                // sums of the stored (bf16) values by packed dot products: x . (1, 1) and x . x, two elements per instruction
                const bf16x2 one = __builtin_bit_cast(bf16x2, 0x3f803f80u);
                const bf16x2 va = __builtin_bit_cast(bf16x2, q[g].x), vb = __builtin_bit_cast(bf16x2, q[g].y);
                stat[g * 2] = __builtin_amdgcn_fdot2_f32_bf16(vb, one, __builtin_amdgcn_fdot2_f32_bf16(va, one, stat[g * 2], false), false);
                stat[g * 2 + 1] = __builtin_amdgcn_fdot2_f32_bf16(vb, vb, __builtin_amdgcn_fdot2_f32_bf16(va, va, stat[g * 2 + 1], false), false);
#else
                const float q0 = bf2f((bf16_t)(q[g].x & 0xffffu)), q1 = bf2f((bf16_t)(q[g].x >> 16));
                const float q2 = bf2f((bf16_t)(q[g].y & 0xffffu)), q3 = bf2f((bf16_t)(q[g].y >> 16));
                stat[g * 2] += (q0 + q1) + (q2 + q3);
                stat[g * 2 + 1] += (q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3);
#endif
            }
        }
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
            const auto rx = __builtin_amdgcn_permlane32_swap(q[g].x, q[g + 1].x, false, false);
            const auto ry = __builtin_amdgcn_permlane32_swap(q[g].y, q[g + 1].y, false, false);
            if (ok) *(uint4*)(o_base + pix * o_stride + o_c0 + 8 * g + 8 * half) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
        }
    }

#if OFD_WP_STAMPS
    WP_STAMP(10);
    WP_DRAIN_VM();
    WP_STAMP(11);
    stamps[15] = __builtin_amdgcn_s_memrealtime();
    stamps[13] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_ID (wave / SIMD / CU / SE ...)
    {
        const unsigned wv = (blockIdx.x + gridDim.x * blockIdx.y) * (NTHREADS / 64) + wave;
        if (lane == 0 && wv < (unsigned)STAMP_WAVES)
#pragma unroll
            for (int i = 0; i < STAMP_SLOTS; ++i) g_wp_stamps[(size_t)wv * STAMP_SLOTS + i] = stamps[i];
    }
#endif
    if (P.gn_partial) {
        // slots of conv_igemm.hip (4 per 8x32 tile; element address: gn_partial_index, conv_params.h), consumed by gn_finalize: this wave owns octets
        // cb/8 .. cb/8 + 3 of the 8-row tile (ph); its sums go to slot ns, every other (slot, octet) of the workgroup's channel
        // block is written as zero by the wave whose slot it is (slots ns, ns + NS, ...)
        wave_reduce8(stat);
        const int ty8 = oy0 / 8 + ph, tiles8 = (P.H + 7) / 8;
        if constexpr (NS * PH == 8) {
            // 8 slices: every wave writes all four slots of its own four octets -- the sums into slot 0, zeros into the others
            const int slot = (lane & 31) >> 3, o = (lane & 7) >> 1, which = lane & 1;
            const float total = __shfl(stat[0], (o * 2 + which) * 8, 64);           // value index k lives in lanes 8k .. 8k+7
            if (ty8 < tiles8 && lane < 32) {
                P.gn_partial[gn_partial_index(b, tiles8 * P.tiles_x * 4, (ty8 * P.tiles_x + (t_in % P.tiles_x)) * 4 + slot, P.Cout / 8, cb / 8 + o) + which] =
                    slot == 0 ? total : 0.0f;
            }
        } else if (ty8 < tiles8) {
            constexpr int OCT = C::BN / 8;                          // octets of the workgroup's channel block
            constexpr int PER_WAVE = (4 / NS) * OCT * 2;            // floats this wave writes (32)
            // lane t < PER_WAVE writes float t of this wave's share: (slot_i, octet o, sum / sum of squares)
            const int slot_i = lane / (OCT * 2), o = (lane % (OCT * 2)) >> 1, which = lane & 1;
            const float total = __shfl(stat[0], ((o & 3) * 2 + which) * 8, 64);      // value index k lives in lanes 8k .. 8k+7
            if (lane < PER_WAVE) {
                const int slot = ns + slot_i * NS;
                const bool own = slot_i == 0 && (o >> 2) == ns;
                P.gn_partial[gn_partial_index(b, tiles8 * P.tiles_x * 4, (ty8 * P.tiles_x + (t_in % P.tiles_x)) * 4 + slot, P.Cout / 8, n0 / 8 + o) + which] =
                    own ? total : 0.0f;
            }
        }
    }
}

// OFD_CONV_WP_LDS_PAD=bytes: at least that much dynamic LDS per workgroup (> 80 KB: ONE 3x3 workgroup per CU, the other half of the CU's
// registers and LDS left to kernels of the other stream of ofd_unet_set_split_streams -- the co-residency experiment of DESIGN 4.0)
static inline int wp_lds(int need) {
    static const int pad = getenv("OFD_CONV_WP_LDS_PAD") ? atoi(getenv("OFD_CONV_WP_LDS_PAD")) : 0;
    return need > pad ? need : pad;
}

template <int NS, int PH, bool PRO>
static int launch(const ConvParams& P, hipStream_t s) {
    using C = Cfg<NS, PH>;
    static bool attr_set = false;
    if (!attr_set) {
        OFD_HIP(hipFuncSetAttribute((const void*)conv3x3_wp_kernel<NS, PH, PRO>, hipFuncAttributeMaxDynamicSharedMemorySize, wp_lds(C::LDS_BYTES)));
        attr_set = true;
    }
    const int tiles_y = (P.H + C::ROWS - 1) / C::ROWS;
    const int ntiles = P.tiles_x * tiles_y * P.B, ny = P.Cout / C::BN;
    dim3 grid(ntiles, ny);
    if (P.cy_fast) grid = dim3((ntiles + 7) / 8 * 8 * ny, 1);
    conv3x3_wp_kernel<NS, PH, PRO><<<grid, C::NTHREADS, wp_lds(C::LDS_BYTES), s>>>(P);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}


// ---- the 128-channel-block 3x3 on MFMA 16x16x32 (r03) -------------------------------------------------------------------------------
// Same workgroup (<4,1>: 8 x 32 pixels x 128 channels, a wave = 32 channels x all 8 rows), same staged tile, same weight layout, same
// bytes from LDS and L2 per FLOP -- but the product runs on v_mfma_f32_16x16x32_bf16: under this kernel's operand pattern (A from
// registers, B re-read from LDS, random data, two workgroups per CU) the chip sustains 1.87 PF on that shape against 1.62 PF on
// 32x32x16 (tools/probe/mfma_shape_probe.hip, same-box: +15 %; the part holds a higher clock on the smaller shape, MI355X guide
// "DVFS give-back" item 7).  What changes:
//   * a 32-channel chunk is ONE k-step (K = 32): per kernel column kx the wave reads 2 x 12 row fragments (16 pixels x 32 channels
//     each: lanes 0-15 octet 0, 16-31 octet 1, ...) and multiplies them with 6 weight fragments (3 kernel rows x 2 halves of its 32
//     output channels; 16 channels x 32 input channels each, lane = (channel, octet)): 96 MFMAs per kx, 288 per chunk;
//   * weights: two register sets of six fragments; the set of the next kx is fetched while this one computes;
//   * accumulators acc[row][pixel half][channel half] (4 registers each, 128 in all): lane = pixel, registers = 4 consecutive
//     channels; one v_permlane16_swap per dword pairs two 16-lane rows into 16-byte stores (a store instruction covers 16 pixels x
//     all 32 channels of the wave); GroupNorm statistics per (lane row pair, channel half) = one 8-channel group each.
// Serves the plain / prologue / GroupNorm-statistics forms (the inference step and the training forward); residual, split and pooled
// epilogues (data gradients) stay on conv3x3_wp_kernel<4,1>.  OFD_CONV_WP16=0 switches it off.
typedef __attribute__((ext_vector_type(4))) float f32x4;
// OFD_WP16_ABL (diagnostic builds, wrong results): 1 no global input loads after a tile's first chunk, 2 the weight sets are never refilled
#ifndef OFD_WP16_ABL
#define OFD_WP16_ABL 0
#endif

template <bool PRO>
__global__ void __launch_bounds__(256, 2) conv3x3_wp16_kernel(const ConvParams P) {
    using C = Cfg<4, 1>;
    constexpr int NTHREADS = 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, ns = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;           // column of a 16-wide tile; k-group (octet) of an operand / row group of an accumulator

    const int tiles_y = (P.H + C::ROWS - 1) / C::ROWS;
    const int tpi = P.tiles_x * tiles_y, ntiles = tpi * P.B;
    int tile = blockIdx.x, cy = blockIdx.y;
    if (P.cy_fast) {                                  // block j -> XCD j % 8, channel block (j / 8) % NY, tile slot j / 8 / NY
        const int ny = P.Cout / C::BN, j = blockIdx.x, g = j >> 3;
        cy = g % ny;
        tile = (g / ny) * 8 + (j & 7);
        if (tile >= ntiles) return;
    }
    if (ntiles >= 8) {
        const int q = ntiles / 8, r = ntiles % 8, xcd = tile % 8, idx = tile / 8;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int b = tile / tpi, t_in = tile % tpi;
    const int oy0 = (t_in / P.tiles_x) * C::ROWS, ox0 = (t_in % P.tiles_x) * TW;
    const int cb = cy * C::BN + 32 * ns;              // this wave's 32 output channels

    // ---- weights: fragment (kx, ky, h) of 32-channel chunk kc = rows [tap][kc * 4 + lg][cb + 16 h + l15][8]
    const int cin8 = P.Cin_total / 8, n32 = P.total_chunks * 2;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)P.weight, 0, 9 * P.Cin_total * P.Cout * 2, 0x00020000);
    const int w_lane = (lg * P.Cout + cb + l15) * 16;
    const int w_row = P.Cout * 16;                    // bytes per [Cin/8] row
    auto load_w = [&](int kc, int kx, int i) -> u4 {  // i = ky * 2 + h
        const int ky = i >> 1, h = i & 1;
        const int row = (ky * 3 + kx) * cin8 + kc * NC;
        return __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, w_lane + h * 256, row * w_row, 0));
    };

    // ---- input staging (as conv3x3_wp_kernel)
    const int c8 = tid % NC;
    int pyx[C::XPT];
    unsigned okmask = 0;
#pragma unroll
    for (int i = 0; i < C::XPT; ++i) {
        const int p = min(tid / NC + i * (NTHREADS / NC), C::NPIX - 1);
        const int ty = p / IW, tx = p - ty * IW;
        const int iy = oy0 - 1 + ty, ix = ox0 - 1 + tx;
        const bool ok = iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
        okmask |= (ok ? 1u : 0u) << i;
        pyx[i] = (min(max(iy, 0), P.H - 1) << 16) | min(max(ix, 0), P.W - 1);
    }
    int src_i = 0, src_first = 0;
    auto load_x = [&](int kc, u4 (&xs)[C::XPT]) {
        const int k64 = kc >> 1;
        while (k64 >= src_first + P.src[src_i].chunks) {
            src_first += P.src[src_i].chunks;
            ++src_i;
        }
        const ConvSrcDev& S = P.src[src_i];
        const bf16_t* base = S.ptr + (size_t)b * S.SH * S.SW * S.src_channels + S.ch_offset + (k64 - src_first) * 64 + (kc & 1) * CK + c8 * 8;
        const int up = S.mode == 1 ? 1 : 0;
#pragma unroll
        for (int i = 0; i < C::XPT; ++i) {
            const int sy = (pyx[i] >> 16) >> up, sx = (pyx[i] & 0xffff) >> up;
            if (!(OFD_WP16_ABL & 1) || kc == 0) xs[i] = *(const u4*)(base + ((size_t)sy * S.SW + sx) * S.src_channels);
            else asm volatile("" : "+v"(xs[i]) : "v"(sy), "v"(sx));
        }
    };
    auto write_x = [&](int kc, const u4 (&xs)[C::XPT], unsigned char* xbuf) {
        float ps[8], pb[8];
        if constexpr (PRO) {
            const float* sp = P.in_scale + (size_t)b * P.Cin_total + kc * CK + c8 * 8;
            const float* bp = P.in_shift + (size_t)b * P.Cin_total + kc * CK + c8 * 8;
            *(float4*)&ps[0] = *(const float4*)sp; *(float4*)&ps[4] = *(const float4*)(sp + 4);
            *(float4*)&pb[0] = *(const float4*)bp; *(float4*)&pb[4] = *(const float4*)(bp + 4);
        }
#pragma unroll
        for (int i = 0; i < C::XPT; ++i) {
            const int p = min(tid / NC + i * (NTHREADS / NC), C::NPIX - 1);
            u4 v = xs[i];
            if constexpr (PRO) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float lo = silu_f(bf2f((bf16_t)(v[j] & 0xffffu)) * ps[2 * j] + pb[2 * j]);
                    const float hi = silu_f(bf2f((bf16_t)(v[j] >> 16)) * ps[2 * j + 1] + pb[2 * j + 1]);
                    v[j] = f2bf2(lo, hi);
                }
            }
            const bool ok = (okmask >> i) & 1u;       // zero padding is applied AFTER the prologue (DD:181-187 -> DD:114)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ok ? v[j] : 0u;
            *(u4*)(xbuf + c8 * C::US + p * 16) = v;
        }
    };

    f32x4 acc[8][2][2];                               // [row][pixel half][channel half]
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[r][q >> 1][q & 1][k] = 0.0f;

    u4 xs[C::XPT];
    load_x(0, xs);
    write_x(0, xs, smem);
    u4 wa[6], wb[6];                                  // weight sets: kx iterations alternate between them (3 per chunk: the parity flips every chunk)
#pragma unroll
    for (int i = 0; i < 6; ++i) wa[i] = load_w(0, 0, i);

    const int xrow_off = lg * C::US + l15 * 16;       // octet lg, pixel l15 of a 16-pixel group
    // one kernel column of one chunk: 96 MFMAs with the weights in `cur`; `nxt` receives the next column's
    auto column = [&](const unsigned char* xrow, const int kx, u4 (&cur)[6], u4 (&nxt)[6], const int nkc, const int nkx, const bool fetch) {
        if (fetch && !(OFD_WP16_ABL & 2)) {
#pragma unroll
            for (int i = 0; i < 6; ++i) nxt[i] = load_w(nkc, nkx, i);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {          // rows 4 hf .. 4 hf + 3 need staged rows 4 hf .. 4 hf + 5
                bf16x8 x[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) x[j] = *(const bf16x8*)(xrow + ((4 * hf + j) * IW + kx + 16 * p) * 16);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            acc[4 * hf + r][p][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(cur[ky * 2 + h]), x[r + ky], acc[4 * hf + r][p][h], 0, 0, 0);
            }
    };
    // chunk kc; EVEN: its first column uses set A (chunks alternate: three columns each)
    auto chunk = [&](const int kc, auto even_tag, auto last_tag) {
        constexpr bool EVEN = decltype(even_tag)::value, LAST = decltype(last_tag)::value;
        const int kn = LAST ? kc : kc + 1;
        if constexpr (!LAST) load_x(kn, xs);
        __syncthreads();                              // tile kc complete; every wave is done reading the other buffer
        const unsigned char* xrow = smem + (kc & 1) * C::XB + xrow_off;
        unsigned char* xnext = smem + ((kc + 1) & 1) * C::XB;
        if constexpr (EVEN) {
            column(xrow, 0, wa, wb, kc, 1, true);
            if constexpr (!LAST) write_x(kn, xs, xnext);
            column(xrow, 1, wb, wa, kc, 2, true);
            column(xrow, 2, wa, wb, kn, 0, !LAST);
        } else {
            column(xrow, 0, wb, wa, kc, 1, true);
            if constexpr (!LAST) write_x(kn, xs, xnext);
            column(xrow, 1, wa, wb, kc, 2, true);
            column(xrow, 2, wb, wa, kn, 0, !LAST);
        }
    };
    for (int kc = 0; kc < n32 - 2; kc += 2) {         // (n32 is even: 64-channel chunks of the descriptors)
        chunk(kc, std::true_type{}, std::false_type{});
        chunk(kc + 1, std::false_type{}, std::false_type{});
    }
    chunk(n32 - 2, std::true_type{}, std::false_type{});
    chunk(n32 - 1, std::false_type{}, std::true_type{});

    // ---- epilogue: bias, bf16, 16-byte stores, GroupNorm partial sums of the values as stored
    // lane (l15, lg) holds channels cb + 16 h + 4 lg + {0..3} of pixel 16 p + l15.  v_permlane16_swap(X = half 0, Y = half 1) leaves row lg
    // with channels 16 (lg & 1) + 8 (lg >> 1) + {0..7} of that pixel: one 16-byte store per lane covers the wave's 32 channels
    float4 bias4[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) bias4[h] = P.bias ? *(const float4*)(P.bias + cb + 16 * h + 4 * lg) : make_float4(0.f, 0.f, 0.f, 0.f);
    float st[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};     // [channel half][sum, sum of squares]: the 8-channel group 2 h + (lg >> 1)
    const bf16x2 one = __builtin_bit_cast(bf16x2, 0x3f803f80u);
    const int c_store = cb + 16 * (lg & 1) + 8 * (lg >> 1);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int oy = oy0 + r;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int ox = ox0 + 16 * p + l15;
            const bool ok = oy < P.H && ox < P.W && !(P.dbg & 16);
            const size_t pix = ((size_t)b * P.H + min(oy, P.H - 1)) * P.W + min(ox, P.W - 1);
            uint2 q[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 a = acc[r][p][h];
                q[h] = make_uint2(f2bf2(a[0] + bias4[h].x, a[1] + bias4[h].y), f2bf2(a[2] + bias4[h].z, a[3] + bias4[h].w));
                if (P.gn_partial && ok) {
                    const bf16x2 va = __builtin_bit_cast(bf16x2, q[h].x), vb = __builtin_bit_cast(bf16x2, q[h].y);
                    st[h][0] = __builtin_amdgcn_fdot2_f32_bf16(vb, one, __builtin_amdgcn_fdot2_f32_bf16(va, one, st[h][0], false), false);
                    st[h][1] = __builtin_amdgcn_fdot2_f32_bf16(vb, vb, __builtin_amdgcn_fdot2_f32_bf16(va, va, st[h][1], false), false);
                }
            }
            const auto rx = __builtin_amdgcn_permlane16_swap(q[0].x, q[1].x, false, false);
            const auto ry = __builtin_amdgcn_permlane16_swap(q[0].y, q[1].y, false, false);
            if (ok) *(uint4*)(P.out + pix * P.Cout + c_store) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
        }
    }
    if (P.gn_partial) {
        // sums over the 32 lanes that hold one 8-channel group (rows lg = 2 m, 2 m + 1 of 16 lanes): xor 1, 2, 4, 8, 16
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                float v = st[h][w];
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64);
                st[h][w] = v;
            }
        // slots of conv3x3_wp_kernel (4 per 8-row tile and tile column; gn_partial_index); this wave's octets cb/8 + o (o = 2 h + m, m = lg >> 1 of the holder):
        // lane t < 32 writes float t of the wave's share (slot_i, octet o, which); the sums go to slot ns, zeros elsewhere
        const int ty8 = oy0 / 8, tiles8 = (P.H + 7) / 8;
        if (ty8 < tiles8) {
            constexpr int OCT = C::BN / 8, PER_WAVE = OCT * 2;     // 16 octets of the workgroup's channel block; this wave writes slot ns of all of them
            const int o = (lane % PER_WAVE) >> 1, which = lane & 1;
            const int oo = o & 3, hh = oo >> 1, mm = oo & 1;        // own octet index -> (channel half, lane row pair)
            const float t00 = __shfl(st[0][0], mm * 32, 64), t01 = __shfl(st[0][1], mm * 32, 64);
            const float t10 = __shfl(st[1][0], mm * 32, 64), t11 = __shfl(st[1][1], mm * 32, 64);
            const float total = hh ? (which ? t11 : t10) : (which ? t01 : t00);
            if (lane < PER_WAVE) {
                const bool own = (o >> 2) == ns;
                P.gn_partial[gn_partial_index(b, tiles8 * P.tiles_x * 4, (ty8 * P.tiles_x + (t_in % P.tiles_x)) * 4 + ns, P.Cout / 8, cy * C::BN / 8 + o) + which] =
                    own ? total : 0.0f;
            }
        }
    }
}

template <bool PRO>
static int launch16(const ConvParams& P, hipStream_t s) {
    using C = Cfg<4, 1>;
    static bool attr_set = false;
    if (!attr_set) {
        OFD_HIP(hipFuncSetAttribute((const void*)conv3x3_wp16_kernel<PRO>, hipFuncAttributeMaxDynamicSharedMemorySize, wp_lds(C::LDS_BYTES)));
        attr_set = true;
    }
    const int tiles_y = (P.H + C::ROWS - 1) / C::ROWS;
    const int ntiles = P.tiles_x * tiles_y * P.B, ny = P.Cout / C::BN;
    dim3 grid(ntiles, ny);
    if (P.cy_fast) grid = dim3((ntiles + 7) / 8 * 8 * ny, 1);
    conv3x3_wp16_kernel<PRO><<<grid, C::NTHREADS, wp_lds(C::LDS_BYTES), s>>>(P);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

// ---- producer / consumer form of the 64 -> 64 3x3 (r04) -----------------------------------------------------------------------------
// What the in-kernel stamps of conv3x3_wp_kernel<2,2> say (tools/probe/wp_stamps.py, 64 -> 64 at 16 x 440 x 1024): a wave lives 41-48 k cycles
// per tile and spends a third of them in its MFMA phases; the rest is serial in the same wave -- tile decode 2.5 k, the first tile's
// global loads 5-8 k (the latency of a tile's loads under load: more than one chunk of MFMAs), its staging 0.6 k (9.5 k with the GroupNorm +
// SiLU prologue), barriers 3 k, epilogue 6-11 k -- and with two waves per SIMD, both in the same program, the matrix pipe idles whenever both
// are outside their MFMA phase: MFMA-busy 0.34-0.41.  The layer itself is close to HBM-bound: 1.9 GB at the ~5 TB/s a 1 : 1 read / write
// mix streams = 0.4 ms against 0.3 ms of MFMAs, so nothing may be serial with the memory stream.
// Here the halves of the work run in DIFFERENT waves of one persistent 512-thread workgroup (one per CU):
//   * waves 0-3, the consumers (one per SIMD): the MFMA chunk body of conv3x3_wp_kernel and its epilogue, nothing else -- no global
//     loads at all inside the MFMA stream: the 9 x 64 x 64 weights (73.7 KB) are staged ONCE per workgroup into LDS, fragment-major (a
//     fragment = one conflict-free ds_read_b128 per lane), the bias lives in registers, the accumulators start at the bias.  (A first
//     version read the weights from L2 as conv3x3_wp_kernel does: in-order return put every weight fragment behind the epilogue's stores
//     and the chunks ran at 50-85 cycles per MFMA; same-box ablation without the refills: 0.68 -> 0.58 ms);
//   * waves 4-7, the producers (the other wave of each SIMD): fetch the input tile of chunk i + 3 into registers (three register sets: a
//     tile's loads take 5-8 k cycles under load), apply the prologue to chunk i + 1 and write it to the other LDS buffer while the
//     consumers multiply chunk i.  Their VALU stream fills the 24 issue cycles an MFMA leaves free on the SIMD;
//   * ONE workgroup barrier per chunk (144 MFMAs per consumer wave), passed by the consumers as soon as their LDS reads of the chunk
//     are issued: the epilogue of a tile runs behind the barrier, beside the producers' staging of the next tile;
//   * the chunk stream runs across tiles (a persistent grid of one workgroup per CU walks the pixel tiles in the XCD-aware order of
//     conv3x3_wp_kernel): tile decode, first-tile latency and pipeline fill are paid once per launch, not per tile.
// Serves Cin = Cout = 64 from one same-size source with the plain / prologue / GroupNorm-statistics epilogues (inference and the training
// forward: 12 of the 43 3x3 launches of a denoise step); everything else stays on conv3x3_wp_kernel.  OFD_CONV_PC=0 switches it off.
//
// OFD_PC_ABL (diagnostic builds, wrong results): 2 the producers fetch nothing (they stage whatever their registers hold), 8 no MFMAs,
// 16 no epilogue stores (statistics kept: the accumulators stay live)
#ifndef OFD_PC_ABL
#define OFD_PC_ABL 0
#endif
#ifndef OFD_PC_PRIO
#define OFD_PC_PRIO 2
#endif
#if OFD_WP_STAMPS
// diagnostic build: the producers stamp step PC_STAMP_STEP of their walk (1 step start, 2 loads issued, 3 chunk staged, 4 barrier passed), the
// consumers the item that step belongs to (5 item start, 6 / 8 chunk issued, 7 / 9 barrier passed, 10 epilogue issued)
#define PC_STAMP_STEP 24
#define PC_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); stamps[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PC_STAMP_P(k) do { if (i == PC_STAMP_STEP) PC_STAMP(k); } while (0)
#define PC_STAMP_C(k) do { if (item_no == PC_STAMP_STEP / 2) PC_STAMP(k); } while (0)
#define PC_STAMP_FLUSH() do { stamps[15] = __builtin_amdgcn_s_memrealtime(); const unsigned wv = blockIdx.x * 8 + (threadIdx.x >> 6);            \
        if ((threadIdx.x & 63) == 0 && wv < (unsigned)STAMP_WAVES) for (int q_ = 0; q_ < STAMP_SLOTS; ++q_) g_wp_stamps[(size_t)wv * STAMP_SLOTS + q_] = stamps[q_]; } while (0)
#else
#define PC_STAMP(i) do { } while (0)
#define PC_STAMP_P(k) do { } while (0)
#define PC_STAMP_C(k) do { } while (0)
#define PC_STAMP_FLUSH() do { } while (0)
#endif

struct PcCfg {
    using C = Cfg<2, 2>;                               // consumers: 2 channel slices x 2 row blocks = a 16 x 32 pixel tile x 64 channels
    static constexpr int NPROD = 256;
    static constexpr int XPT = (C::NPIX * NC + NPROD - 1) / NPROD;
    static constexpr int WSLOT = FRAGS * 2 * 1024;     // weights of one 32-channel chunk: [fragment][slice][lane][16 B] = 36.9 KB
    static constexpr int MAX_ITEMS = 512;              // item descriptors of a workgroup, decoded once (16 bytes each)
    static constexpr int ITEMS_OFF = 2 * C::XB + 2 * WSLOT;
    static constexpr int LDS_BYTES = ITEMS_OFF + MAX_ITEMS * 16;
};

template <bool PRO>
__global__ void __launch_bounds__(512, 2) conv3x3_pc_kernel(const ConvParams P) {
    using C = PcCfg::C;
    constexpr int NPROD = PcCfg::NPROD, XPT = PcCfg::XPT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const wlds = smem + 2 * C::XB;
    const int tid = threadIdx.x;
#if OFD_WP_STAMPS
    unsigned long long stamps[STAMP_SLOTS] = {};
    stamps[14] = __builtin_amdgcn_s_memrealtime();
    stamps[12] = __builtin_amdgcn_s_memtime();
#endif

    const int tiles_y = (P.H + C::ROWS - 1) / C::ROWS;
    const int tpi = P.tiles_x * tiles_y, ntiles = tpi * P.B, ny = P.Cout / C::BN, G = gridDim.x;
    const int nitems = (ntiles + 7) / 8 * 8 * ny;     // item j -> XCD j % 8, channel block (j / 8) % ny, tile slot j / 8 / ny (as conv3x3_wp_kernel, cy_fast)
    const int n32 = P.total_chunks * 2;
    // item j -> (sample, tile origin, channel block), XCD-aware as conv3x3_wp_kernel: blocks that share an XCD (j % 8) walk a contiguous run
    // of tiles (the halo rows of neighbours hit one L2), the channel blocks of a tile back to back.  G is a multiple of 8: once an item of
    // this workgroup is past the end, every later one is too.
    auto decode = [&](int j, int& b, int& oy0, int& ox0, int& cy) -> bool {
        if (j >= nitems) return false;
        const int g = j >> 3;
        cy = ny > 1 ? g % ny : 0;
        int tile = (ny > 1 ? g / ny : g) * 8 + (j & 7);
        if (tile >= ntiles) return false;
        if (ntiles >= 8) {
            const int q = ntiles / 8, r = ntiles % 8, xcd = tile % 8, idx = tile / 8;
            tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
        b = tile / tpi;
        const int t_in = tile - b * tpi;
        oy0 = (t_in / P.tiles_x) * C::ROWS;
        ox0 = (t_in % P.tiles_x) * TW;
        return true;
    };
    // the items of this workgroup (j = blockIdx.x + k G), decoded ONCE into LDS: the walk reads a descriptor instead of dividing
    const int nit = blockIdx.x < nitems ? min((nitems - 1 - (int)blockIdx.x) / G + 1, PcCfg::MAX_ITEMS) : 0;
    int4* const items = (int4*)(smem + PcCfg::ITEMS_OFF);
    for (int k = tid; k < nit; k += 512) {
        int b_, y_, x_, c_;
        const bool ok = decode(blockIdx.x + k * G, b_, y_, x_, c_);
        items[k] = make_int4(ok ? b_ : -1, y_, x_, c_);
    }
    __syncthreads();
    int nvalid_items = 0;
    if (nit > 0) {                                     // (valid items come first: see decode)
        int lo = 0, hi = nit;                          // first invalid index
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (items[mid].x >= 0) lo = mid + 1; else hi = mid; }
        nvalid_items = lo;
    }
    const int T = n32 * nvalid_items;                  // chunks it walks: both roles execute 1 + T barriers
    if (T == 0) return;
    auto item_at = [&](int k, int& b, int& oy0, int& ox0, int& cy) -> bool {
        if (k >= nvalid_items) return false;
        const int4 d = items[k];                       // (the same for every lane: scalar registers from here on)
        b = __builtin_amdgcn_readfirstlane(d.x); oy0 = __builtin_amdgcn_readfirstlane(d.y);
        ox0 = __builtin_amdgcn_readfirstlane(d.z); cy = __builtin_amdgcn_readfirstlane(d.w);
        return true;
    };

    if (tid >= 256) {
        // =========================================================== producers
        const int ptid = tid - 256;
        const int c8 = ptid % NC;
        int tyx[XPT];                                  // tile-relative (row << 8 | column) of this thread's units, halo included
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int p = min(ptid / NC + i * (NPROD / NC), C::NPIX - 1);
            const int ty = p / IW;
            tyx[i] = (ty << 8) | (p - ty * IW);
        }
        // load cursor: the chunk whose global loads are issued next.  Past the workgroup's last chunk it stays there (the loads are repeated
        // into registers nobody stages for a consumer): no step of the walk is conditional, so a step is one basic block
        int lk = 0, lkc = 0, lb, loy0, lox0, lcy;
        item_at(0, lb, loy0, lox0, lcy);
        auto advance = [&]() {
            if (++lkc == n32) {
                if (item_at(lk + 1, lb, loy0, lox0, lcy)) { lkc = 0; ++lk; }
                else lkc = n32 - 1;
            }
        };
        auto issue = [&](u4 (&xs)[XPT], unsigned& okmask, float (&ps)[8], float (&pb)[8]) {
            const int k64 = lkc >> 1;
            int si = 0, first = 0;                     // the source that owns 64-channel chunk k64 (concatenated inputs, DD:405)
            while (k64 >= first + P.src[si].chunks) { first += P.src[si].chunks; ++si; }
            const ConvSrcDev& S = P.src[si];
            const bf16_t* base = S.ptr + (size_t)lb * S.SH * S.SW * S.src_channels + S.ch_offset + (k64 - first) * 64 + (lkc & 1) * CK + c8 * 8;
            const int up = S.mode == 1 ? 1 : 0;       // nearest x2 up-sampling of the source (DD:91) is a shift of the coordinates
            okmask = 0;
#pragma unroll
            for (int i = 0; i < XPT; ++i) {
                const int iy = loy0 - 1 + (tyx[i] >> 8), ix = lox0 - 1 + (tyx[i] & 0xff);
                const bool ok = iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
                okmask |= (ok ? 1u : 0u) << i;
                const int sy = min(max(iy, 0), P.H - 1) >> up, sx = min(max(ix, 0), P.W - 1) >> up;
                if (!(OFD_PC_ABL & 2)) xs[i] = *(const u4*)(base + ((size_t)sy * S.SW + sx) * S.src_channels);
                else asm volatile("" : "+v"(xs[i]) : "v"(sy), "v"(sx));
            }
            if constexpr (PRO) {
                const float* sp = P.in_scale + (size_t)lb * P.Cin_total + lkc * CK + c8 * 8;
                const float* bp = P.in_shift + (size_t)lb * P.Cin_total + lkc * CK + c8 * 8;
                *(float4*)&ps[0] = *(const float4*)sp; *(float4*)&ps[4] = *(const float4*)(sp + 4);
                *(float4*)&pb[0] = *(const float4*)bp; *(float4*)&pb[4] = *(const float4*)(bp + 4);
            }
        };
        auto stage = [&](const u4 (&xs)[XPT], const unsigned okmask, const float (&ps)[8], const float (&pb)[8], unsigned char* xbuf) {
#pragma unroll
            for (int i = 0; i < XPT; ++i) {
                const int p = min(ptid / NC + i * (NPROD / NC), C::NPIX - 1);
                u4 v = xs[i];
                if constexpr (PRO) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float lo = silu_f(bf2f((bf16_t)(v[j] & 0xffffu)) * ps[2 * j] + pb[2 * j]);
                        const float hi = silu_f(bf2f((bf16_t)(v[j] >> 16)) * ps[2 * j + 1] + pb[2 * j + 1]);
                        v[j] = f2bf2(lo, hi);
                    }
                }
                const bool ok = (okmask >> i) & 1u;   // zero padding is applied AFTER the prologue (DD:181-187 -> DD:114)
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = ok ? v[j] : 0u;
                *(u4*)(xbuf + c8 * C::US + p * 16) = v;
            }
        };
        // ---- weights: chunk c of the walk lives in LDS slot c % 2, fragment-major [fragment (ks, kx, ky)][slice][lane][16 B] (a consumer's
        //      fragment = one conflict-free ds_read_b128 per lane).  The producers fetch the NEXT chunk's 36 fragments by LDS-DMA, nine per
        //      wave, at the top of a step, and wait for them (counted: this step's input loads stay in flight) before its barrier.  Issued from
        //      inline asm: a DMA the compiler can see makes it drain vmcnt in front of every LDS access.  An LDS-DMA costs its wave 60-180 issue
        //      cycles: the consumers issued their own in a first version and lost a fifth of every chunk to it.  A slot that already holds the
        //      chunk is left alone (the 64 -> 64 layers: both chunks resident for the whole launch).
        const int pw = __builtin_amdgcn_readfirstlane(ptid >> 6), plane_ = ptid & 63;
        const int cin8 = P.Cin_total / 8;
        const unsigned wlds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)wlds;
        int held0 = -1, held1 = -1;                    // (cy << 16 | kc) each slot holds
        int sk = 0, skc = 0, scy, sb_, sy_, sx_;       // stage cursor: the chunk staged next (its weights are fetched with it)
        item_at(0, sb_, sy_, sx_, scy);
        auto stage_advance = [&]() {
            if (++skc == n32) {
                if (item_at(sk + 1, sb_, sy_, sx_, scy)) { skc = 0; ++sk; }
                else skc = n32 - 1;
            }
        };
        auto weights_dma = [&](const int slot) -> bool {      // the stage cursor's chunk -> slot; false: the slot holds it already
            const int key = (scy << 16) | skc;
            int& held = slot ? held1 : held0;
            if (held == key) return false;
            held = key;
#pragma unroll
            for (int q = 0; q < FRAGS * 2 / 4; ++q) {
                const int fr = pw * (FRAGS * 2 / 4) + q, fi = fr >> 1, ns_ = fr & 1;      // (fragment, slice)
                const int g = fi / 3, ky = fi - g * 3, ks = g / 3, kx = g - ks * 3;
                const int row = (ky * 3 + kx) * cin8 + skc * NC + ks * 2;
                const bf16_t* sbase = P.weight + ((size_t)row * P.Cout + scy * C::BN + 32 * ns_) * 8;                 // (uniform)
                const unsigned dst = wlds_addr + (unsigned)(((slot * FRAGS + fi) * 2 + ns_) * 1024);                  // (uniform)
                const unsigned voff = (unsigned)(((plane_ >> 5) * P.Cout + (plane_ & 31)) * 16);                     // row + half, column l31
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(voff), "s"(sbase) : "memory");
            }
            return true;
        };
        // three register sets: chunk c lives in set c % 3 from its fetch (three steps before the consumers need it) to its staging
        u4 x0[XPT], x1[XPT], x2[XPT];
        unsigned ok0 = 0, ok1 = 0, ok2 = 0;
        float ps0[8], pb0[8], ps1[8], pb1[8], ps2[8], pb2[8];
        weights_dma(0); stage_advance();                                     // chunk 0's weights
        issue(x0, ok0, ps0, pb0); advance();                                 // chunk 0
        issue(x1, ok1, ps1, pb1); advance();                                 // chunk 1
        issue(x2, ok2, ps2, pb2); advance();                                 // chunk 2
        stage(x0, ok0, ps0, pb0, smem);
        PC_STAMP(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                     // barrier 0: chunk 0 and its weights are staged
        // step i (the consumers multiply chunk i from buffer i % 2): fetch chunk i + 3 -> set i % 3, stage chunk i + 1 (set (i + 1) % 3) ->
        // buffer (i + 1) % 2.  Period 6.  (Past the end of the walk a step stages stale registers into the buffer nobody reads.)
#define PC_STEP(XL, OKL, PSL, PBL, XS, OKS, PSS, PBS, BUF)                                          \
        {                                                                                           \
            if (i >= T) break;                                                                      \
            PC_STAMP_P(1);                                                                          \
            const bool wd_ = weights_dma(BUF);          /* chunk i + 1 -> slot (i + 1) % 2 */        \
            stage_advance();                                                                        \
            issue(XL, OKL, PSL, PBL);                                                               \
            PC_STAMP_P(2);                                                                          \
            stage(XS, OKS, PSS, PBS, smem + (BUF) * C::XB);                                         \
            advance();                                                                              \
            PC_STAMP_P(3);                                                                          \
            if (wd_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XPT + (PRO ? 4 : 0)) : "memory");     \
            __syncthreads();                                                                        \
            PC_STAMP_P(4);                                                                          \
            ++i;                                                                                    \
        }
        for (int i = 0; i < T;) {
            PC_STEP(x0, ok0, ps0, pb0, x1, ok1, ps1, pb1, 1)
            PC_STEP(x1, ok1, ps1, pb1, x2, ok2, ps2, pb2, 0)
            PC_STEP(x2, ok2, ps2, pb2, x0, ok0, ps0, pb0, 1)
            PC_STEP(x0, ok0, ps0, pb0, x1, ok1, ps1, pb1, 0)
            PC_STEP(x1, ok1, ps1, pb1, x2, ok2, ps2, pb2, 1)
            PC_STEP(x2, ok2, ps2, pb2, x0, ok0, ps0, pb0, 0)
        }
#undef PC_STEP
        PC_STAMP_FLUSH();
        return;
    }

    // =============================================================== consumers
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int ns = wave & 1, ph = wave >> 1;
    if (OFD_PC_PRIO) __builtin_amdgcn_s_setprio(OFD_PC_PRIO);      // the MFMA stream goes first; the producer wave of this SIMD fills its gaps

    int ik = 0, b, oy0, ox0, cy;
    item_at(0, b, oy0, ox0, cy);                       // (T > 0: the first item is valid)

    f32x16 biasv;                                     // register 4 g + k of an accumulator row = channel cb + 8 g + 4 half + k
    auto load_bias = [&](int cy_) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b4 = P.bias ? *(const float4*)(P.bias + cy_ * C::BN + 32 * ns + 8 * g + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
            biasv[4 * g] = b4.x; biasv[4 * g + 1] = b4.y; biasv[4 * g + 2] = b4.z; biasv[4 * g + 3] = b4.w;
        }
    };
    load_bias(cy);
    f32x16 acc[8];

    const int xrow_off = half * C::US + (8 * ph * IW + l31) * 16;
    const unsigned char* const wfrag = wlds + ns * 1024 + lane * 16;
    // one 32-channel chunk: 144 MFMAs, operands from LDS only
    // (FIRST: the first chunk of an item -- its first MFMA per accumulator row takes the bias as its C operand: no accumulator initialisation)
    auto chunk = [&](const int slot, const unsigned char* xbase, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const unsigned char* xrow = xbase + xrow_off;
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            const int ks = g / 3, kx = g % 3;
            bf16x8 x[10];
#pragma unroll
            for (int jr = 0; jr < 10; ++jr) x[jr] = *(const bf16x8*)(xrow + (jr * IW + kx) * 16 + ks * 2 * C::US);
            bf16x8 a[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) a[ky] = *(const bf16x8*)(wfrag + (slot * FRAGS + g * 3 + ky) * 2048);
            if (OFD_PC_ABL & 8) {
#pragma unroll
                for (int jr = 0; jr < 10; ++jr) asm volatile("" :: "v"(x[jr]));
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) asm volatile("" :: "v"(a[ky]));
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
                        acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ky], x[r + ky], (FIRST && g == 0 && ky == 0) ? biasv : acc[r], 0, 0, 0);
            }
        }
    };

    const int tiles8 = (P.H + 7) / 8;
    __syncthreads();                                   // barrier 0: chunk 0 and its weights are staged
    PC_STAMP(11);
    int item_no = 0;
    (void)item_no;
    while (true) {
        PC_STAMP_C(5);
        int nb, noy0, nox0, ncy;
        const bool nvalid = item_at(ik + 1, nb, noy0, nox0, ncy);
        for (int kc = 0; kc < n32; kc += 2) {
            // (n32 is even: an item starts on slot / buffer 0)
            if (kc == 0) chunk(0, smem, std::true_type{}); else chunk(0, smem, std::false_type{});
            PC_STAMP_C(6);
            __syncthreads();
            PC_STAMP_C(7);
            chunk(1, smem + C::XB, std::false_type{});
            PC_STAMP_C(8);
            __syncthreads();
            PC_STAMP_C(9);
        }
        // ---- epilogue of the tile (behind the barrier: the producers are already staging the next tile): bf16 16-byte stores (one
        //      v_permlane32_swap per dword pairs two register quads), GroupNorm partial sums of the values as stored
        if (OFD_PC_PRIO) __builtin_amdgcn_s_setprio(0);
        float stat[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) stat[i] = 0.0f;
        const int oyb = oy0 + 8 * ph, ox = ox0 + l31, cb = cy * C::BN + 32 * ns;
        const bool okx = ox < P.W && !(P.dbg & 16);
        // stores through a buffer descriptor of the sample's plane: scalar base and row offsets, one 32-bit lane offset (an offset past the
        // end is dropped by the hardware, so a tile at the right / bottom edge needs no branch around its stores)
        const size_t oplane_b = (size_t)P.H * P.W * P.Cout * 2;
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(P.out + (size_t)b * P.H * P.W * P.Cout), 0, (int)oplane_b, 0x00020000);
        const unsigned olane = (unsigned)(((min(oyb, P.H - 1) * P.W + min(ox, P.W - 1)) * P.Cout + cb + 8 * half) * 2);
        const unsigned ostride_b = (unsigned)(P.W * P.Cout * 2);
        const bool full = ox0 + TW <= P.W && oyb + 8 <= P.H && !(P.dbg & 16) && oplane_b < (1ull << 31);      // (uniform) every pixel of this wave's block is inside
        if (full) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                uint2 q[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    q[g] = make_uint2(f2bf2(acc[r][4 * g], acc[r][4 * g + 1]), f2bf2(acc[r][4 * g + 2], acc[r][4 * g + 3]));
                    if (P.gn_partial) {
                        const bf16x2 one = __builtin_bit_cast(bf16x2, 0x3f803f80u);
                        const bf16x2 va = __builtin_bit_cast(bf16x2, q[g].x), vb = __builtin_bit_cast(bf16x2, q[g].y);
                        stat[g * 2] = __builtin_amdgcn_fdot2_f32_bf16(vb, one, __builtin_amdgcn_fdot2_f32_bf16(va, one, stat[g * 2], false), false);
                        stat[g * 2 + 1] = __builtin_amdgcn_fdot2_f32_bf16(vb, vb, __builtin_amdgcn_fdot2_f32_bf16(va, va, stat[g * 2 + 1], false), false);
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    const auto rx = __builtin_amdgcn_permlane32_swap(q[g].x, q[g + 1].x, false, false);
                    const auto ry = __builtin_amdgcn_permlane32_swap(q[g].y, q[g + 1].y, false, false);
                    u4 pk = {rx[0], ry[0], rx[1], ry[1]};
                    if (!(OFD_PC_ABL & 16)) __builtin_amdgcn_raw_buffer_store_b128(pk, orsrc, (int)olane, (int)(r * ostride_b + 16 * g), 0);
                    else asm volatile("" :: "v"(pk));
                }
            }
        } else {
        bf16_t* orow = P.out + (((size_t)b * P.H + min(oyb, P.H - 1)) * P.W + min(ox, P.W - 1)) * P.Cout + cb + 8 * half;
        const size_t ostride = (size_t)P.W * P.Cout;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const bool ok = okx && oyb + r < P.H;
            uint2 q[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                q[g] = make_uint2(f2bf2(acc[r][4 * g], acc[r][4 * g + 1]), f2bf2(acc[r][4 * g + 2], acc[r][4 * g + 3]));
                if (P.gn_partial && ok) {
                    const bf16x2 one = __builtin_bit_cast(bf16x2, 0x3f803f80u);
                    const bf16x2 va = __builtin_bit_cast(bf16x2, q[g].x), vb = __builtin_bit_cast(bf16x2, q[g].y);
                    stat[g * 2] = __builtin_amdgcn_fdot2_f32_bf16(vb, one, __builtin_amdgcn_fdot2_f32_bf16(va, one, stat[g * 2], false), false);
                    stat[g * 2 + 1] = __builtin_amdgcn_fdot2_f32_bf16(vb, vb, __builtin_amdgcn_fdot2_f32_bf16(va, va, stat[g * 2 + 1], false), false);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const auto rx = __builtin_amdgcn_permlane32_swap(q[g].x, q[g + 1].x, false, false);
                const auto ry = __builtin_amdgcn_permlane32_swap(q[g].y, q[g + 1].y, false, false);
                if (ok && !(OFD_PC_ABL & 16)) *(uint4*)(orow + 8 * g) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
                else if (OFD_PC_ABL & 16) asm volatile("" :: "v"(rx[0]), "v"(ry[0]), "v"(rx[1]), "v"(ry[1]));
            }
            orow += ostride;
        }
        }
        if (P.gn_partial) {
            // slots of conv3x3_wp_kernel (4 per 8-row tile and tile column; gn_partial_index); this wave's sums go to slot ns, every other
            // (slot, octet) of the workgroup's channel block is written as zero by the wave whose slot it is (slots ns, ns + 2)
            wave_reduce8(stat);
            const int ty8 = oy0 / 8 + ph;
            if (ty8 < tiles8) {
                constexpr int OCT = 8, PER_WAVE = 2 * OCT * 2;
                const int slot_i = lane / (OCT * 2), o = (lane % (OCT * 2)) >> 1, which = lane & 1;
                const float total = __shfl(stat[0], ((o & 3) * 2 + which) * 8, 64);      // value index k lives in lanes 8k .. 8k+7
                if (lane < PER_WAVE) {
                    const int slot = ns + slot_i * 2;
                    const bool own = slot_i == 0 && (o >> 2) == ns;
                    P.gn_partial[gn_partial_index(b, tiles8 * P.tiles_x * 4, (ty8 * P.tiles_x + ox0 / TW) * 4 + slot, P.Cout / 8, cy * C::BN / 8 + o) + which] =
                        own ? total : 0.0f;
                }
            }
        }
        PC_STAMP_C(10);
#if OFD_WP_STAMPS
        if (tid == 0 && item_no < 64) g_wp_stamps[65536 + blockIdx.x * 64 + item_no] = __builtin_amdgcn_s_memtime();      // item end times of wave 0
#endif
        ++item_no;
        if (!nvalid) break;
        ++ik; b = nb; oy0 = noy0; ox0 = nox0;
        if (ncy != cy) { cy = ncy; load_bias(cy); }
        if (OFD_PC_PRIO) __builtin_amdgcn_s_setprio(OFD_PC_PRIO);
    }
    PC_STAMP_FLUSH();
}

// 64-channel output blocks, same-size or nearest-x2 sources, plain / prologue / statistics epilogue: the shapes conv3x3_pc_kernel serves
static bool pc_serves(const ConvParams& P) {
    for (int i = 0; i < P.n_src; ++i)
        if (P.src[i].mode != 0 && P.src[i].mode != 1) return false;
    return P.Cout % 64 == 0 && !P.residual && !P.residual_b && !P.res_act && !P.split && !P.pool2 && !P.out2 &&
           P.W <= 8160 /* tile-relative columns are packed into 8 bits + origin */;
}

template <bool PRO>
static int launch_pc(const ConvParams& P, hipStream_t s) {
    using C = PcCfg::C;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        OFD_HIP(hipGetDevice(&dev));
        OFD_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        OFD_HIP(hipFuncSetAttribute((const void*)conv3x3_pc_kernel<PRO>, hipFuncAttributeMaxDynamicSharedMemorySize, PcCfg::LDS_BYTES));
        static const int g_env = getenv("OFD_CONV_PC_GRID") ? atoi(getenv("OFD_CONV_PC_GRID")) : 0;
        if (g_env > 0) cus = g_env;
        cus = cus / 8 * 8;
        if (cus < 8) cus = 8;
    }
    const int tiles_y = (P.H + C::ROWS - 1) / C::ROWS;
    const int ntiles = P.tiles_x * tiles_y * P.B, ny = P.Cout / C::BN;
    const int nitems = (ntiles + 7) / 8 * 8 * ny;
    const int grid = nitems < cus ? nitems : cus;      // one 512-thread workgroup per CU; a multiple of 8 (the kernel's item order relies on it)
    if ((long)grid * PcCfg::MAX_ITEMS < nitems) return 1;      // more items per workgroup than its descriptor table holds: not served
    conv3x3_pc_kernel<PRO><<<grid, 512, PcCfg::LDS_BYTES, s>>>(P);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

// ---- producer / consumer form of the 128-channel-block 3x3 (r04) -----------------------------------------------------------------------
// conv3x3_wp16_kernel with its global loads taken out of the MFMA stream (same-box ablations of that kernel, profiles/r04_wp16_ablations.txt:
// without the weight refills -11 %, without the input loads -2..6 %, without both -14..18 %: a wave's weight fragments come from L2 but return
// in order behind the HBM loads of its own input tile).  One persistent 512-thread workgroup per CU:
//   * waves 0-3, the consumers: wave = 32 output channels x 8 rows x 32 pixels on MFMA 16x16x32, the column body of conv3x3_wp16_kernel, both
//     operands from LDS.  Weights: a PRIVATE ring of four kernel-column slots per wave (6 fragments = 6 KB each), filled by LDS-DMA the wave
//     itself issues two columns (192 MFMAs) ahead and waits for with a counted vmcnt -- no barrier, no registers, and the only other VMEM
//     operations of the wave are the stores of a tile's epilogue;
//   * waves 4-7, the producers: input tile of chunk i + 3 into registers, prologue + LDS write of chunk i + 1 (as conv3x3_pc_kernel);
//   * one workgroup barrier per 32-channel chunk (288 MFMAs per consumer wave); items = (8 x 32 pixel tile, 128-channel block), the channel
//     blocks of a tile back to back, XCD-aware order; item descriptors decoded once into LDS.
// Plain / prologue / statistics epilogues, same-size or nearest-x2 sources.  Parity-green and SLOWER than conv3x3_wp16_kernel (see launch_conv3x3_wp):
// opt-in, OFD_CONV_PCW=1.
struct PcwCfg {
    using C = Cfg<4, 1>;
    static constexpr int NPROD = 256;
    static constexpr int XPT = (C::NPIX * NC + NPROD - 1) / NPROD;
    static constexpr int WCOL = 6 * 1024, WSLOTS = 4, WWAVE = WSLOTS * WCOL;      // a wave's weight ring: 4 columns x 6 fragments x 1 KB
    static constexpr int W_OFF = 2 * C::XB, ITEMS_OFF = W_OFF + 4 * WWAVE, MAX_ITEMS = 512;
    static constexpr int LDS_BYTES = ITEMS_OFF + MAX_ITEMS * 16;
};

template <bool PRO>
__global__ void __launch_bounds__(512, 2) conv3x3_pcw_kernel(const ConvParams P) {
    using C = PcwCfg::C;
    constexpr int NPROD = PcwCfg::NPROD, XPT = PcwCfg::XPT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;

    const int tiles_y = (P.H + C::ROWS - 1) / C::ROWS;
    const int tpi = P.tiles_x * tiles_y, ntiles = tpi * P.B, ny = P.Cout / C::BN, G = gridDim.x;
    const int nitems = (ntiles + 7) / 8 * 8 * ny;
    const int n32 = P.total_chunks * 2;
    auto decode = [&](int j, int& b, int& oy0, int& ox0, int& cy) -> bool {
        if (j >= nitems) return false;
        const int g = j >> 3;
        cy = ny > 1 ? g % ny : 0;
        int tile = (ny > 1 ? g / ny : g) * 8 + (j & 7);
        if (tile >= ntiles) return false;
        if (ntiles >= 8) {
            const int q = ntiles / 8, r = ntiles % 8, xcd = tile % 8, idx = tile / 8;
            tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
        b = tile / tpi;
        const int t_in = tile - b * tpi;
        oy0 = (t_in / P.tiles_x) * C::ROWS;
        ox0 = (t_in % P.tiles_x) * TW;
        return true;
    };
    const int nit = blockIdx.x < nitems ? min((nitems - 1 - (int)blockIdx.x) / G + 1, PcwCfg::MAX_ITEMS) : 0;
    int4* const items = (int4*)(smem + PcwCfg::ITEMS_OFF);
    for (int k = tid; k < nit; k += 512) {
        int b_, y_, x_, c_;
        const bool ok = decode(blockIdx.x + k * G, b_, y_, x_, c_);
        items[k] = make_int4(ok ? b_ : -1, y_, x_, c_);
    }
    __syncthreads();
    int nvalid_items = 0;
    if (nit > 0) {
        int lo = 0, hi = nit;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (items[mid].x >= 0) lo = mid + 1; else hi = mid; }
        nvalid_items = lo;
    }
    const int T = n32 * nvalid_items;                  // chunks this workgroup walks: both roles execute 1 + T barriers
    if (T == 0) return;
    auto item_at = [&](int k, int& b, int& oy0, int& ox0, int& cy) -> bool {
        if (k >= nvalid_items) return false;
        const int4 d = items[k];                       // (the same for every lane: scalar registers from here on)
        b = __builtin_amdgcn_readfirstlane(d.x); oy0 = __builtin_amdgcn_readfirstlane(d.y);
        ox0 = __builtin_amdgcn_readfirstlane(d.z); cy = __builtin_amdgcn_readfirstlane(d.w);
        return true;
    };

    if (tid >= 256) {
        // =========================================================== producers (as conv3x3_pc_kernel, 10 x 34 pixel tiles)
        const int ptid = tid - 256;
        const int c8 = ptid % NC;
        int tyx[XPT];
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int p = min(ptid / NC + i * (NPROD / NC), C::NPIX - 1);
            const int ty = p / IW;
            tyx[i] = (ty << 8) | (p - ty * IW);
        }
        int lk = 0, lkc = 0, lb, loy0, lox0, lcy;
        item_at(0, lb, loy0, lox0, lcy);
        auto advance = [&]() {
            if (++lkc == n32) {
                if (item_at(lk + 1, lb, loy0, lox0, lcy)) { lkc = 0; ++lk; }
                else lkc = n32 - 1;
            }
        };
        auto issue = [&](u4 (&xs)[XPT], unsigned& okmask, float (&ps)[8], float (&pb)[8]) {
            const int k64 = lkc >> 1;
            int si = 0, first = 0;
            while (k64 >= first + P.src[si].chunks) { first += P.src[si].chunks; ++si; }
            const ConvSrcDev& S = P.src[si];
            const bf16_t* base = S.ptr + (size_t)lb * S.SH * S.SW * S.src_channels + S.ch_offset + (k64 - first) * 64 + (lkc & 1) * CK + c8 * 8;
            const int up = S.mode == 1 ? 1 : 0;
            okmask = 0;
#pragma unroll
            for (int i = 0; i < XPT; ++i) {
                const int iy = loy0 - 1 + (tyx[i] >> 8), ix = lox0 - 1 + (tyx[i] & 0xff);
                const bool ok = iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
                okmask |= (ok ? 1u : 0u) << i;
                const int sy = min(max(iy, 0), P.H - 1) >> up, sx = min(max(ix, 0), P.W - 1) >> up;
                xs[i] = *(const u4*)(base + ((size_t)sy * S.SW + sx) * S.src_channels);
            }
            if constexpr (PRO) {
                const float* sp = P.in_scale + (size_t)lb * P.Cin_total + lkc * CK + c8 * 8;
                const float* bp = P.in_shift + (size_t)lb * P.Cin_total + lkc * CK + c8 * 8;
                *(float4*)&ps[0] = *(const float4*)sp; *(float4*)&ps[4] = *(const float4*)(sp + 4);
                *(float4*)&pb[0] = *(const float4*)bp; *(float4*)&pb[4] = *(const float4*)(bp + 4);
            }
        };
        auto stage = [&](const u4 (&xs)[XPT], const unsigned okmask, const float (&ps)[8], const float (&pb)[8], unsigned char* xbuf) {
#pragma unroll
            for (int i = 0; i < XPT; ++i) {
                const int p = min(ptid / NC + i * (NPROD / NC), C::NPIX - 1);
                u4 v = xs[i];
                if constexpr (PRO) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float lo = silu_f(bf2f((bf16_t)(v[j] & 0xffffu)) * ps[2 * j] + pb[2 * j]);
                        const float hi = silu_f(bf2f((bf16_t)(v[j] >> 16)) * ps[2 * j + 1] + pb[2 * j + 1]);
                        v[j] = f2bf2(lo, hi);
                    }
                }
                const bool ok = (okmask >> i) & 1u;   // zero padding is applied AFTER the prologue (DD:181-187 -> DD:114)
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = ok ? v[j] : 0u;
                *(u4*)(xbuf + c8 * C::US + p * 16) = v;
            }
        };
        u4 x0[XPT], x1[XPT], x2[XPT];
        unsigned ok0 = 0, ok1 = 0, ok2 = 0;
        float ps0[8], pb0[8], ps1[8], pb1[8], ps2[8], pb2[8];
        issue(x0, ok0, ps0, pb0); advance();
        issue(x1, ok1, ps1, pb1); advance();
        issue(x2, ok2, ps2, pb2); advance();
        stage(x0, ok0, ps0, pb0, smem);
        __syncthreads();                                                     // barrier 0: chunk 0 is staged
#define PCW_STEP(XL, OKL, PSL, PBL, XS, OKS, PSS, PBS, BUF)                                         \
        {                                                                                           \
            if (i >= T) break;                                                                      \
            issue(XL, OKL, PSL, PBL);                                                               \
            stage(XS, OKS, PSS, PBS, smem + (BUF) * C::XB);                                         \
            advance();                                                                              \
            __syncthreads();                                                                        \
            ++i;                                                                                    \
        }
        for (int i = 0; i < T;) {
            PCW_STEP(x0, ok0, ps0, pb0, x1, ok1, ps1, pb1, 1)
            PCW_STEP(x1, ok1, ps1, pb1, x2, ok2, ps2, pb2, 0)
            PCW_STEP(x2, ok2, ps2, pb2, x0, ok0, ps0, pb0, 1)
            PCW_STEP(x0, ok0, ps0, pb0, x1, ok1, ps1, pb1, 0)
            PCW_STEP(x1, ok1, ps1, pb1, x2, ok2, ps2, pb2, 1)
            PCW_STEP(x2, ok2, ps2, pb2, x0, ok0, ps0, pb0, 0)
        }
#undef PCW_STEP
        return;
    }

    // =============================================================== consumers
    const int lane = tid & 63, ns = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;           // column of a 16-wide tile; k-group (octet) of an operand / row group of an accumulator
    __builtin_amdgcn_s_setprio(2);

    // ---- weights: fragment (kx, ky, h) of 32-channel chunk kc = rows [tap][kc * 4 + lg][cb + 16 h + l15][8] of the prepared tensor.  The wave's
    //      ring: column c of its walk (a column = one kx of one chunk: 6 fragments, i = ky * 2 + h) in slot c % 4, two columns ahead of its use
    const int cin8 = P.Cin_total / 8;
    unsigned char* const wring = smem + PcwCfg::W_OFF + ns * PcwCfg::WWAVE;
    const unsigned wring_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)wring;
    const unsigned w_lane0 = (unsigned)((lg * P.Cout + 32 * ns + l15) * 16), w_lane1 = w_lane0 + 256u;      // h = 0 / 1
    int wk = 0, wkc = 0, wkx = 0, wcol = 0, wb_, wy_, wx_, wcy;       // weight cursor: (item, chunk, kernel column) of the column fetched next; its index
    item_at(0, wb_, wy_, wx_, wcy);
    auto weights_issue = [&]() {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int ky = i >> 1, h = i & 1;
            const int row = (ky * 3 + wkx) * cin8 + wkc * NC;
            const bf16_t* sbase = P.weight + ((size_t)row * P.Cout + wcy * C::BN) * 8;                        // (uniform)
            const unsigned dst = wring_addr + (unsigned)((wcol & (PcwCfg::WSLOTS - 1)) * PcwCfg::WCOL + i * 1024);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(h ? w_lane1 : w_lane0), "s"(sbase) : "memory");
        }
        ++wcol;
        if (++wkx == 3) {
            wkx = 0;
            if (++wkc == n32) {
                if (item_at(wk + 1, wb_, wy_, wx_, wcy)) { wkc = 0; ++wk; }
                else { wkc = n32 - 1; wkx = 2; }               // past the end: the last column again (into a slot nobody reads any more)
            }
        }
    };
    weights_issue();
    weights_issue();

    int ik = 0, b, oy0, ox0, cy;
    item_at(0, b, oy0, ox0, cy);
    f32x4 bias4[2];
    auto load_bias = [&](int cy_) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 t = P.bias ? *(const float4*)(P.bias + cy_ * C::BN + 32 * ns + 16 * h + 4 * lg) : make_float4(0.f, 0.f, 0.f, 0.f);
            bias4[h][0] = t.x; bias4[h][1] = t.y; bias4[h][2] = t.z; bias4[h][3] = t.w;
        }
    };
    load_bias(cy);
    f32x4 acc[8][2][2];                               // [row][pixel half][channel half]

    const int xrow_off = lg * C::US + l15 * 16;       // octet lg, pixel l15 of a 16-pixel group
    int ccol = 0;                                     // index of the column multiplied next (slot ccol % 4)
    // one kernel column of one chunk: 96 MFMAs; FIRST: the first column of an item (the bias is the first MFMA's C operand)
    auto column = [&](const unsigned char* xrow, const int kx, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        weights_issue();                               // column ccol + 2
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");          // column ccol has landed (all but the two columns issued after it)
        const unsigned char* wsl = wring + (ccol & (PcwCfg::WSLOTS - 1)) * PcwCfg::WCOL + lane * 16;
        ++ccol;
        bf16x8 w[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i] = *(const bf16x8*)(wsl + i * 1024);
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {          // rows 4 hf .. 4 hf + 3 need staged rows 4 hf .. 4 hf + 5
                bf16x8 x[6];
#pragma unroll
                for (int jr = 0; jr < 6; ++jr) x[jr] = *(const bf16x8*)(xrow + ((4 * hf + jr) * IW + kx + 16 * p) * 16);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            acc[4 * hf + r][p][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ky * 2 + h], x[r + ky], (FIRST && ky == 0) ? bias4[h] : acc[4 * hf + r][p][h], 0, 0, 0);
            }
    };

    const bf16x2 one = __builtin_bit_cast(bf16x2, 0x3f803f80u);
    const int tiles8 = (P.H + 7) / 8;
    __syncthreads();                                   // barrier 0: chunk 0 is staged
    while (true) {
        int nb, noy0, nox0, ncy;
        const bool nvalid = item_at(ik + 1, nb, noy0, nox0, ncy);
        for (int kc = 0; kc < n32; kc += 2) {
            const unsigned char* xr0 = smem + xrow_off;
            if (kc == 0) column(xr0, 0, std::true_type{}); else column(xr0, 0, std::false_type{});
            column(xr0, 1, std::false_type{});
            column(xr0, 2, std::false_type{});
            __syncthreads();
            const unsigned char* xr1 = smem + C::XB + xrow_off;
            column(xr1, 0, std::false_type{});
            column(xr1, 1, std::false_type{});
            column(xr1, 2, std::false_type{});
            __syncthreads();
        }
        // ---- epilogue (as conv3x3_wp16_kernel): bf16, 16-byte stores, GroupNorm partial sums of the values as stored
        __builtin_amdgcn_s_setprio(0);
        const int cb = cy * C::BN + 32 * ns;
        float st[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
        const int c_store = cb + 16 * (lg & 1) + 8 * (lg >> 1);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int oy = oy0 + r;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int ox = ox0 + 16 * p + l15;
                const bool ok = oy < P.H && ox < P.W && !(P.dbg & 16);
                const size_t pix = ((size_t)b * P.H + min(oy, P.H - 1)) * P.W + min(ox, P.W - 1);
                uint2 q[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 a = acc[r][p][h];
                    q[h] = make_uint2(f2bf2(a[0], a[1]), f2bf2(a[2], a[3]));
                    if (P.gn_partial && ok) {
                        const bf16x2 va = __builtin_bit_cast(bf16x2, q[h].x), vb = __builtin_bit_cast(bf16x2, q[h].y);
                        st[h][0] = __builtin_amdgcn_fdot2_f32_bf16(vb, one, __builtin_amdgcn_fdot2_f32_bf16(va, one, st[h][0], false), false);
                        st[h][1] = __builtin_amdgcn_fdot2_f32_bf16(vb, vb, __builtin_amdgcn_fdot2_f32_bf16(va, va, st[h][1], false), false);
                    }
                }
                const auto rx = __builtin_amdgcn_permlane16_swap(q[0].x, q[1].x, false, false);
                const auto ry = __builtin_amdgcn_permlane16_swap(q[0].y, q[1].y, false, false);
                if (ok) *(uint4*)(P.out + pix * P.Cout + c_store) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
            }
        }
        if (P.gn_partial) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int w_ = 0; w_ < 2; ++w_) {
                    float v = st[h][w_];
                    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64);
                    st[h][w_] = v;
                }
            const int ty8 = oy0 / 8;
            if (ty8 < tiles8) {
                constexpr int OCT = C::BN / 8, PER_WAVE = OCT * 2;
                const int o = (lane % PER_WAVE) >> 1, which = lane & 1;
                const int oo = o & 3, hh = oo >> 1, mm = oo & 1;
                const float t00 = __shfl(st[0][0], mm * 32, 64), t01 = __shfl(st[0][1], mm * 32, 64);
                const float t10 = __shfl(st[1][0], mm * 32, 64), t11 = __shfl(st[1][1], mm * 32, 64);
                const float total = hh ? (which ? t11 : t10) : (which ? t01 : t00);
                if (lane < PER_WAVE) {
                    const bool own = (o >> 2) == ns;
                    P.gn_partial[gn_partial_index(b, tiles8 * P.tiles_x * 4, (ty8 * P.tiles_x + ox0 / TW) * 4 + ns, P.Cout / 8, cy * C::BN / 8 + o) + which] =
                        own ? total : 0.0f;
                }
            }
        }
        if (!nvalid) break;
        ++ik; b = nb; oy0 = noy0; ox0 = nox0;
        if (ncy != cy) { cy = ncy; load_bias(cy); }
        __builtin_amdgcn_s_setprio(2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the look-ahead weight columns land before the workgroup gives its LDS back
}

template <bool PRO>
static int launch_pcw(const ConvParams& P, hipStream_t s) {
    using C = PcwCfg::C;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        OFD_HIP(hipGetDevice(&dev));
        OFD_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        OFD_HIP(hipFuncSetAttribute((const void*)conv3x3_pcw_kernel<PRO>, hipFuncAttributeMaxDynamicSharedMemorySize, PcwCfg::LDS_BYTES));
        cus = cus / 8 * 8;
        if (cus < 8) cus = 8;
    }
    const int tiles_y = (P.H + C::ROWS - 1) / C::ROWS;
    const int ntiles = P.tiles_x * tiles_y * P.B, ny = P.Cout / C::BN;
    const int nitems = (ntiles + 7) / 8 * 8 * ny;
    const int grid = nitems < cus ? nitems : cus;
    if ((long)grid * PcwCfg::MAX_ITEMS < nitems) return 1;     // more items per workgroup than its descriptor table holds: not served
    conv3x3_pcw_kernel<PRO><<<grid, 512, PcwCfg::LDS_BYTES, s>>>(P);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}

// ---- Upsample(x2, nearest) + 3x3 (DD:89-93) as its four 2x2 phase convs on the LOW-RES tensor, all four in one workgroup (r03) ---------
// Output pixel (2y + py, 2x + px) reads low-res rows y - 1 + py + {0, 1} and columns x - 1 + px + {0, 1} with the collapsed weights of
// ofd_conv_upsample_phase_weight_prep (4 x [2x2 taps][Cin/8][Cout][8]): 2.25x fewer MACs than the 3x3 over the up-sampled tensor.  The
// shared-slab kernel ran these at 500-900 TF/s of REAL work (conv_igemm_kernel<2, BN>: a weight slab copy and a barrier per tap); here the
// wave-private-weights scheme of conv3x3_wp_kernel with the four phases as four wave pairs of ONE workgroup:
//   * wave = (phase, 32-channel slice): 4 phases x 2 slices = 8 waves = 64 output channels of an 8 x 32 block of low-res pixels, i.e. a
//     16 x 64 block of output pixels.  All eight waves read the SAME staged input tile (10 x 34 pixels with halo, 64 channels per chunk,
//     double buffered): the tile is fetched and written to LDS once for the four phases (the one-launch shared-slab form fetched it from
//     L2 four times);
//   * a wave's A operand (its phase's weights, its 32 output channels) comes straight from L2 into registers, a ring of four fragments
//     ahead of their use; every fragment feeds 8 MFMAs (rows) and every row fragment from LDS two (the two kernel rows): 9 ds_read_b128
//     and 2 weight loads per 16 MFMAs;
//   * one workgroup barrier per 128 MFMAs per wave.
constexpr int PCK = 64, PNC = PCK / 8, PFRAGS = 16, PRING = 4;          // staged channels / octets per chunk; weight fragments per chunk
struct PCfg {
    static constexpr int NTHREADS = 512, IH = 10, NPIX = IH * IW;
    static constexpr int US = (NPIX + 1) * 16, XB = PNC * US, LDS_BYTES = 2 * XB;
    static constexpr int XPT = (NPIX * PNC + NTHREADS - 1) / NTHREADS;
};

__global__ void __launch_bounds__(512, 2) conv_up2_phases_wp_kernel(const ConvParams P) {
    using C = PCfg;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int ns = wave & 1, phase = wave >> 1, py = phase >> 1, px = phase & 1;

    // XCD-aware tile order; channel blocks of a pixel tile adjacent (cy fastest) so that their input tile comes from one L2
    const int tiles_y = (P.H + 7) / 8, tpi = P.tiles_x * tiles_y, ntiles = tpi * P.B, ny = P.Cout / 64;
    const int j = blockIdx.x, gq = j >> 3, cy = gq % ny;
    int tile = (gq / ny) * 8 + (j & 7);
    if (tile >= ntiles) return;
    if (ntiles >= 8) {
        const int q = ntiles / 8, r = ntiles % 8, xcd = tile % 8, idx = tile / 8;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int b = tile / tpi, t_in = tile % tpi;
    const int oy0 = (t_in / P.tiles_x) * 8, ox0 = (t_in % P.tiles_x) * TW;           // low-res block origin
    const int cb = cy * 64 + 32 * ns;                 // this wave's 32 output channels

    // ---- weights of this wave's phase: buffer loads, per-lane offset fixed for the launch, per-fragment offset scalar
    const int cin8 = P.Cin_total / 8, nck = P.total_chunks;                           // 64-channel chunks
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(P.weight + (size_t)phase * 4 * P.Cin_total * P.Cout), 0,
                                                                           4 * P.Cin_total * P.Cout * 2, 0x00020000);
    const int w_lane = (half * P.Cout + cb + l31) * 16;
    const int w_row = P.Cout * 16;                    // bytes per [Cin/8] row
    auto load_w = [&](int kc, int fi) -> u4 {         // fragment fi = (ks, tx, ty) of 64-channel chunk kc; tap = ty * 2 + tx
        const int ty = fi & 1, tx = (fi >> 1) & 1, ks = fi >> 2;
        const int row = (ty * 2 + tx) * cin8 + kc * PNC + ks * 2;
        return __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, w_lane, row * w_row, 0));
    };
    u4 ring[PRING];
#pragma unroll
    for (int i = 0; i < PRING; ++i) ring[i] = load_w(0, i);

    // ---- input staging map: unit u = tid + i * 512 -> octet c8 = tid % 8, tile pixel p = u / 8 (low-res tile with a one-pixel halo)
    const int c8 = tid % PNC;
    int pyx[C::XPT];
    unsigned okmask = 0;
#pragma unroll
    for (int i = 0; i < C::XPT; ++i) {
        const int p = min(tid / PNC + i * (C::NTHREADS / PNC), C::NPIX - 1);
        const int ty_ = p / IW, tx_ = p - ty_ * IW;
        const int iy = oy0 - 1 + ty_, ix = ox0 - 1 + tx_;
        const bool ok = iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
        okmask |= (ok ? 1u : 0u) << i;
        pyx[i] = (min(max(iy, 0), P.H - 1) << 16) | min(max(ix, 0), P.W - 1);
    }
    const ConvSrcDev& S = P.src[0];
    auto load_x = [&](int kc, u4 (&xs)[C::XPT]) {
        const bf16_t* base = S.ptr + (size_t)b * S.SH * S.SW * S.src_channels + S.ch_offset + kc * PCK + c8 * 8;
#pragma unroll
        for (int i = 0; i < C::XPT; ++i) {
            const int sy = pyx[i] >> 16, sx = pyx[i] & 0xffff;
            xs[i] = *(const u4*)(base + ((size_t)sy * S.SW + sx) * S.src_channels);
        }
    };
    auto write_x = [&](const u4 (&xs)[C::XPT], unsigned char* xbuf) {
#pragma unroll
        for (int i = 0; i < C::XPT; ++i) {
            const int p = min(tid / PNC + i * (C::NTHREADS / PNC), C::NPIX - 1);
            u4 v = xs[i];
            const bool ok = (okmask >> i) & 1u;       // zero padding of the up-sampled tensor = zero padding of the low-res one
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = ok ? v[q] : 0u;
            *(u4*)(xbuf + c8 * C::US + p * 16) = v;
        }
    };

    f32x16 acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[r][k] = 0.0f;

    u4 xs[C::XPT];
    load_x(0, xs);
    write_x(xs, smem);

    // this wave's fragment origin: octet `half` of a k-step, staged row py, column px + l31 (rows / columns of the tile count from the halo)
    const int xrow_off = half * C::US + ((py * IW) + px + l31) * 16;
    auto chunk = [&](const int kc, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const int kn = LAST ? kc : kc + 1;
        if constexpr (!LAST) load_x(kn, xs);
        __syncthreads();                              // tile kc complete; every wave is done reading the other buffer
        const unsigned char* xrow = smem + (kc & 1) * C::XB + xrow_off;
        unsigned char* xnext = smem + ((kc + 1) & 1) * C::XB;
#pragma unroll
        for (int g = 0; g < 8; ++g) {                 // g = (ks, tx): k-step of 16 channels, kernel column
            const int ks = g >> 1, tx = g & 1;
            bf16x8 x[9];
#pragma unroll
            for (int jr = 0; jr < 9; ++jr) x[jr] = *(const bf16x8*)(xrow + (jr * IW + tx) * 16 + ks * 2 * C::US);
            bf16x8 a[2];
#pragma unroll
            for (int ty = 0; ty < 2; ++ty) {
                const int fi = g * 2 + ty;
                a[ty] = as_frag(ring[fi % PRING]);
                if (fi + PRING < PFRAGS) ring[fi % PRING] = load_w(kc, fi + PRING);
                else if constexpr (!LAST) ring[fi % PRING] = load_w(kn, fi + PRING - PFRAGS);
            }
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int ty = 0; ty < 2; ++ty) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ty], x[r + ty], acc[r], 0, 0, 0);
            if constexpr (!LAST) { if (g == 2) write_x(xs, xnext); }
        }
    };
    for (int kc = 0; kc < nck - 1; ++kc) chunk(kc, std::false_type{});
    chunk(nck - 1, std::true_type{});

    // ---- epilogue: bias, bf16, 16-byte stores to pixel (2 y + py, 2 x + px) of the (2H, 2W) tensor
    float4 bias4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias4[g] = P.bias ? *(const float4*)(P.bias + cb + 8 * g + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int OW = 2 * P.W;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int y = oy0 + r, x = ox0 + l31;
        const bool ok = y < P.H && x < P.W && !(P.dbg & 16);
        const size_t pix = ((size_t)b * 2 * P.H + (2 * min(y, P.H - 1) + py)) * OW + (2 * min(x, P.W - 1) + px);
        uint2 q[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            q[g] = make_uint2(f2bf2(acc[r][4 * g] + bias4[g].x, acc[r][4 * g + 1] + bias4[g].y),
                              f2bf2(acc[r][4 * g + 2] + bias4[g].z, acc[r][4 * g + 3] + bias4[g].w));
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
            const auto rx = __builtin_amdgcn_permlane32_swap(q[g].x, q[g + 1].x, false, false);
            const auto ry = __builtin_amdgcn_permlane32_swap(q[g].y, q[g + 1].y, false, false);
            if (ok) *(uint4*)(P.out + pix * P.Cout + cb + 8 * g + 8 * half) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
        }
    }
}

}  // namespace wp

#if OFD_WP_STAMPS
extern "C" int ofd_dbg_wp_stamps(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(wp::g_wp_stamps), bytes, 0, hipMemcpyDeviceToHost);
}
#endif

// 3x3, stride 1, sources of mode 0 (same size) or 1 (nearest x2): called from conv_forward_impl
int launch_conv3x3_wp(const ConvParams& P0, bool wide, hipStream_t s) {
    static const int cy_fast = getenv("OFD_CONV_WP_CYFAST") ? atoi(getenv("OFD_CONV_WP_CYFAST")) : 1;
    ConvParams P = P0;
    P.cy_fast = cy_fast && P.Cout / (wide ? 128 : 64) > 1;
    // 256-channel blocks (8 waves over one staged tile: half the tile loads, LDS writes and prologue arithmetic per MFMA): OFD_CONV_WP_BN256=1
    static const int bn256 = getenv("OFD_CONV_WP_BN256") ? atoi(getenv("OFD_CONV_WP_BN256")) : 0;
    if (wide && bn256 && P.Cout % 256 == 0) return P.in_scale ? wp::launch<8, 1, true>(P, s) : wp::launch<8, 1, false>(P, s);
    // producer / consumer kernel on the wide layers too, as Cout / 64 channel blocks per pixel tile (A/B switch: OFD_CONV_PC=2)
    static const int pcw = getenv("OFD_CONV_PC") ? atoi(getenv("OFD_CONV_PC")) : 1;
    if (wide && pcw == 2 && wp::pc_serves(P)) {
        const int r = P.in_scale ? wp::launch_pc<true>(P, s) : wp::launch_pc<false>(P, s);
        if (r != 1) return r;
    }
    // producer / consumer form of the 128-channel-block kernel: opt-in (OFD_CONV_PCW=1).  Same-box A/B against conv3x3_wp16_kernel
    // (profiles/r04_pcw_ab.txt): 3-7 % SLOWER per layer, class 8.75 -> 9.2 ms per step -- with one MFMA wave per SIMD every barrier, column
    // start and epilogue of that wave is matrix-pipe idle time, which two independent 4-wave workgroups per CU cover for each other
    static const int pcwide = getenv("OFD_CONV_PCW") ? atoi(getenv("OFD_CONV_PCW")) : 0;
    if (wide && pcwide && wp::pc_serves(P) && P.Cout % 128 == 0) {
        const int r = P.in_scale ? wp::launch_pcw<true>(P, s) : wp::launch_pcw<false>(P, s);
        if (r != 1) return r;
    }
    // MFMA 16x16x32 form of the 128-channel-block kernel for the plain / prologue / statistics epilogues (OFD_CONV_WP16=0: off)
    static const int wp16 = getenv("OFD_CONV_WP16") ? atoi(getenv("OFD_CONV_WP16")) : 1;
    if (wide && wp16 && !P.residual && !P.residual_b && !P.res_act && !P.split && !P.pool2 && !P.out2)
        return P.in_scale ? wp::launch16<true>(P, s) : wp::launch16<false>(P, s);
    // producer / consumer form for the 64 -> 64 layers with plain / prologue / statistics epilogues (OFD_CONV_PC=0: off)
    const int pc = getenv("OFD_CONV_PC") ? atoi(getenv("OFD_CONV_PC")) : 1;          // (read per call: the tests compare the two kernels in one process)
    if (!wide && pc && wp::pc_serves(P)) {
        const int r = P.in_scale ? wp::launch_pc<true>(P, s) : wp::launch_pc<false>(P, s);
        if (r != 1) return r;
    }
    if (P.in_scale) return wide ? wp::launch<4, 1, true>(P, s) : wp::launch<2, 2, true>(P, s);
    return wide ? wp::launch<4, 1, false>(P, s) : wp::launch<2, 2, false>(P, s);
}

}  // namespace ofd

namespace ofd {
// Upsample(x2) + 3x3 as four 2x2 phase convs in one launch (ConvParams of the ksize-2 / phase_all form): 1 = shape not served
int launch_conv_up2_phases_wp(const ConvParams& P, hipStream_t s) {
    static const bool off = getenv("OFD_PHASE_WP") && atoi(getenv("OFD_PHASE_WP")) == 0;
    if (off || !P.phase_all || P.n_src != 1 || P.src[0].mode != 0 || P.Cout % 64 || P.Cin_total % 64 || P.residual || P.res_act || P.gn_partial ||
        P.in_scale || P.split || P.total_chunks < 1)
        return 1;
    static bool attr_set = false;
    if (!attr_set) {
        OFD_HIP(hipFuncSetAttribute((const void*)wp::conv_up2_phases_wp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, wp::PCfg::LDS_BYTES));
        attr_set = true;
    }
    const int tiles_y = (P.H + 7) / 8, ntiles = P.tiles_x * tiles_y * P.B, ny = P.Cout / 64;
    wp::conv_up2_phases_wp_kernel<<<(ntiles + 7) / 8 * 8 * ny, wp::PCfg::NTHREADS, wp::PCfg::LDS_BYTES, s>>>(P);
    OFD_LAUNCH_CHECK();
    return OFD_OK;
}
}  // namespace ofd
